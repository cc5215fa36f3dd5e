"""Aggregate the rocprofv3 --pmc passes of scripts/pmc_passes.sh: mean counter value per launch of the CG kernels ->
profiles/r01_pmc_counters.csv, and the HBM-traffic figures bench.py reports as roofline.traffic ->
profiles/r01_pmc_summary.json ((2 * FETCH_SIZE + WRITE_SIZE) KB: MI355X_MICROARCH.md, HBM section).
The on-chip kernel is one launch per solve: runs of 40 and 120 iterations give its traffic per iteration (difference)
and its one-off part (loading the mesh and the right-hand side, storing x)."""
import collections
import csv
import glob
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "pmc")
acc = collections.defaultdict(lambda: collections.defaultdict(list))  # (kernel, iters) -> counter -> values
for path in glob.glob(os.path.join(src, "v*_it*_pass*", "**", "*counter_collection.csv"), recursive=True):
    m = re.search(r"v(\d+)_it(\d+)_pass", path)
    iters = int(m.group(2))
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if "k_cg_fused" not in name and "k_cg_persist" not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            acc[(short, iters)][row["Counter_Name"]].append(float(row["Counter_Value"]))
mean = lambda v: sum(v) / len(v)
with open(os.path.join(root, "profiles", "r01_pmc_counters.csv"), "w") as f:
    f.write("kernel,cg_iterations_in_run,counter,launches,mean_per_launch\n")
    for (k, iters) in sorted(acc):
        for c in sorted(acc[(k, iters)]):
            v = acc[(k, iters)][c]
            f.write(f'"{k}",{iters},{c},{len(v)},{mean(v):.1f}\n')
note = ("FETCH_SIZE doubled (gfx950 reports half of a wide coalesced stream, MI355X_MICROARCH.md HBM); WRITE_SIZE as "
        "reported; separate --pmc passes, scripts/prof_iter.py, launches isolated by the profiler")
summary = {}
hbm = lambda c: (2.0 * mean(c["FETCH_SIZE"]) + mean(c["WRITE_SIZE"])) * 1024.0
for (k, iters), counters in acc.items():
    if "k_cg_fused_dma<512" in k and "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        summary["hole1m:tile512:kernel1"] = {"kernel": k, "FETCH_SIZE_KB": mean(counters["FETCH_SIZE"]),
                                             "WRITE_SIZE_KB": mean(counters["WRITE_SIZE"]),
                                             "hbm_bytes_per_launch": hbm(counters), "note": note}
pk = sorted((iters, k) for (k, iters) in acc if "k_cg_persist<512" in k and "FETCH_SIZE" in acc[(k, iters)]
            and "WRITE_SIZE" in acc[(k, iters)])
if len(pk) >= 2:
    (i0, k0), (i1, k1) = pk[0], pk[-1]
    b0, b1 = hbm(acc[(k0, i0)]), hbm(acc[(k1, i1)])
    per_it = (b1 - b0) / (i1 - i0)
    summary["hole1m:tile512:kernel2"] = {"kernel": k0, "runs": {str(i0): b0, str(i1): b1},
                                         "hbm_bytes_per_iteration": per_it,
                                         # two single-launch samples: the intercept is noisy (the sweeps of a run
                                         # repeat a data-dependent number of times); never below zero
                                         "hbm_bytes_setup": max(0.0, b0 - per_it * i0),
                                         "note": note + "; one launch per solve, two run lengths"}
json.dump(summary, open(os.path.join(root, "profiles", "r01_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
