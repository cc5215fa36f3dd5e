"""Aggregate the rocprofv3 --pmc passes of scripts/pmc_passes.sh: mean counter value per launch of the CG kernels ->
profiles/r01_pmc_counters.csv, and the HBM-traffic figure bench.py reports as roofline.traffic ->
profiles/r01_pmc_summary.json ((2 * FETCH_SIZE + WRITE_SIZE) KB: MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "pmc")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(src, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if "k_cg_fused" not in name and "k_operator_lds" not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(os.path.join(root, "profiles", "r01_pmc_counters.csv"), "w") as f:
    f.write("kernel,counter,launches,mean_per_launch\n")
    for k in sorted(acc):
        for c in sorted(acc[k]):
            v = acc[k][c]
            f.write(f'"{k}",{c},{len(v)},{sum(v) / len(v):.1f}\n')
summary = {}
for k, counters in acc.items():
    if "k_cg_fused_dma<512" in k and "FETCH_SIZE" in counters:
        fetch = sum(counters["FETCH_SIZE"]) / len(counters["FETCH_SIZE"])
        write = sum(counters["WRITE_SIZE"]) / len(counters["WRITE_SIZE"])
        summary["hole1m:tile512:variant1"] = {
            "kernel": k, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
            "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
            "note": "FETCH_SIZE doubled (gfx950 reports half of a wide coalesced stream, MI355X_MICROARCH.md HBM); "
                    "WRITE_SIZE as reported; separate --pmc passes, scripts/prof_iter.py, launches isolated by the "
                    "profiler"}
json.dump(summary, open(os.path.join(root, "profiles", "r01_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
