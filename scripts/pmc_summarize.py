"""Aggregate the rocprofv3 --pmc passes of scripts/pmc_passes.sh: mean counter value per launch of the CG kernels ->
profiles/<tag>_pmc_counters.csv, and what bench.py reads -> profiles/pmc_summary.json (+ a copy named
profiles/<tag>_pmc_summary.json for the round's record).  The summary carries `_meta.source_hash`, the digest of the kernel
sources the profiled library was linked from (magnetite_amd/libmagnetite_hip.srchash, written by csrc/Makefile):
bench.py compares it with the digest of the library IT loads and marks the counters stale on a mismatch.

      python scripts/pmc_summarize.py <dir of the passes> <tag>

  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB, as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE
  reports half of a wide coalesced stream, WRITE_SIZE is exact for 16-byte-per-lane stores.
  Streaming kernels (one launch per iteration / per SpMV): per launch.
  On-chip kernel (one launch per solve): runs of 40 and 120 iterations give per-iteration figures (difference / 80)
  and the one-off part.  Its utilisation figures per CG iteration:
      valu_busy      = d SQ_ACTIVE_INST_VALU * waves / (SIMDs * d SQ_WAVE_CYCLES)   (both in quad-cycles; every wave lives
                       for the whole launch, so SQ_WAVE_CYCLES / waves is the launch's length in quad-cycles)
      lds_busy       = d SQ_LDS_IDX_ACTIVE / (CUs * cycles per iteration)
      lds_conflict   = d SQ_LDS_BANK_CONFLICT / d SQ_LDS_IDX_ACTIVE
      wait / issue_stall / active = d SQ_WAIT_ANY, d SQ_WAIT_INST_ANY, d SQ_ACTIVE_INST_ANY over d SQ_WAVE_CYCLES
"""
import collections
import csv
import glob
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "pmc")
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
SIMDS, CUS = 1024, 256
acc = collections.defaultdict(lambda: collections.defaultdict(list))  # (workload, kernel, iters) -> counter -> values
for path in glob.glob(os.path.join(src, "*_v*_it*_pass*", "**", "*counter_collection.csv"), recursive=True):
    m = re.search(r"([a-z0-9]+)_v(\d+)_it(\d+)_pass", path)
    wl, iters = m.group(1), int(m.group(3))
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if not any(k in name for k in ("k_cg_fused", "k_cg_persist", "k_operator_lds", "k_assemble", "k_element")):
                continue
            short = name.split("(")[0].replace("void ", "")
            acc[(wl, short, iters)][row["Counter_Name"]].append(float(row["Counter_Value"]))
mean = lambda v: sum(v) / len(v)
with open(os.path.join(root, "profiles", f"{tag}_pmc_counters.csv"), "w") as f:
    f.write("workload,kernel,cg_iterations_in_run,counter,launches,mean_per_launch\n")
    for key in sorted(acc):
        for c in sorted(acc[key]):
            v = acc[key][c]
            f.write(f'{key[0]},"{key[1]}",{key[2]},{c},{len(v)},{mean(v):.1f}\n')
note = ("FETCH_SIZE doubled (gfx950 reports half of a wide coalesced stream, MI355X_MICROARCH.md HBM); WRITE_SIZE as "
        "reported; separate --pmc passes, scripts/prof_iter.py, launches isolated by the profiler")
summary = {}
hbm = lambda c: (2.0 * mean(c["FETCH_SIZE"]) + mean(c["WRITE_SIZE"])) * 1024.0
for (wl, k, iters), counters in acc.items():
    if "FETCH_SIZE" not in counters or "WRITE_SIZE" not in counters:
        continue
    m = re.search(r"<(\d+)", k)
    tile = m.group(1) if m else "0"
    if "k_cg_fused_dma" in k:
        key = f"{wl}:tile{tile}:kernel1"
    elif "k_operator_lds" in k and "false, false" in k.replace(" ", "").replace(",", ", "):
        key = f"{wl}:tile{tile}:spmv"
    elif "k_assemble" in k:
        key = f"{wl}:{k.split('::')[-1].split('<')[0]}"
    else:
        continue
    summary[key] = {"kernel": k, "FETCH_SIZE_KB": mean(counters["FETCH_SIZE"]), "WRITE_SIZE_KB": mean(counters["WRITE_SIZE"]),
                    "hbm_bytes_per_launch": hbm(counters), "launches": len(counters["FETCH_SIZE"]), "note": note}
for wl in sorted({w for (w, k, i) in acc}):
    pk = sorted((iters, k) for (w, k, iters) in acc if w == wl and "k_cg_persist<512" in k and "FETCH_SIZE" in acc[(w, k, iters)]
                and "WRITE_SIZE" in acc[(w, k, iters)])
    if len(pk) < 2:
        continue
    (i0, k0), (i1, k1) = pk[0], pk[-1]
    c0, c1 = acc[(wl, k0, i0)], acc[(wl, k1, i1)]
    b0, b1 = hbm(c0), hbm(c1)
    per_it = (b1 - b0) / (i1 - i0)
    d = lambda name: (mean(c1[name]) - mean(c0[name])) / (i1 - i0) if name in c0 and name in c1 else None
    util, per = {}, {n: d(n) for n in sorted(c1)}
    waves = mean(c1["SQ_WAVES"]) if "SQ_WAVES" in c1 else None
    if waves and per.get("SQ_WAVE_CYCLES"):
        wc = per["SQ_WAVE_CYCLES"]
        cyc_it = 4.0 * wc / waves  # shader cycles per CG iteration, from the waves' own lifetime
        util["cycles_per_iteration"] = cyc_it
        if per.get("SQ_ACTIVE_INST_VALU"):
            util["valu_busy"] = per["SQ_ACTIVE_INST_VALU"] * waves / (SIMDS * wc)
        for name, cn in (("wait", "SQ_WAIT_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"), ("active", "SQ_ACTIVE_INST_ANY")):
            if per.get(cn):
                util[name] = per[cn] / wc
        if per.get("SQ_LDS_IDX_ACTIVE"):
            util["lds_busy"] = per["SQ_LDS_IDX_ACTIVE"] / (CUS * cyc_it)
            if per.get("SQ_LDS_BANK_CONFLICT"):
                util["lds_conflict"] = per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"]
        if per.get("SQ_INSTS_VALU"):
            util["valu_wave_instructions_per_iteration"] = per["SQ_INSTS_VALU"]
        if per.get("SQ_INSTS_LDS"):
            util["lds_wave_instructions_per_iteration"] = per["SQ_INSTS_LDS"]
    summary[f"{wl}:tile512:kernel2"] = {
        "kernel": k0, "runs": {str(i0): b0, str(i1): b1}, "hbm_bytes_per_iteration": per_it,
        # two single-launch samples: the intercept is noisy (the sweeps of a run repeat a data-dependent number of
        # times); never below zero
        "hbm_bytes_setup": max(0.0, b0 - per_it * i0), "waves": waves, "utilisation": util,
        "counters_per_iteration": per, "note": note + "; one launch per solve, two run lengths"}
sys.path.insert(0, root)
from magnetite_amd import _lib  # noqa: E402  (no GPU call: only reads the digest next to the library)
summary["_meta"] = {"source_hash": _lib.built_source_hash(), "hashed_sources": list(_lib.HASHED_SOURCES), "tag": tag,
                    "passes": os.path.relpath(src, root)}
for name in (f"{tag}_pmc_summary.json", "pmc_summary.json"):
    json.dump(summary, open(os.path.join(root, "profiles", name), "w"), indent=1)
print(json.dumps(summary, indent=1))
