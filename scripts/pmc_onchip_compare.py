"""Per-iteration PMC counters of k_cg_persist on several workloads, side by side (passes of scripts/pmc_onchip_compare.sh:
runs of 40 and 120 iterations, difference / 80).   python scripts/pmc_onchip_compare.py <dir> [out.json]"""
import collections
import csv
import glob
import json
import os
import re
import sys

src = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(src, "*_v2_it*_pass*", "**", "*counter_collection.csv"), recursive=True):
    m = re.search(r"([a-z0-9]+)_v2_it(\d+)_pass", path)
    wl, iters = m.group(1), int(m.group(2))
    with open(path) as f:
        for row in csv.DictReader(f):
            if "k_cg_persist" in row["Kernel_Name"]:
                acc[(wl, iters)][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for wl in sorted({k[0] for k in acc}):
    a, b = acc.get((wl, 40)), acc.get((wl, 120))
    if not a or not b:
        continue
    d = {c: (sum(b[c]) / len(b[c]) - sum(a[c]) / len(a[c])) / 80.0 for c in b if c in a}
    waves = sum(b["SQ_WAVES"]) / len(b["SQ_WAVES"]) if "SQ_WAVES" in b else 0
    if "SQ_WAVE_CYCLES" in d and waves:
        d["cycles_per_iteration"] = 4.0 * d["SQ_WAVE_CYCLES"] / waves
        d["valu_busy"] = d["SQ_ACTIVE_INST_VALU"] * waves / (1024.0 * d["SQ_WAVE_CYCLES"]) if "SQ_ACTIVE_INST_VALU" in d else None
        d["lds_busy"] = d["SQ_LDS_IDX_ACTIVE"] / (256.0 * d["cycles_per_iteration"]) if "SQ_LDS_IDX_ACTIVE" in d else None
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_share"] = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]
    out[wl] = d
keys = sorted({k for d in out.values() for k in d})
print("%-28s" % "counter / iteration" + "".join("%16s" % w for w in out))
for k in keys:
    print("%-28s" % k + "".join("%16.4g" % (out[w].get(k) or 0) for w in out))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
