"""Counters of the assembly kernels from scripts/pmc_assembly.sh -> profiles/<tag>_pmc_assembly.csv (mean per launch)."""
import collections
import csv
import glob
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "pmc_asm")
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(src, "asm_*_pass*", "**", "*counter_collection.csv"), recursive=True):
    how = re.search(r"asm_([a-z]+)_pass", path).group(1)
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if "k_assemble" not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            acc[(how, short)][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = os.path.join(root, "profiles", f"{tag}_pmc_assembly.csv")
with open(out, "w") as f:
    f.write("assembly,kernel,counter,launches,mean_per_launch\n")
    for key in sorted(acc):
        for c in sorted(acc[key]):
            v = acc[key][c]
            f.write(f'{key[0]},"{key[1]}",{c},{len(v)},{sum(v) / len(v):.1f}\n')
print(open(out).read())
