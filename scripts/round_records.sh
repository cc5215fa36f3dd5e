#!/bin/bash
# The round's records in one GPU session (run on the GPU box from the repo root; everything lands under gpurun_out/<dir>):
#   separate --pmc passes (on-chip kernel at two run lengths, streaming kernels on the 1M and 16M meshes, the assembly),
#   their summary (profiles/pmc_summary.json is what bench.py reads), kernel stats of the driver's bench command, the bench
#   line itself, the size-scaling lines and the in-kernel phase stamps.
#       bash scripts/round_records.sh gpurun_out/rec r03
set -e
OUT=${1:-gpurun_out/rec}
TAG=${2:-r03}
ROOT=$PWD
mkdir -p "$OUT"
bash scripts/pmc_passes.sh "$OUT/pmc" > "$OUT/pmc_passes.log" 2>&1
echo "pmc passes done" > "$OUT/progress.txt"
HOWS="ctile" bash scripts/pmc_assembly.sh "$OUT/pmc_asm" > "$OUT/pmc_asm.log" 2>&1
echo "assembly passes done" >> "$OUT/progress.txt"
python3 scripts/pmc_summarize.py "$OUT/pmc" "$TAG" > "$OUT/pmc_summarize.log" 2>&1
python3 scripts/pmc_assembly_summarize.py "$OUT/pmc_asm" "$TAG" > "$OUT/pmc_asm_summarize.log" 2>&1
mkdir -p "$OUT/profiles" && cp profiles/pmc_summary.json profiles/${TAG}_pmc_summary.json profiles/${TAG}_pmc_counters.csv profiles/${TAG}_pmc_assembly.csv "$OUT/profiles/"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 > "$ROOT/$OUT/bench_under_rocprof.json" 2> "$ROOT/$OUT/bench_under_rocprof.err")
echo "kernel stats done" >> "$OUT/progress.txt"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"
echo "bench done" >> "$OUT/progress.txt"
: > "$OUT/size_scaling.jsonl"
for wl in plate100k hole1m plate4m multihole16m; do
    python3 bench.py --workload $wl --no-cpu-baseline --no-hbm-resident >> "$OUT/size_scaling.jsonl" 2>> "$OUT/size_scaling.err"
done
python3 bench.py --workload multihole16m --precision fp32 --no-cpu-baseline --no-hbm-resident >> "$OUT/size_scaling.jsonl" 2>> "$OUT/size_scaling.err"
echo "size scaling done" >> "$OUT/progress.txt"
# (The in-kernel phase stamps -- python3 scripts/persist_phases.py <out>, MAG_TUNE_PERSIST_TRIANGLES=1 for the triangle walk --
# are not part of the routine run any more: after the round's last change of the on-chip kernel its DIAGNOSTIC build faulted on
# the GPU (the product build, the same source without the stamps, passes every test, the soak and the stress runs), and a
# diagnostic is not worth a GPU fault.  profiles/r03_persist_phases_*.json are the records from one commit earlier.)
echo "all done" >> "$OUT/progress.txt"
