#!/bin/bash
# The round's records in one GPU session (run on the GPU box from the repo root; everything lands under gpurun_out/<dir>):
#   separate --pmc passes (on-chip kernel at two run lengths, streaming kernels on the 1M and 16M meshes, the assembly),
#   their summary (profiles/pmc_summary.json is what bench.py reads), kernel stats of the driver's bench command, the bench
#   line itself, the size-scaling lines and the in-kernel phase stamps.
#       bash scripts/round_records.sh gpurun_out/rec r04
set -e
OUT=${1:-gpurun_out/rec}
TAG=${2:-r04}
ROOT=$PWD
mkdir -p "$OUT"
if [ -z "$SKIP_PMC" ]; then  # (SKIP_PMC=1: the counters of these very sources are already in profiles/pmc_summary.json)
bash scripts/pmc_passes.sh "$OUT/pmc" > "$OUT/pmc_passes.log" 2>&1
echo "pmc passes done" > "$OUT/progress.txt"
HOWS="ctile" bash scripts/pmc_assembly.sh "$OUT/pmc_asm" > "$OUT/pmc_asm.log" 2>&1
echo "assembly passes done" >> "$OUT/progress.txt"
python3 scripts/pmc_summarize.py "$OUT/pmc" "$TAG" > "$OUT/pmc_summarize.log" 2>&1
python3 scripts/pmc_assembly_summarize.py "$OUT/pmc_asm" "$TAG" > "$OUT/pmc_asm_summarize.log" 2>&1
mkdir -p "$OUT/profiles" && cp profiles/pmc_summary.json profiles/${TAG}_pmc_summary.json profiles/${TAG}_pmc_counters.csv profiles/${TAG}_pmc_assembly.csv "$OUT/profiles/"
fi
if [ "${PART:-AB}" != "B" ]; then  # PART=A / PART=B: the session in two halves (a gpurun call is limited to 20 minutes)
bash scripts/pmc_onchip_compare.sh "$OUT/pmc_cmp" "frontal1m hole1m" > "$OUT/pmc_cmp.log" 2>&1
python3 scripts/pmc_onchip_compare.py "$OUT/pmc_cmp" "$OUT/pmc_onchip_compare.json" > "$OUT/pmc_onchip_compare.txt" 2>&1
echo "on-chip counter comparison done" >> "$OUT/progress.txt"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 > "$ROOT/$OUT/bench_under_rocprof.json" 2> "$ROOT/$OUT/bench_under_rocprof.err")
echo "kernel stats done" >> "$OUT/progress.txt"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"
echo "bench done" >> "$OUT/progress.txt"
fi
if [ "${PART:-AB}" = "A" ]; then exit 0; fi
: > "$OUT/size_scaling.jsonl"
for wl in plate100k hole1m frontal1m plate4m multihole16m; do
    python3 bench.py --workload $wl --no-cpu-baseline --no-hbm-resident --no-unstructured >> "$OUT/size_scaling.jsonl" 2>> "$OUT/size_scaling.err"
done
python3 bench.py --workload multihole16m --precision fp32 --no-cpu-baseline --no-hbm-resident >> "$OUT/size_scaling.jsonl" 2>> "$OUT/size_scaling.err"
MAG_TUNE_PERSIST_TRIANGLES=1 python3 bench.py --workload frontal1m --no-cpu-baseline --no-hbm-resident --no-unstructured >> "$OUT/size_scaling.jsonl" 2>> "$OUT/size_scaling.err"
echo "size scaling done" >> "$OUT/progress.txt"
# in-kernel phase stamps (diagnostic build; round 3's GPU fault of this build was an SGPR hazard of the inline-asm stores, fixed
# in round 4: DESIGN.md R4.1): the edge-block kernel on the structured meshes and the overflow instantiation on the frontal
# mesh, the triangle walk, and the multi-GPU kernel with two ranks sharing the GPU
python3 scripts/persist_phases.py --workloads=hole1m,plate100k,frontal1m "$OUT/persist_phases_blocks.json" > "$OUT/persist_phases_blocks.log" 2>&1
python3 scripts/persist_phases.py --triangles --workloads=hole1m,frontal1m "$OUT/persist_phases_triangles.json" > "$OUT/persist_phases_triangles.log" 2>&1
python3 scripts/persist_phases_mg.py "$OUT/persist_phases_mg.json" > "$OUT/persist_phases_mg.log" 2>&1
echo "phases done" >> "$OUT/progress.txt"
# multi-GPU rehearsals (ranks share the ONE GPU, gloo): two ranks at four tiles per workgroup, edge blocks and triangle walk;
# config 4 across four ranks against its fixture
: > "$OUT/multirank_rehearsals.jsonl"
MAG_TUNE_PERSIST_K=4 python3 bench.py --gpus 2 --share-gpu --partition strong --workload hole1m --exchange inboxes --no-cpu-baseline --no-hbm-resident --steps 3 --warmup 1 --check-fixture >> "$OUT/multirank_rehearsals.jsonl" 2>> "$OUT/multirank_rehearsals.err"
MAG_TUNE_PERSIST_K=4 MAG_TUNE_PERSIST_MG_BLOCKS=0 python3 bench.py --gpus 2 --share-gpu --partition strong --workload hole1m --exchange inboxes --no-cpu-baseline --no-hbm-resident --steps 3 --warmup 1 --check-fixture >> "$OUT/multirank_rehearsals.jsonl" 2>> "$OUT/multirank_rehearsals.err"
# the unstructured 1M mesh across two ranks: edge blocks with overflow records (round 4) and the triangle walk; its fixture stops at 1e-10
MAG_TUNE_PERSIST_K=4 python3 bench.py --gpus 2 --share-gpu --partition strong --workload frontal1m --tol 1e-10 --exchange inboxes --no-cpu-baseline --no-hbm-resident --no-unstructured --steps 2 --warmup 1 --check-fixture >> "$OUT/multirank_rehearsals.jsonl" 2>> "$OUT/multirank_rehearsals.err"
MAG_TUNE_PERSIST_K=4 MAG_TUNE_PERSIST_MG_OVERFLOW=0 python3 bench.py --gpus 2 --share-gpu --partition strong --workload frontal1m --tol 1e-10 --exchange inboxes --no-cpu-baseline --no-hbm-resident --no-unstructured --steps 2 --warmup 1 --check-fixture >> "$OUT/multirank_rehearsals.jsonl" 2>> "$OUT/multirank_rehearsals.err"
python3 bench.py --gpus 4 --share-gpu --partition strong --workload plate4m --exchange inboxes --cg-variant 1 --no-cpu-baseline --no-hbm-resident --steps 1 --warmup 1 --check-fixture >> "$OUT/multirank_rehearsals.jsonl" 2>> "$OUT/multirank_rehearsals.err"
echo "all done" >> "$OUT/progress.txt"
