#!/bin/bash
# Separate rocprofv3 --pmc passes (never combined with --stats / sys-trace) over scripts/prof_iter.py:
#   hole1m  cg_variant 2, 40 and 120 CG iterations: the on-chip kernel is ONE launch per solve, so its per-iteration
#           counters are the difference of two run lengths / 80;
#   hole1m  cg_variant 1, 40 iterations + 20 plain SpMV launches: streaming iteration kernel and SpMV, cache-resident;
#   multihole16m, no solve: 23 launches each of the streaming iteration kernel and the SpMV, HBM-resident.
# Run on the GPU box from the repo root:   bash scripts/pmc_passes.sh gpurun_out/pmc
# then   python scripts/pmc_summarize.py gpurun_out/pmc   writes profiles/r02_pmc_counters.csv / r02_pmc_summary.json.
set -e
OUT=${1:-gpurun_out/pmc}
ROOT=$PWD
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
i=0
for run in "hole1m 2 40" "hole1m 2 120" "hole1m 1 40" "multihole16m 1 0"; do
    set -- $run
    wl=$1
    variant=$2
    iters=$3
    extra=""
    if [ "$iters" = "0" ]; then extra="--no-solve"; fi
    for group in "FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
                 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
                 "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
        i=$((i + 1))
        d="${wl}_v${variant}_it${iters}_pass$i"
        echo "$d: $group"
        # a refused counter set aborts the tool but leaves it hanging: short limit, and stop at the first failed pass
        timeout -k 10 240 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$ROOT/$OUT/$d" -- \
            python3 "$ROOT/scripts/prof_iter.py" --workload $wl --cg-variant $variant --iters $iters $extra \
            > "$ROOT/$OUT/$d.log" 2>&1 || { echo "$d failed"; tail -5 "$ROOT/$OUT/$d.log"; exit 1; }
    done
done
echo done
