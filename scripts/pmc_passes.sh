#!/bin/bash
# Separate rocprofv3 --pmc passes (never combined with --stats / sys-trace) over scripts/prof_iter.py: 40 and 120 CG
# iterations of the 1M-triangle bench workload (two lengths: the on-chip kernel is ONE launch per solve, its traffic
# per iteration is the difference), for the default kernel choice and for the streaming kernels (--cg-variant 1).
# Run on the GPU box from the repo root:
#     bash scripts/pmc_passes.sh gpurun_out/pmc
# then  python scripts/pmc_summarize.py gpurun_out/pmc  writes profiles/r01_pmc_counters.csv / r01_pmc_summary.json.
set -e
OUT=${1:-gpurun_out/pmc}
ROOT=$PWD
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
i=0
for run in "2 40" "2 120" "1 40"; do
    set -- $run
    variant=$1
    iters=$2
    for group in "FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
                 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
                 "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
        i=$((i + 1))
        d="v${variant}_it${iters}_pass$i"
        echo "$d: $group"
        # a refused counter set aborts the tool but leaves it hanging: short limit, and stop at the first failed pass
        timeout -k 10 120 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$ROOT/$OUT/$d" -- \
            python3 "$ROOT/scripts/prof_iter.py" --cg-variant $variant --iters $iters > "$ROOT/$OUT/$d.log" 2>&1 \
            || { echo "$d failed"; exit 1; }
    done
done
echo done
