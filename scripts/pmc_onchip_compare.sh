#!/bin/bash
# rocprofv3 --pmc passes (never combined with --stats / sys-trace) of the on-chip CG kernel on several workloads: runs of 40
# and 120 iterations, so that per-iteration counters are differences / 80 (scripts/pmc_onchip_compare.py prints them side by
# side).  Round 4: what an iteration on the frontal (gmsh-type) mesh pays over the structured one.
#   bash scripts/pmc_onchip_compare.sh gpurun_out/pmc_cmp "frontal1m hole1m"
set -e
OUT=${1:-gpurun_out/pmc_cmp}
WLS=${2:-"frontal1m hole1m"}
ROOT=$PWD
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
i=0
for wl in $WLS; do
  for iters in 40 120; do
    for group in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES" \
                 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_ADDR_CONFLICT" \
                 "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
        i=$((i + 1))
        d="${wl}_v2_it${iters}_pass$i"
        echo "$d: $group"
        timeout -k 10 240 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$ROOT/$OUT/$d" -- \
            python3 "$ROOT/scripts/prof_iter.py" --workload $wl --cg-variant 2 --iters $iters \
            > "$ROOT/$OUT/$d.log" 2>&1 || { echo "$d failed"; tail -5 "$ROOT/$OUT/$d.log"; exit 1; }
    done
  done
done
echo done
