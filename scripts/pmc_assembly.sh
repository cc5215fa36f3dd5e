#!/bin/bash
# rocprofv3 --pmc passes over ONE launch each of the two numeric-assembly kernels on the 1M-triangle mesh
# (k_assemble_ctile: fed from the CG tiles, the default; k_assemble_tiles: round 2's, MAG_TUNE_ASSEMBLY=tiles).
#   bash scripts/pmc_assembly.sh gpurun_out/pmc_asm      then   python scripts/pmc_assembly_summarize.py gpurun_out/pmc_asm
set -e
OUT=${1:-gpurun_out/pmc_asm}
ROOT=$PWD
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
i=0
for how in ${HOWS:-ctile tiles}; do
    for group in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
                 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
                 "FETCH_SIZE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
                 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
        i=$((i + 1))
        d="asm_${how}_pass$i"
        echo "$d: $group"
        MAG_TUNE_ASSEMBLY=$how timeout -k 10 240 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$ROOT/$OUT/$d" -- \
            python3 "$ROOT/scripts/prof_iter.py" --workload hole1m --cg-variant 1 --iters 2 \
            > "$ROOT/$OUT/$d.log" 2>&1 || { echo "$d failed"; tail -5 "$ROOT/$OUT/$d.log"; exit 1; }
    done
done
echo done
