#!/bin/bash
# Pure kernel time of the numeric assembly (rocprofv3 --kernel-trace --stats, no counters) for a few settings:
#   bash scripts/asm_kernel_time.sh [workload] [out-dir]      prints  <setting>: <kernel> calls avg-ns
WL=${1:-hole1m}
OUT=${2:-gpurun_out/asm_kt}
ROOT=$PWD
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
for setting in "default" "MAG_TUNE_ASM_LANES=8" "MAG_TUNE_ASM_SEGS=1" "MAG_TUNE_ASM_SEGS=2"; do
    d="$ROOT/$OUT/$(echo "$WL-$setting" | tr '=' '_')"
    rm -rf "$d"
    if [ "$setting" = default ]; then
        timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- \
            python3 "$ROOT/scripts/prof_iter.py" --workload "$WL" --cg-variant 1 --iters 2 > "$d.log" 2>&1 || { echo "$setting failed"; exit 1; }
    else
        export "$setting"
        timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- \
            python3 "$ROOT/scripts/prof_iter.py" --workload "$WL" --cg-variant 1 --iters 2 > "$d.log" 2>&1 || { echo "$setting failed"; exit 1; }
        unset "${setting%%=*}"
    fi
    f=$(find "$d" -name "*kernel_stats.csv" | head -1)
    echo "$setting: $(grep -E 'k_assemble_fan|k_fill_ell16' "$f" | awk -F'",' '{split($1,a,"("); n=split($2,b,","); print a[1] " calls=" b[1] " avg_ns=" b[3]}' | tr '\n' ' ')"
done
