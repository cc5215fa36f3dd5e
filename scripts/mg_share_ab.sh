#!/bin/bash
# Two ranks sharing ONE GPU at the real per-workgroup load (four 512-node tiles per workgroup: MAG_TUNE_PERSIST_K=4 makes
# each rank's grid 123 workgroups, both grids co-resident): on-chip multi-GPU kernels through same-device inboxes, one
# bench line per library given (MAG_LIB_PATH).  A rehearsal of the kernel side of N > 1; the xGMI hop is not in it.
#   bash scripts/mg_share_ab.sh lib_a.so lib_b.so ...
export MAG_TUNE_PERSIST_K=4
for round in 1 2; do
for lib in "$@"; do
    MAG_LIB_PATH=$PWD/$lib timeout -k 10 300 python bench.py --gpus 2 --share-gpu --partition strong --workload hole1m \
        --exchange inboxes --no-cpu-baseline --no-hbm-resident --steps 3 --warmup 1 2> /dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('$lib', 'kernel', d['config']['cg_kernel'], d['cg_iterations'], round(d['roofline']['us_per_iteration'], 3), 'us/it', d.get('fallback'))
"
done
done
