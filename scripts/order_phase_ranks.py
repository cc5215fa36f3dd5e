"""The ordering phase across ranks: eight ranks as eight threads of this process on ONE GPU (tests/multirank_threads_impl.py's
harness), config 5's geometry at a chosen size, a solve capped at a few iterations; prints every rank's ms_order with the
sharded phase (default) and with MAG_TUNE_SHARD_ORDER=0 (the whole mesh's tables on every rank, as until round 4).  The eight
ranks time-share the GPU, so a rank's figure is about eight times what it would be alone on its own device -- in both modes.
What the phase costs a rank that has its GPU to itself is the kernel time of ONE rank's stream: trace one mode at a time --
    cd /tmp && GPU_MAX_HW_QUEUES=16 rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 <repo>/scripts/order_phase_ranks.py 8e6 sharded
and divide the symbolic kernels' total by ranks x solves (scripts/order_phase_ranks_summarize.py).
    GPU_MAX_HW_QUEUES=16 python scripts/order_phase_ranks.py [triangles=8e6] [sharded|replicated]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import multirank_threads_impl as mt  # noqa: E402
from magnetite_amd import meshgen  # noqa: E402

tri = float(sys.argv[1]) if len(sys.argv) > 1 else 8e6
n = meshgen.grid_for_triangles(tri, np.pi * 0.25 ** 2)
prob = meshgen.config_fixed_left_pull_right(meshgen.multi_hole(n, 4, 0.25))
out = {"triangles": prob.mesh.num_elements, "nodes": prob.mesh.num_nodes, "ranks": mt.R}
modes = [a for a in sys.argv[2:] if a in ("sharded", "replicated")] or ["sharded", "replicated"]  # (one mode: for a kernel trace)
for mode in modes:
    if mode == "replicated":
        os.environ["MAG_TUNE_SHARD_ORDER"] = "0"
    else:
        os.environ.pop("MAG_TUNE_SHARD_ORDER", None)
    res = mt.run_ranks(prob, inboxes=True, inbox_bytes=32 << 20, solves=3, cg_variant=1, max_iter=4)
    rows = []
    for outs in res:
        o = outs[-1]
        rows.append({k: round(float(o[k]), 3) for k in ("ms_order", "ms_csr_symbolic", "ms_assemble", "ms_bc")} |
                    {k: int(o[k]) for k in ("ell_entries", "halo_nodes", "nnz")})
    out[mode] = {"ms_order_mean": round(float(np.mean([r["ms_order"] for r in rows])), 3),
                 "ms_order_over_ranks": round(float(np.mean([r["ms_order"] for r in rows])) / mt.R, 3), "per_rank": rows}
print(json.dumps(out))
