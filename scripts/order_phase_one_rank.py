"""What the ordering phase costs ONE rank of eight that has its GPU to itself: this process is rank 3 of a communicator of 8
whose all-reduce callback returns at once (the other ranks' contributions taken as zero), BASELINE config 5's geometry at a
chosen size, solves capped at two CG iterations (their numbers mean nothing: only the phases before the CG are looked at).
With the sharded phase (default) and with MAG_TUNE_SHARD_ORDER=0 (the whole mesh's tables, as every rank built them before).
    python scripts/order_phase_one_rank.py [triangles=8e6] [rank=3] [ranks=8] [sharded|replicated]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magnetite_amd import Context, meshgen  # noqa: E402

tri = float(sys.argv[1]) if len(sys.argv) > 1 else 8e6
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ranks = int(sys.argv[3]) if len(sys.argv) > 3 else 8
n = meshgen.grid_for_triangles(tri, np.pi * 0.25 ** 2)
prob = meshgen.config_fixed_left_pull_right(meshgen.multi_hole(n, 4, 0.25))
out = {"triangles": prob.mesh.num_elements, "nodes": prob.mesh.num_nodes, "rank": rank, "ranks": ranks}
for mode in ([a for a in sys.argv[4:] if a in ("sharded", "replicated")] or ["sharded", "replicated"]):
    if mode == "replicated":
        os.environ["MAG_TUNE_SHARD_ORDER"] = "0"
    else:
        os.environ.pop("MAG_TUNE_SHARD_ORDER", None)
    with Context(device=0, cg_variant=1, max_iter=2) as c:
        c.init_callback(lambda a: None, rank, ranks)
        c.upload_problem(prob)
        rows = []
        for _ in range(4):
            c.run()
            st = c.stats()
            rows.append({k: round(float(st[k]), 3) for k in ("ms_order", "ms_csr_symbolic", "ms_assemble", "ms_bc")})
        out[mode] = {"runs": rows[1:], "ell_entries": int(st["ell_entries"]), "halo_nodes": int(st["halo_nodes"]), "nnz": int(st["nnz"])}
print(json.dumps(out))
