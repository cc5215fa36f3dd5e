"""Small meshes (the size of the reference's own examples: 1k-30k triangles): the library's default since round 4 -- the on-chip
kernel for every mesh one GPU can hold -- against round 3's choice below 32768 nodes (cg_variant 1: 256-node tiles, streaming
kernels replayed from a hipGraph), reference stop rule.  First solve of a fresh context and the best of five repeats.
    python scripts/small_mesh_probe.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magnetite_amd import Context, meshgen  # noqa: E402

for n in (24, 48, 80, 120):
    prob = meshgen.config_fixed_left_pull_right(meshgen.shuffle(meshgen.plate_with_holes(n), 3))
    row = {"triangles": prob.mesh.num_elements, "nodes": prob.mesh.num_nodes}
    for mode in ("streamed", "default"):
        kw = {"cg_variant": 1} if mode == "streamed" else {}
        with Context(device=0, **kw) as c:
            c.upload_problem(prob)
            t0 = time.perf_counter()
            c.run()
            first_wall = (time.perf_counter() - t0) * 1e3
            st = c.stats()
            first = (st["ms_total"], st["ms_cg"])
            best = (1e9, 1e9)
            for _ in range(5):
                c.run()
                st = c.stats()
                best = min(best, (st["ms_total"], st["ms_cg"]))
        row[mode] = {"cg_kernel": st["cg_kernel"], "iterations": st["iterations"], "first_solve_wall_ms": round(first_wall, 3),
                     "first_ms_total": round(first[0], 3), "best_ms_total": round(best[0], 3), "best_ms_cg": round(best[1], 3),
                     "us_per_iteration": round(best[1] * 1e3 / max(1, st["iterations"]), 3)}
    print(json.dumps(row), flush=True)
