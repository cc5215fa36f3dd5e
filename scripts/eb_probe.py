"""Edge-block instantiation of the on-chip CG kernel against its triangle-walk instantiation, in ONE process: the same
library, MAG_TUNE_PERSIST_TRIANGLES=1 selects the triangle walk.  Prints us per CG iteration, iteration counts, which
instantiation ran, and the relative difference of the two solutions.

    python scripts/eb_probe.py [workload ...]        (default: hole1m plate100k)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from magnetite_amd import Context, _lib  # noqa: E402

for wl in (sys.argv[1:] or ["hole1m", "plate100k"]):
    prob, desc = bench.build_problem(wl, 1)
    out = {}
    with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8, cg_variant=2) as c:
        c.upload_problem(prob)
        for mode in ("blocks", "triangles", "blocks", "triangles"):
            if mode == "triangles":
                os.environ["MAG_TUNE_PERSIST_TRIANGLES"] = "1"
            else:
                os.environ.pop("MAG_TUNE_PERSIST_TRIANGLES", None)
            ts = []
            for _ in range(3):
                c.run()
                st = c.stats()
                ts.append(st["ms_cg"] * 1e3 / max(1, st["iterations"]))
            u = c.download()[0]
            out.setdefault(mode, []).append({"us_per_iteration": round(min(ts), 3), "iterations": st["iterations"],
                                             "cg_kernel": st["cg_kernel"], "edge_blocks": st["edge_blocks"],
                                             "ms_cg": round(st["ms_cg"], 3), "ms_total": round(st["ms_total"], 3)})
            out[mode + "_u"] = u
    ub, ut = out.pop("blocks_u"), out.pop("triangles_u")
    out["rel_l2_blocks_vs_triangles"] = float(np.linalg.norm(ub - ut) / np.linalg.norm(ut))
    print(json.dumps({"workload": wl, **out}), flush=True)
