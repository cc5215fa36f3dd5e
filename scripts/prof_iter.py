"""Profiling driver: a capped number of CG iterations on the bench workload (few launches, for rocprofv3 --pmc)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from magnetite_amd import Context, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="hole1m")
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--tile", type=int, default=512)
ap.add_argument("--variant", type=int, default=0, help="op_variant")
ap.add_argument("--cg-variant", type=int, default=2)
a = ap.parse_args()
prob, desc = bench.build_problem(a.workload, 1)
with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8, max_iter=a.iters, tile_nodes=a.tile, use_graph=0,
             check_every=a.iters + (a.iters & 1), op_variant=a.variant, cg_variant=a.cg_variant) as ctx:
    ctx.upload_problem(prob)
    ctx.run(allow_not_converged=True)
    print(desc, ctx.stats())
