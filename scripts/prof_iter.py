"""Profiling driver: a capped number of CG iterations on a bench workload (few launches, for rocprofv3 --pmc), then a
few launches of the plain SpMV kernel; --no-solve only runs the symbolic phase and launches the two streaming kernels
(the 16M-triangle mesh needs 20 000 iterations to converge: the counters are per launch anyway)."""
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
ap.add_argument("--no-solve", action="store_true")
ap.add_argument("--kernel-reps", type=int, default=20)
a = ap.parse_args()
prob, desc = bench.build_problem(a.workload, 1)
with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8, max_iter=a.iters, tile_nodes=a.tile, use_graph=0,
             check_every=a.iters + (a.iters & 1), op_variant=a.variant, cg_variant=a.cg_variant) as ctx:
    ctx.upload_problem(prob)
    if a.no_solve:
        print(desc, "iteration kernel ms", ctx.time_operator(a.kernel_reps), "spmv ms", ctx.time_spmv(a.kernel_reps))
    else:
        ctx.run()
        print(desc, ctx.stats())
        if a.cg_variant != 2:
            print("spmv ms", ctx.time_spmv(a.kernel_reps))
