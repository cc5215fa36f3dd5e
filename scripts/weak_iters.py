"""Iteration counts of the weak-scaling workloads (global meshes of bench.py --gpus N), solved on ONE GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from magnetite_amd import Context, _lib
for k in (1, 2, 4, 8):
    prob, desc = bench.build_problem("hole1m", k)
    with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8) as c:
        out = c.solve(prob)
    print(k, desc, "E", prob.mesh.num_elements, "iters", out["iterations"], "cg_ms %.1f" % out["ms_cg"],
          "setup_ms %.1f" % (out["ms_total"] - out["ms_cg"]), "tiles", out["num_tiles"], flush=True)
