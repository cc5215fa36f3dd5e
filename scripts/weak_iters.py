"""Iteration counts of the weak-scaling workloads (global meshes of bench.py --gpus N), solved on ONE GPU,
for the two right-edge conditions: ux=delta with fy=0 (free to contract) and ux=delta with uy=0 (gripped)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from magnetite_amd import Context, _lib, meshgen
for grip in (0, 1):
    for k in (1, 2, 4, 8):
        prob, desc = bench.build_problem("hole1m", k)
        if grip:
            m = prob.mesh
            eps = 1e-9 * max(np.ptp(m.xy[:, 0]), np.ptp(m.xy[:, 1]))
            rules = [meshgen.BoundaryRule("restraint", x_max=m.xy[:, 0].min() + eps, ux=0.0, uy=0.0),
                     meshgen.BoundaryRule("load", x_min=m.xy[:, 0].max() - eps, ux=1e-3, uy=0.0)]
            prob = meshgen.apply_boundary_rules(m, rules)
        with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8) as c:
            out = c.solve(prob)
        print("grip" if grip else "free", k, desc, "E", prob.mesh.num_elements, "iters", out["iterations"],
              "cg_ms %.1f" % out["ms_cg"], flush=True)
