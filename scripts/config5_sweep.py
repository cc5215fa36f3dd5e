"""BASELINE config 5 on ONE GPU: 16M-triangle multi-hole plate, fp64 vs fp32 CG, relative tolerance sweep.
Reports iterations, CG time and the achieved relative L2 distance to the fp64 round-off solution."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magnetite_amd import Context, _lib, meshgen

which = sys.argv[1] if len(sys.argv) > 1 else "multihole16m"
prob = meshgen.baseline_problem(which)
with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-13, max_iter=200000) as c:
    ref = c.solve(prob, allow_not_converged=True)
rows = [dict(workload=which, elements=prob.mesh.num_elements, reference="fp64 tol 1e-13", iterations=ref["iterations"],
             final_rel_residual=ref["final_cost"] / ref["rhs_norm"])]
for prec in (0, 1):
    for tol in (1e-4, 1e-6, 1e-8, 1e-10):
        with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=tol, precision=prec, max_iter=60000) as c:
            c.solve(prob, allow_not_converged=True)            # warm-up (allocations, graph)
            out = c.solve(prob, allow_not_converged=True)
        err = float(np.linalg.norm(out["u"] - ref["u"]) / np.linalg.norm(ref["u"]))
        rows.append(dict(precision="fp32" if prec else "fp64", tol=tol, converged=int(out["converged"]),
                         iterations=int(out["iterations"]), cg_ms=out["ms_cg"], us_per_iter=out["ms_cg"] * 1e3 / max(out["iterations"], 1),
                         rel_l2_vs_fp64=err))
        print(rows[-1], flush=True)
print(json.dumps(rows))
