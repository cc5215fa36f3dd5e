"""One-off: displacement parity GPU vs the CPU oracle on the FULL 1M-triangle benchmark mesh (the routine tests check
full-size meshes through size-independent properties only, because the oracle needs about a minute here).
Both solvers run the same stop rule; two tolerances, to separate stop-rule looseness from arithmetic differences."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

import oracle  # noqa: E402
from magnetite_amd import Context, _lib, meshgen  # noqa: E402

p = meshgen.baseline_problem("hole1m")
for tol in (1e-8, 1e-11):
    with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=tol) as c:
        out = c.solve(p)
    t0 = time.time()
    ref = oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                     p.part_thickness, path="sparse", stop_mode=oracle.STOP_REL, tol=tol)
    dt = time.time() - t0
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    print(json.dumps({"elements": p.mesh.num_elements, "rel_tol": tol, "gpu_iterations": int(out["iterations"]),
                      "oracle_iterations": int(ref["iterations"]), "rel_l2_u": rel(out["u"], ref["u"]),
                      "rel_l2_f": rel(out["f"], ref["f"]), "rel_l2_stress": rel(out["stress"], ref["stress"]),
                      "oracle_seconds": round(dt, 1)}), flush=True)
