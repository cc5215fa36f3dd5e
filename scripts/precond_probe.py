"""Opt-in preconditioner (mag_options.preconditioner) on one MI355X: iterations, solve time and iteration-kernel time
for plain CG / Jacobi / block-Jacobi on the 1M-triangle benchmark mesh, uniform and distorted (interior nodes moved by
up to 0.2 cells per coordinate: every triangle keeps a positive area; 0.3 can invert elements, K is then indefinite
and CG needs ~100x the iterations), relative stop 1e-8.  Output: one JSON line per case (copied into profiles/ when refreshed)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  first: shares its HIP runtime with the library

from magnetite_amd import Context, _lib, meshgen  # noqa: E402

n = meshgen.grid_for_triangles(1e6, 3.141592653589793 * 0.15 ** 2)
cases = {"uniform": meshgen.plate_with_holes(n), "distorted0.2": meshgen.perturb(meshgen.plate_with_holes(n), 0.2, 1)}
for name, mesh in cases.items():
    p = meshgen.config_fixed_left_pull_right(mesh)
    tri = mesh.xy[mesh.conn]
    area = 0.5 * ((tri[:, 1, 0] - tri[:, 0, 0]) * (tri[:, 2, 1] - tri[:, 0, 1])
                  - (tri[:, 2, 0] - tri[:, 0, 0]) * (tri[:, 1, 1] - tri[:, 0, 1]))
    print(json.dumps({"mesh": name, "min_area_over_mean": float(area.min() / area.mean())}), flush=True)
    for kind in (0, 1, 2):
        with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8, preconditioner=kind) as c:
            c.upload_problem(p)
            c.run()
            t0 = time.perf_counter()
            c.run()
            ms = (time.perf_counter() - t0) * 1e3
            st = c.stats()
            us = c.time_operator(300) * 1e3
        print(json.dumps({"mesh": name, "elements": mesh.num_elements, "preconditioner": kind,
                          "iterations": int(st["iterations"]), "ms_solve": round(ms, 2), "ms_cg": round(st["ms_cg"], 2),
                          "us_per_iteration_kernel": round(us, 2)}), flush=True)
