"""Iteration-kernel time vs mesh size and tile size (validates the automatic tile choice of ensure_order)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from magnetite_amd import Context, _lib, meshgen  # noqa: E402

for n in (160, 224, 320, 400, 512, 620, 734, 1000):
    p = meshgen.config_fixed_left_pull_right(meshgen.plate_with_holes(n))
    row = {"cells": n, "elements": p.mesh.num_elements, "nodes": p.mesh.num_nodes}
    for tile in (256, 512, 1024):
        with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8, max_iter=200, tile_nodes=tile) as c:
            c.upload_problem(p)
            c.run(allow_not_converged=True)
            row[f"us_tile{tile}"] = round(c.time_operator(200) * 1e3, 2)
    row["auto"] = 512 if p.mesh.num_nodes >= 512 * 512 else 256
    print(json.dumps(row), flush=True)
