"""The ordering phase (Hilbert order, incidence, tile tables, halo lists, ring words) alone, on a mesh of BASELINE config 5's
geometry at a chosen size: one solve capped at a single CG iteration, so that a kernel trace of this script is a trace of
the ordering phase and the CSR pattern.  (Across ranks the phase is replicated: what one rank spends here every rank does.)
    python scripts/order_phase_probe.py [triangles=8e6]
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 <repo>/scripts/order_phase_probe.py 8e6"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magnetite_amd import Context, meshgen  # noqa: E402

tri = float(sys.argv[1]) if len(sys.argv) > 1 else 8e6
n = meshgen.grid_for_triangles(tri, np.pi * 0.25 ** 2)
prob = meshgen.config_fixed_left_pull_right(meshgen.multi_hole(n, 4, 0.25))
with Context(device=0, cg_variant=1, max_iter=1) as c:
    c.upload_problem(prob)
    rows = []
    for _ in range(3):
        c.run()
        st = c.stats()
        rows.append({k: round(st[k], 4) for k in ("ms_order", "ms_csr_symbolic", "ms_assemble", "ms_bc", "ms_total")})
print(json.dumps({"triangles": prob.mesh.num_elements, "nodes": prob.mesh.num_nodes, "runs": rows}))
