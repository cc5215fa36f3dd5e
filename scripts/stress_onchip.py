"""One-off randomized stress run of the on-chip CG: random Delaunay meshes (512-node tiles, one or more workgroups)
against the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MAG_TUNE_PERSIST_MIN_K"] = "1"
import oracle
from magnetite_amd import Context
from test_gpu_parity import _random_delaunay_problem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
worst, bad, kernels = 0.0, [], {}
t0 = time.time()
for seed in range(500, 500 + n):
    p = _random_delaunay_problem(seed)
    ref = oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                     p.part_thickness, path="sparse")
    with Context(device=0, tile_nodes=512, assemble_csr=(1, 0)[seed % 2]) as c:
        out = c.solve(p)
        k = c.stats()["cg_kernel"]
    kernels[k] = kernels.get(k, 0) + 1
    err = np.linalg.norm(out["u"] - ref["u"]) / np.linalg.norm(ref["u"])
    worst = max(worst, err)
    if not (out["converged"] == 1 and err <= 1e-8 and abs(out["iterations"] - ref["iterations"]) <= max(5, ref["iterations"] // 20)):
        bad.append((seed, err, out["iterations"], ref["iterations"], k))
print(f"{n} problems in {time.time() - t0:.0f}s, kernels used {kernels}, worst rel-L2 {worst:.3e}, failures: {bad}")
sys.exit(1 if bad else 0)
