"""One-off randomized stress run: random Delaunay meshes x tile sizes x CG/operator variants against the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from magnetite_amd import Context
from test_gpu_parity import _random_delaunay_problem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
worst, bad = 0.0, []
t0 = time.time()
for seed in range(100, 100 + n):
    p = _random_delaunay_problem(seed)
    ref = oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                     p.part_thickness, path="sparse")
    kw = dict(tile_nodes=(0, 256, 512, 1024)[seed % 4], cg_variant=(1, 0)[(seed // 4) % 2], op_variant=(0, 0, 1)[seed % 3],
              assemble_csr=(1, 0)[(seed // 2) % 2])
    with Context(device=0, **kw) as c:
        out = c.solve(p)
    err = np.linalg.norm(out["u"] - ref["u"]) / np.linalg.norm(ref["u"])
    worst = max(worst, err)
    if not (out["converged"] == 1 and err <= 1e-8):
        bad.append((seed, kw, err, out["iterations"], ref["iterations"]))
    if seed % 25 == 0:
        print(f"seed {seed} E={p.mesh.num_elements} {kw} err {err:.2e} iters {out['iterations']}/{ref['iterations']}", flush=True)
print(f"{n} problems in {time.time() - t0:.0f}s, worst rel-L2 {worst:.3e}, failures: {bad}")
sys.exit(1 if bad else 0)
