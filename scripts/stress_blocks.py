"""One-off randomized parity run of the on-chip CG kernel's edge-block instantiation: random structured meshes (plates with
rectangles of removed cells, some also with scattered single cells -- nodes with two fans, which must send the mesh to the
triangle walk), random tile size, random boundary configuration, perturbed coordinates, shuffled numbering; every solution
against the CPU oracle.  Prints a summary; exit code 1 on any miss.

    python scripts/stress_blocks.py [n=60]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (the checker)
from magnetite_amd import Context, meshgen  # noqa: E402

os.environ["MAG_TUNE_PERSIST_MIN_K"] = "1"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad, blocks, walks = 0, 0, 0
for seed in range(n):
    rng = np.random.default_rng(1000 + seed)
    nx = int(rng.integers(24, 140))
    xy, tri, cx, cy = meshgen._grid(nx, nx, 1.0, 1.0)
    keep = np.ones(cx.shape[0], dtype=bool)
    for _ in range(int(rng.integers(0, 4))):
        x0, y0 = rng.uniform(0.1, 0.7, 2)
        w, h = rng.uniform(0.05, 0.25, 2)
        keep &= ~((cx > x0) & (cx < x0 + w) & (cy > y0) & (cy < y0 + h))
    if seed % 3 == 2:
        inner = (cx > 0.1) & (cx < 0.9) & (cy > 0.1) & (cy < 0.9)
        keep &= ~(inner & (rng.uniform(size=cx.shape[0]) < 0.01))
    mesh = meshgen._compact(xy, tri, keep, f"stress_{seed}")
    if seed % 2:
        mesh = meshgen.perturb(mesh, float(rng.uniform(0.05, 0.25)), seed)
    if seed % 5 == 0:
        mesh = meshgen.clockwise(mesh)
    mesh = meshgen.shuffle(mesh, seed)
    p = (meshgen.config_fixed_left_pull_right if seed % 2 else meshgen.config_fixed_left_point_load)(mesh)
    ref = oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                     p.part_thickness, path="sparse")
    with Context(device=0, tile_nodes=int(rng.choice([256, 512]))) as c:
        out = c.solve(p)
        st = c.stats()
    err = float(np.linalg.norm(out["u"] - ref["u"]) / np.linalg.norm(ref["u"]))
    ok = st["cg_kernel"] == 2 and out["converged"] == 1 and err <= 1e-8 and abs(out["iterations"] - ref["iterations"]) <= max(3, ref["iterations"] // 50)
    blocks += int(st["edge_blocks"] == 1)
    walks += int(st["edge_blocks"] == 0)
    if not ok:
        bad += 1
        print(f"MISS seed {seed}: nx {nx} kernel {st['cg_kernel']} blocks {st['edge_blocks']} err {err:.2e} "
              f"iterations {out['iterations']} / {ref['iterations']}", flush=True)
print(f"{n} problems: {blocks} on edge blocks, {walks} on the triangle walk, {bad} misses")
sys.exit(1 if bad else 0)
