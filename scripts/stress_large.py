"""One-off: a few LARGE unstructured (Delaunay) problems -- hundreds of tiles with irregular halos -- vs the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from magnetite_amd import Context, meshgen
from scipy.spatial import Delaunay

for seed, k in ((1, 150), (2, 260)):
    rng = np.random.default_rng(seed)
    lx, ly = 2.0, 1.0
    # jittered lattice (2k x k points, interior ones moved by up to 0.35 h): Delaunay then gives a well-shaped but
    # genuinely unstructured mesh (valence 4..9, no row-major order after shuffle)
    X, Y = np.meshgrid(np.linspace(0, lx, 2 * k + 1), np.linspace(0, ly, k + 1))
    pts = np.stack([X.reshape(-1), Y.reshape(-1)], 1)
    h = ly / k
    inner = (pts[:, 0] > 1e-9) & (pts[:, 0] < lx - 1e-9) & (pts[:, 1] > 1e-9) & (pts[:, 1] < ly - 1e-9)
    pts[inner] += rng.uniform(-0.35 * h, 0.35 * h, size=(inner.sum(), 2))
    n = len(pts)
    tri = Delaunay(pts).simplices
    a = pts[tri]
    area = 0.5 * ((a[:, 1, 0] - a[:, 0, 0]) * (a[:, 2, 1] - a[:, 0, 1]) - (a[:, 2, 0] - a[:, 0, 0]) * (a[:, 1, 1] - a[:, 0, 1]))
    keep = np.abs(area) > 1e-12
    tri, area = tri[keep], area[keep]
    tri[area < 0] = tri[area < 0][:, ::-1]
    m = meshgen.shuffle(meshgen.Mesh(pts, tri.astype(np.int32)), seed)
    p = meshgen.config_fixed_left_pull_right(m)
    t0 = time.time()
    ref = oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio, p.part_thickness, path="sparse")
    t1 = time.time()
    for kw in (dict(), dict(tile_nodes=256), dict(cg_variant=0), dict(op_variant=1)):
        with Context(device=0, **kw) as c:
            out = c.solve(p)
        err = np.linalg.norm(out["u"] - ref["u"]) / np.linalg.norm(ref["u"])
        print(f"n={n} E={m.num_elements} {kw} tiles {out['num_tiles']} max_halo {out['max_tile_halo']} lds {out['lds_operator']} "
              f"iters {out['iterations']}/{ref['iterations']} err {err:.2e} cg_ms {out['ms_cg']:.1f} (oracle {t1 - t0:.1f}s)", flush=True)
        assert out["converged"] == 1 and err <= 1e-8
print("OK", flush=True)
