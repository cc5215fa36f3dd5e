"""CPU: the oracle's two paths against each other, against an independent numpy twin, against an analytic
patch test and against the committed golden fixtures."""
import os

import numpy as np
import pytest

import numpy_twin as twin
import oracle
from magnetite_amd import meshgen

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run(p, path, **kw):
    return oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                      p.part_thickness, path=path, **kw)


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


CASES = {
    "plate": lambda: meshgen.config_fixed_left_pull_right(meshgen.plate(7, 5, 1.4, 1.0)),
    "shuffled_load": lambda: meshgen.config_fixed_left_point_load(meshgen.shuffle(meshgen.plate(9), 1)),
    "hole_perturbed": lambda: meshgen.config_fixed_left_pull_right(
        meshgen.shuffle(meshgen.perturb(meshgen.plate_with_holes(14), 0.2), 2)),
    "clockwise": lambda: meshgen.config_fixed_left_pull_right(meshgen.clockwise(meshgen.plate(8))),
}


@pytest.mark.parametrize("name", list(CASES))
def test_dense_and_sparse_paths_are_bit_identical(built, name):
    """The sparse restatement must reproduce the reference-faithful O(n^2) path exactly: same K_e, same '+=' order."""
    p = CASES[name]()
    a, b = run(p, "dense", hist_len=50), run(p, "sparse", hist_len=50)
    for k in ("u", "f", "stress", "history"):
        assert np.array_equal(a[k], b[k]), k
    assert a["iterations"] == b["iterations"] and a["nnz_ff"] == b["nnz_ff"] and a["n_free"] == b["n_free"]


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_numpy_twin(built, name):
    p = CASES[name]()
    t = twin.solve(p.mesh.xy, p.mesh.conn, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                   p.part_thickness)
    # element matrices and assembled K: same formulas, different summation order
    ke = oracle.element_stiffness_all(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    for e in (0, len(ke) // 2, len(ke) - 1):
        want = twin.element_stiffness(p.mesh.xy, p.mesh.conn[e], p.poisson_ratio, p.youngs_modulus, p.part_thickness)
        assert np.abs(ke[e] - want).max() <= 1e-13 * np.abs(want).max()
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    assert np.abs(K.toarray() - t["K"]).max() <= 1e-13 * np.abs(t["K"]).max()
    dense = oracle.assemble_dense(p.mesh.num_nodes, p.conn_flat, ke)
    assert np.array_equal(dense, K.toarray())  # CSR holds exactly the dense scatter's values
    A, b = oracle.reduce_system(K, p.u_known, p.u_in, p.f_in)
    assert np.abs(b - t["b"]).max() <= 1e-12 * np.abs(t["b"]).max()
    Kff, b2 = oracle.partition_dense(dense, p.u_known, p.u_in, p.f_in)
    assert np.array_equal(b, b2)
    assert np.array_equal(oracle.sparsify_dense(Kff).toarray(), A.toarray())
    assert np.array_equal(Kff, t["K"][np.ix_(t["free"], t["free"])] * 0 + Kff)  # shape check
    # CG (to round-off: absolute 1e-4 on a 1e10-scale rhs) against the direct solve
    r = run(p, "sparse")
    assert rel(r["u"], t["u"]) <= 1e-9
    fs = np.abs(t["K"]).max() * np.abs(t["u"]).max()
    assert np.abs(r["f"] - t["f"]).max() <= 1e-9 * fs
    stable = np.abs(t["stress"]) > 1e-6 * np.abs(t["stress"]).max()
    assert np.allclose(r["stress"][stable], t["stress"][stable], rtol=1e-6)


def test_patch_test_uniform_tension(built):
    """CST reproduces a linear field exactly: left edge ux=0 (one node also uy=0), right edge ux=delta =>
    ux = delta x / L, uy = -nu delta (y - y0) / L, sigma_x = E delta / L everywhere."""
    L, H, delta = 2.0, 1.0, 1e-3
    m = meshgen.shuffle(meshgen.perturb(meshgen.plate(8, 4, L, H), 0.2), 4)
    eps = 1e-9
    rules = [meshgen.BoundaryRule("left", x_max=eps, ux=0.0, fy=0.0),
             meshgen.BoundaryRule("pin", x_max=eps, y_max=eps, ux=0.0, uy=0.0),
             meshgen.BoundaryRule("right", x_min=L - eps, ux=delta, fy=0.0)]
    p = meshgen.apply_boundary_rules(m, rules)
    r = run(p, "dense")
    ux = delta * m.xy[:, 0] / L
    uy = -p.poisson_ratio * delta * m.xy[:, 1] / L
    want = np.stack([ux, uy], axis=1).reshape(-1)
    assert np.abs(r["u"] - want).max() <= 1e-9 * delta
    assert np.allclose(r["stress"], p.youngs_modulus * delta / L, rtol=1e-8)
    # total reaction on the right edge = sigma * H * t
    right = np.where(m.xy[:, 0] > L - eps)[0]
    assert r["f"][2 * right].sum() == pytest.approx(p.youngs_modulus * delta / L * H * p.part_thickness, rel=1e-8)


def test_cg_restatement_semantics(built):
    p = CASES["plate"]()
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    A, b = oracle.reduce_system(K, p.u_known, p.u_in, p.f_in)
    x, it, cost, hist = oracle.cg(A, b, hist_len=1000)
    # stops after the FIRST iteration whose cost is <= target (solver.rs:154), absolute threshold
    assert cost <= oracle.TARGET_CG_COST and np.all(hist[:it - 1] > oracle.TARGET_CG_COST) and hist[it - 1] == cost
    # cost is the recurrence residual norm; the true residual agrees to round-off of the rhs scale
    assert np.linalg.norm(A.spmv(x) - b) <= 1e-10 * np.linalg.norm(b)
    # first iteration by hand: r0=-b, p0=b, alpha=b.b/(b.Ab), r1=r0+alpha A p0
    Ab = A.spmv(b)
    alpha = (b @ b) / (b @ Ab)
    assert hist[0] == pytest.approx(np.linalg.norm(-b + alpha * Ab), rel=1e-12)
    # r.r variant stops earlier; relative variant
    _, it_sq, cost_sq, _ = oracle.cg(A, b, stop_mode=oracle.STOP_RNORM_SQ)
    assert it_sq <= it and cost_sq <= 1e-4
    _, it_rel, cost_rel, _ = oracle.cg(A, b, stop_mode=oracle.STOP_REL, tol=1e-6)
    assert it_rel < it and cost_rel <= 1e-6 * np.linalg.norm(b)
    # max_iters (solver.rs:153) returns the best iterate so far
    xb, itb, costb, hb = oracle.cg(A, b, max_iter=5, hist_len=5)
    assert itb == 5 and costb == hb.min()
    # zero right-hand side: documented deviation, x = 0 in 0 iterations
    x0, it0, c0, _ = oracle.cg(A, np.zeros_like(b))
    assert it0 == 0 and c0 == 0.0 and not x0.any()


def test_parallel_cg_variant_agrees(built):
    p = CASES["hole_perturbed"]()
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    A, b = oracle.reduce_system(K, p.u_known, p.u_in, p.f_in)
    x, it, _, _ = oracle.cg(A, b)
    xp, itp, cost = oracle.cg_parallel(A, b, threads=4)
    assert abs(it - itp) <= 3 and cost <= 1e-4 and rel(xp, x) <= 1e-9


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("threads", [1, 3])
def test_openmp_assembly_is_bit_identical_to_the_serial_restatement(built, name, threads):
    """the all-cores CPU baseline (bench.py cpu_baseline.all_cores): rows of K are independent, each row adds its
    elements in ascending element order like solver.rs:299-325 -- same pattern, same bits, for K, K_ff and b"""
    p = CASES[name]()
    args = (p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    K, Kp = oracle.assemble_sparse(*args), oracle.assemble_sparse(*args, threads=threads)
    A, b = oracle.reduce_system(K, p.u_known, p.u_in, p.f_in)
    Ap, bp = oracle.reduce_system(Kp, p.u_known, p.u_in, p.f_in, threads=threads)
    for x, y in ((K, Kp), (A, Ap)):
        assert x.n == y.n and x.nnz == y.nnz
        assert np.array_equal(x.rowptr, y.rowptr) and np.array_equal(x.col, y.col) and np.array_equal(x.val, y.val)
    assert np.array_equal(b, bp)


def test_clockwise_mesh_same_displacements_for_displacement_loading(built):
    """K -> -K and b -> -b when every element is reversed and all loads are prescribed displacements."""
    a = run(meshgen.config_fixed_left_pull_right(meshgen.plate(8)), "dense")
    b = run(CASES["clockwise"](), "dense")
    assert rel(b["u"], a["u"]) <= 1e-10


@pytest.mark.parametrize("name", ["tensile", "plate"])
def test_oracle_reproduces_golden_fixture(built, name):
    """tests/golden/*.npz were written by tests/golden/make_fixtures.py from this oracle: bit-for-bit or the
    oracle has drifted."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    E, nu, t = g["material"]
    r = oracle.run(g["xy"].reshape(-1), g["conn"].reshape(-1), g["u_known"], g["u_in"], g["f_in"], E, nu, t,
                   path="sparse", hist_len=32)
    for k in ("u", "f", "stress", "history"):
        assert np.array_equal(r[k], g[k]), k
    assert r["iterations"] == int(g["iterations"]) and r["nnz_ff"] == int(g["nnz_ff"])


def test_tensile_example_physics(built):
    """examples/tensile-example: right grip moves by exactly ux=3 (input.json:27-28), the specimen necks
    symmetrically (media/tensilve-results.png), left grip stays put."""
    g = np.load(os.path.join(GOLD, "tensile.npz"))
    xy, u = g["xy"], g["u"].reshape(-1, 2)
    assert np.all(u[xy[:, 0] > 10][:, 0] == 3.0)
    assert np.all(u[xy[:, 0] < -10] == 0.0)
    gauge = np.abs(xy[:, 0]) < 3
    top, bot = gauge & (xy[:, 1] > 2.0), gauge & (xy[:, 1] < -2.0)
    assert u[top][:, 1].mean() < 0 < u[bot][:, 1].mean()  # lateral contraction
    assert abs(u[top][:, 1].mean() + u[bot][:, 1].mean()) < 0.05 * abs(u[top][:, 1].mean())
    # every element was reversed by check_ccw (area < 1.0), so K is negative definite and CG still converged
    a = xy[g["conn"]]
    area = 0.5 * ((a[:, 1, 0] - a[:, 0, 0]) * (a[:, 2, 1] - a[:, 0, 1]) - (a[:, 2, 0] - a[:, 0, 0]) * (a[:, 1, 1] - a[:, 0, 1]))
    assert np.all(area < 0) and float(g["final_cost"]) <= 1e-4


def test_oracle_pcg_is_textbook_pcg_with_fp32_node_blocks():
    """The opt-in preconditioner has no reference counterpart; its checker (orc_pcg / orc_block_jacobi) is pinned
    here against an independent numpy statement: M^-1 = fp32-rounded inverses of the 2x2 node blocks of K_ff."""
    import scipy.sparse as sp

    from magnetite_amd import meshgen
    p = meshgen.config_fixed_left_pull_right(meshgen.perturb(meshgen.plate_with_holes(20, 24, 1.0, 1.2), 0.2, 3))
    N = p.mesh.num_nodes
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    Kd = sp.csr_matrix((K.val, K.col, K.rowptr), shape=(K.n, K.n)).toarray()
    free = (p.u_known == 0)
    for kind in (1, 2):
        minv = oracle.block_jacobi(K, N, p.u_known, kind)
        assert minv.dtype == np.float32
        Minv = np.zeros((2 * N, 2 * N))
        for i in range(N):
            blk = Kd[2 * i:2 * i + 2, 2 * i:2 * i + 2].copy()
            f = free[2 * i:2 * i + 2]
            if kind == 1 or not f.all():
                inv = np.diag([1.0 / blk[0, 0] if f[0] else 0.0, 1.0 / blk[1, 1] if f[1] else 0.0])
            else:
                inv = np.linalg.inv(blk)
            assert np.allclose([inv[0, 0], inv[0, 1], inv[1, 1]], minv[i], rtol=1e-6, atol=0)
            Minv[2 * i:2 * i + 2, 2 * i:2 * i + 2] = [[minv[i, 0], minv[i, 1]], [minv[i, 1], minv[i, 2]]]
        A = Kd[np.ix_(free, free)]
        b = (p.f_in - Kd @ np.where(free, 0.0, p.u_in))[free]
        Mi = Minv[np.ix_(free, free)]
        x = np.zeros_like(b)
        r = -b
        z = Mi @ r
        d = -z
        rho = r @ z
        it = 0
        while it < 10000:
            q = A @ d
            alpha = rho / (d @ q)
            x += alpha * d
            r += alpha * q
            z = Mi @ r
            rho_n = r @ z
            d = -z + (rho_n / rho) * d
            rho = rho_n
            it += 1
            if np.sqrt(r @ r) <= 1e-4:
                break
        ref = oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                         p.part_thickness, precond=kind)
        plain = oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                           p.part_thickness)
        assert abs(ref["iterations"] - it) <= 2
        assert np.linalg.norm(ref["u"][free] - x) <= 1e-9 * np.linalg.norm(x)
        assert np.linalg.norm(ref["u"] - plain["u"]) <= 1e-8 * np.linalg.norm(plain["u"])
        assert ref["iterations"] < plain["iterations"]


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_cg_follows_the_numpy_restatement_of_argmin(built, name):
    """The CG is third-party code the reference tree does not hold (argmin 0.10): the oracle's C restatement and an
    independent numpy restatement of the published recurrences must walk the same residual history, stop in the same
    iteration and -- at an iteration cap that falls into a rising stretch -- return the same best_param."""
    p = CASES[name]()
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    A, b = oracle.reduce_system(K, p.u_known, p.u_in, p.f_in)
    Ad = A.toarray()
    x_o, it_o, cost_o, hist_o = oracle.cg(A, b, hist_len=100000)
    x_t, it_t, hist_t = twin.argmin_cg(Ad, b, oracle.TARGET_CG_COST, oracle.MAX_CG_ITER)
    # same recurrences, different summation order: the histories coincide at first and drift apart as CG amplifies the
    # rounding differences (the reference's absolute 1e-4 on a 1e7-scale right-hand side runs the iteration 12 orders of
    # magnitude down, to round-off); both end at the same solution
    n = min(len(hist_o), len(hist_t), 8)
    assert n >= 3 and np.allclose(hist_o[:n], hist_t[:n], rtol=1e-9)
    assert abs(it_o - it_t) <= max(3, it_t // 10)
    assert rel(x_o, x_t) <= 1e-9
    # the iteration cap: best_param is the lowest-cost iterate, not necessarily the last one
    cap = min(8, it_t - 1)
    x_oc, it_oc, cost_oc, hist_oc = oracle.cg(A, b, max_iter=cap, hist_len=cap)
    x_tc, it_tc, hist_tc = twin.argmin_cg(Ad, b, oracle.TARGET_CG_COST, cap)
    assert it_oc == it_tc == cap and int(np.argmin(hist_oc)) == int(np.argmin(hist_tc))
    assert abs(cost_oc - hist_tc.min()) <= 1e-9 * hist_tc.min() and rel(x_oc, x_tc) <= 1e-9
    # r.r as the cost (the other reading of argmin's `cost`): same walk, its own stop
    x_os, it_os, _, _ = oracle.cg(A, b, stop_mode=oracle.STOP_RNORM_SQ, tol=1e-2)
    x_ts, it_ts, _ = twin.argmin_cg(Ad, b, 1e-2, oracle.MAX_CG_ITER, squared=True)
    assert abs(it_os - it_ts) <= max(2, it_ts // 50) and rel(x_os, x_ts) <= 1e-8
