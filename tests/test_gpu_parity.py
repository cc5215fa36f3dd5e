"""-m gpu: parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars: bit-exact for integer/index work (CSR pattern, K_ff pattern, iteration bookkeeping) and for the
reference-order fp64 kernels (K_e, assembled K, K_ff values, b: same operations in the same order, no FMA);
<= 1e-8 relative L2 on nodal displacements (BASELINE.json north_star) for everything downstream of CG.
"""
import ctypes as C

import numpy as np
import pytest

import oracle
from magnetite_amd import Context, MagnetiteError, _lib, meshgen
from magnetite_amd._lib import (MAG_ERR_BAD_ARGS, MAG_ERR_BC_MISMATCH, MAG_OP_CSR, MAG_STOP_REL,
                                MAG_STOP_RNORM_SQ)

pytestmark = pytest.mark.gpu

TOL_U = 1e-8  # north_star: relative L2 on nodal displacements


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def oracle_run(p, **kw):
    return oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                      p.part_thickness, path="sparse", **kw)


def problems():
    yield "plate12", meshgen.config_fixed_left_pull_right(meshgen.plate(12))
    yield "plate_shuffled", meshgen.config_fixed_left_point_load(meshgen.shuffle(meshgen.plate(40, 25, 2.0), 3))
    yield "hole_perturbed", meshgen.config_fixed_left_pull_right(
        meshgen.shuffle(meshgen.perturb(meshgen.plate_with_holes(64), 0.2), 11))
    yield "clockwise", meshgen.config_fixed_left_pull_right(meshgen.clockwise(meshgen.plate(24)))
    yield "multihole", meshgen.config_fixed_left_point_load(meshgen.multi_hole(90, 3, 0.25))
    yield "one_tile_ragged", meshgen.config_fixed_left_pull_right(meshgen.plate(3, 2))


PROBLEMS = dict(problems())


@pytest.fixture(scope="module")
def ctx(built):
    with Context(device=0, history_len=64) as c:
        yield c


@pytest.mark.parametrize("name", list(PROBLEMS))
def test_element_stiffness_bit_exact(ctx, name):
    p = PROBLEMS[name]
    ctx.upload_problem(p)
    ke = ctx.element_stiffness()
    ref = oracle.element_stiffness_all(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    assert np.array_equal(ke, ref)


@pytest.mark.parametrize("name", list(PROBLEMS))
def test_assembled_csr_bit_exact(ctx, name):
    p = PROBLEMS[name]
    ctx.upload_problem(p)
    rowptr, col, val = ctx.assemble_csr()
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    assert np.array_equal(rowptr.astype(np.int64), K.rowptr)
    assert np.array_equal(col, K.col)
    assert np.array_equal(val, K.val)


@pytest.mark.parametrize("name", list(PROBLEMS))
def test_reduced_system_bit_exact(ctx, name):
    p = PROBLEMS[name]
    ctx.upload_problem(p)
    rowptr, col, val, b = ctx.reduce_system()
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    A, bo = oracle.reduce_system(K, p.u_known, p.u_in, p.f_in)
    assert np.array_equal(rowptr.astype(np.int64), A.rowptr)
    assert np.array_equal(col, A.col)
    assert np.array_equal(val, A.val)
    assert np.array_equal(b, bo)


@pytest.mark.parametrize("name", list(PROBLEMS))
def test_matrix_free_operator_matches_csr(ctx, name):
    p = PROBLEMS[name]
    ctx.upload_problem(p)
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    x = np.random.default_rng(2024).standard_normal(K.n)
    y = ctx.apply_operator(x, masked=False)
    assert rel(y, K.spmv(x)) < 1e-12
    free = (p.u_known == 0).astype(np.float64)
    ym = ctx.apply_operator(x, masked=True)
    assert rel(ym, free * K.spmv(free * x)) < 1e-12
    assert np.all(ym[p.u_known == 1] == 0.0)


@pytest.mark.parametrize("name", list(PROBLEMS))
@pytest.mark.parametrize("assemble,variant", [(1, 1), (0, 1), (1, 0)])
def test_full_solve_parity(built, name, assemble, variant):
    """variant 1: one fused launch per iteration (default); 0: two launches, argmin's recurrences to the letter."""
    p = PROBLEMS[name]
    ref = oracle_run(p, hist_len=16)
    with Context(device=0, history_len=16, assemble_csr=assemble, cg_variant=variant) as c:
        out = c.solve(p)
        hist = c.history(min(16, out["iterations"]))
    assert out["converged"] == 1
    assert rel(out["u"], ref["u"]) <= TOL_U
    # prescribed values come back untouched (solver.rs:443-454 only fills unknowns)
    k = p.u_known == 1
    assert np.array_equal(out["u"][k], p.u_in[k])
    assert np.array_equal(out["f"][~k], p.f_in[~k])
    # reactions are sums of O(|K||u|) terms that may cancel to round-off: absolute bar on that scale
    fscale = np.abs(oracle.element_stiffness_all(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus,
                                                 p.part_thickness)).max() * np.abs(ref["u"]).max()
    assert np.abs(out["f"][k] - ref["f"][k]).max() <= 1e-8 * fscale
    # first iterations follow the oracle's residual history (argmin recurrences)
    n = min(len(hist), len(ref["history"]), 8)
    assert np.allclose(hist[:n], ref["history"][:n], rtol=1e-9)
    # iteration counts agree to within round-off noise near the absolute threshold
    assert abs(out["iterations"] - ref["iterations"]) <= max(3, ref["iterations"] // 50)
    # stress: same arithmetic on u; the `< 1.0` sign rule (solver.rs:524-530) is discontinuous,
    # so the sign is compared only away from its threshold
    mag_ok = np.abs(np.abs(out["stress"]) - np.abs(ref["stress"])) <= 1e-6 * np.abs(ref["stress"]).max()
    assert mag_ok.all()
    stable = np.abs(ref["stress"]) > 1e-3 * np.abs(ref["stress"]).max()
    assert np.array_equal(np.sign(out["stress"][stable]), np.sign(ref["stress"][stable]))


def test_stop_modes_and_eager(built):
    p = PROBLEMS["hole_perturbed"]
    for kw in (dict(stop_mode=MAG_STOP_RNORM_SQ), dict(stop_mode=MAG_STOP_REL, tol=1e-10),
               dict(use_graph=0), dict(check_every=2), dict(tile_nodes=256), dict(tile_nodes=1024),
               dict(cg_variant=0, use_graph=0), dict(cg_variant=0, tile_nodes=256, stop_mode=MAG_STOP_RNORM_SQ)):
        okw = {}
        if "stop_mode" in kw:
            okw = dict(stop_mode=kw["stop_mode"], tol=kw.get("tol", 1e-4))
        ref = oracle_run(p, **okw)
        with Context(device=0, **kw) as c:
            out = c.solve(p)
        assert out["converged"] == 1, kw
        assert rel(out["u"], ref["u"]) <= (1e-6 if kw.get("stop_mode") == MAG_STOP_REL else TOL_U), kw
        assert abs(out["iterations"] - ref["iterations"]) <= max(3, ref["iterations"] // 50), kw


@pytest.mark.parametrize("name", ["plate_shuffled", "hole_perturbed", "clockwise"])
def test_csr_operator_mode_is_the_reference_iteration(built, name):
    """MAG_OP_CSR: CG on K_ff in CSR, compact ascending-DOF numbering, row sums in ascending column order --
    solver.rs:23-37 + argmin, i.e. exactly what the oracle's orc_cg does; only the dot products are summed
    in a different order."""
    p = PROBLEMS[name]
    ref = oracle_run(p, hist_len=32)
    with Context(device=0, cg_operator=MAG_OP_CSR, history_len=32) as c:
        out = c.solve(p)
        hist = c.history(min(32, out["iterations"]))
    assert out["converged"] == 1 and out["n_free"] == ref["n_free"]
    assert rel(out["u"], ref["u"]) <= TOL_U
    n = min(len(hist), len(ref["history"]))
    assert np.allclose(hist[:n], ref["history"][:n], rtol=1e-10)
    assert abs(out["iterations"] - ref["iterations"]) <= max(3, ref["iterations"] // 50)
    with Context(device=0) as c:
        mf = c.solve(p)
    assert rel(mf["u"], out["u"]) <= 1e-9 and mf["n_free"] == ref["n_free"]


def test_unreferenced_nodes_and_all_prescribed(built):
    """Nodes no element uses have empty rows of K (u stays 0 there); a BC set with no unknown is an error."""
    m = meshgen.plate(6)
    xy = np.concatenate([m.xy, [[5.0, 5.0], [6.0, 5.0]]])
    p = meshgen.config_fixed_left_pull_right(meshgen.Mesh(m.xy, m.conn))
    uk = np.concatenate([p.u_known, [0, 0, 0, 0]]).astype(np.uint8)
    ui = np.concatenate([p.u_in, np.zeros(4)])
    fi = np.concatenate([p.f_in, np.zeros(4)])
    with Context(device=0) as c:
        c.upload(xy.reshape(-1), m.conn.reshape(-1), uk, ui, fi, p.youngs_modulus, p.poisson_ratio, p.part_thickness)
        c.run()
        u, f, s = c.download()
    ref = oracle_run(p)
    assert rel(u[:-4], ref["u"]) <= TOL_U and not u[-4:].any()
    with Context(device=0) as c:
        c.upload(m.xy.reshape(-1), m.conn.reshape(-1), np.ones_like(p.u_known), p.u_in, p.f_in, 1.0, 0.3, 1.0)
        with pytest.raises(MagnetiteError) as ei:
            c.run()
        assert ei.value.code == MAG_ERR_BC_MISMATCH


def test_config5_size_multihole_16m(built):
    """BASELINE config 5 geometry (16M-triangle multi-hole plate) on one GPU: one solve, checked through the
    operator itself (K_ff u_f = b to round-off of the right-hand-side scale) and through symmetry of K."""
    p = meshgen.baseline_problem("multihole16m")
    n = 2 * p.mesh.num_nodes
    assert p.mesh.num_elements > 15_500_000
    rng = np.random.default_rng(5)
    with Context(device=0, stop_mode=MAG_STOP_REL, tol=1e-9) as c:
        out = c.solve(p)
        assert out["converged"] == 1 and out["lds_operator"] == 1
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        Ky = c.apply_operator(y)
        assert abs(x @ Ky - y @ c.apply_operator(x)) <= 1e-11 * np.linalg.norm(x) * np.linalg.norm(Ky)
        free = p.u_known == 0
        res = (c.apply_operator(out["u"]) - p.f_in)[free]
        fs = np.abs(c.apply_operator(np.where(free, 0.0, out["u"]))).max()
        assert np.linalg.norm(res) <= 1e-7 * fs * np.sqrt(free.sum())
    k = p.u_known == 1
    assert np.array_equal(out["u"][k], p.u_in[k])


def test_run_to_run_bitwise_reproducible(built):
    p = PROBLEMS["multihole"]
    with Context(device=0) as c:
        a = c.solve(p)
        b = c.solve(p)
    assert a["iterations"] == b["iterations"]
    assert np.array_equal(a["u"], b["u"]) and np.array_equal(a["f"], b["f"])


@pytest.mark.parametrize("variant", [2, 1, 0])
def test_iteration_cap_returns_best_param_like_argmin(built, variant):
    """solver.rs:149-176: argmin's MaxItersReached is a normal termination and the reference returns
    Ok(state.best_param) -- the LOWEST-cost iterate (plain CG's residual norm is not monotone), not the last one.
    mag_run therefore returns MAG_OK with converged = 0, termination = MAG_TERM_MAX_ITERS, iterations = the cap (what
    the reference's observer prints) and the iterate of best_iteration; the oracle's orc_cg keeps best_param too."""
    p = PROBLEMS["plate_shuffled"]
    cap = 50
    ref = oracle_run(p, max_iter=cap, hist_len=cap)
    kbest = int(np.argmin(ref["history"])) + 1
    assert kbest == 41 and ref["iterations"] == cap  # the cap falls into a rising stretch of the residual history
    with Context(device=0, max_iter=cap, history_len=cap, cg_variant=variant, tile_nodes=512 if variant == 2 else 0) as c:
        out = c.solve(p)  # no error
        hist = c.history(cap)
        assert c.stats()["cg_kernel"] == variant
    assert out["converged"] == 0 and out["termination"] == _lib.MAG_TERM_MAX_ITERS and out["iterations"] == cap
    assert out["best_iteration"] == kbest == int(np.argmin(hist)) + 1
    assert out["best_param_mismatch"] == 0  # the repeat up to kbest took the same kernel and hit the recorded cost bit for bit
    assert out["final_cost"] == hist[kbest - 1] and abs(out["final_cost"] - ref["final_cost"]) <= 1e-9 * ref["final_cost"]
    assert rel(out["u"], ref["u"]) <= TOL_U
    # and it IS a different vector from the last iterate
    last = oracle_run(p, max_iter=kbest)  # orc_cg stopped at kbest returns x_kbest (its best so far)
    assert rel(out["u"], last["u"]) <= TOL_U


def test_zero_rhs_returns_zero(built):
    p = meshgen.apply_boundary_rules(meshgen.plate(6), [meshgen.BoundaryRule("r", x_max=1e-9, ux=0.0, uy=0.0)])
    with Context(device=0) as c:
        out = c.solve(p)
    assert out["iterations"] == 0 and np.all(out["u"] == 0.0)


def test_bad_connectivity_is_an_error_not_a_fault(built):
    p = meshgen.config_fixed_left_pull_right(meshgen.plate(6))
    bad = p.conn_flat.copy()
    bad[5] = p.mesh.num_nodes + 7
    with Context(device=0) as c:
        c.upload(p.xy_flat, bad, p.u_known, p.u_in, p.f_in, 1.0, 0.3, 1.0)
        with pytest.raises(MagnetiteError) as ei:
            c.run()
        assert ei.value.code == MAG_ERR_BAD_ARGS


def test_reference_interface_run(built):
    """solver::run semantics: nodes/elements mutated in place, every Option filled."""
    from magnetite_amd import Element, ModelMetadata, Node, Vertex, run
    m = meshgen.plate(5)
    p = meshgen.config_fixed_left_pull_right(m)
    nodes = []
    for i in range(m.num_nodes):
        kx, ky = p.u_known[2 * i], p.u_known[2 * i + 1]
        nodes.append(Node(Vertex(*m.xy[i]), ux=p.u_in[2 * i] if kx else None, uy=p.u_in[2 * i + 1] if ky else None,
                          fx=None if kx else p.f_in[2 * i], fy=None if ky else p.f_in[2 * i + 1]))
    elements = [Element(list(map(int, t))) for t in m.conn]
    run(nodes, elements, ModelMetadata(p.youngs_modulus, p.poisson_ratio, p.part_thickness))
    ref = oracle_run(p)
    u = np.array([[n.ux, n.uy] for n in nodes]).reshape(-1)
    assert rel(u, ref["u"]) <= TOL_U
    assert all(n.fx is not None and n.fy is not None for n in nodes)
    assert all(e.stress is not None for e in elements)
    nodes[0].fx = 1.0  # both known on one DOF: the reference panics (solver.rs:431); here a Solver error
    with pytest.raises(MagnetiteError):
        run(nodes, elements, ModelMetadata(p.youngs_modulus, p.poisson_ratio, p.part_thickness))


@pytest.mark.parametrize("name", ["tensile", "plate"])
def test_against_committed_golden_fixtures(built, name):
    """tests/golden/*.npz (oracle outputs, generated by tests/golden/make_fixtures.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    E, nu, t = g["material"]
    with Context(device=0, history_len=32) as c:
        c.upload(g["xy"].reshape(-1), g["conn"].reshape(-1), g["u_known"], g["u_in"], g["f_in"], E, nu, t)
        if "ke" in g:
            assert np.array_equal(c.element_stiffness(), g["ke"])
            rowptr, col, val = c.assemble_csr()
            assert np.array_equal(rowptr, g["K_rowptr"]) and np.array_equal(col, g["K_col"])
            assert np.array_equal(val, g["K_val"])
            rp, cl, vl, b = c.reduce_system()
            assert np.array_equal(rp, g["A_rowptr"]) and np.array_equal(cl, g["A_col"])
            assert np.array_equal(vl, g["A_val"]) and np.array_equal(b, g["b"])
        c.run()
        u, f, s = c.download()
        st = c.stats()
        hist = c.history(8)
    assert rel(u, g["u"]) <= TOL_U
    assert np.allclose(hist, g["history"][:8], rtol=1e-9)
    assert abs(st["iterations"] - int(g["iterations"])) <= max(3, int(g["iterations"]) // 50)
    stable = np.abs(g["stress"]) > 1e-3 * np.abs(g["stress"]).max()
    assert np.allclose(s[stable], g["stress"][stable], rtol=1e-6)


def test_gather_fallback_operator(built):
    """op_variant=1 forces the global-gather operator (used when a tile's halo does not fit LDS)."""
    p = PROBLEMS["hole_perturbed"]
    ref = oracle_run(p)
    with Context(device=0, op_variant=1) as c:
        out = c.solve(p)
    assert out["lds_operator"] == 0 and rel(out["u"], ref["u"]) <= TOL_U
    with Context(device=0) as c:
        out2 = c.solve(p)
    assert out2["lds_operator"] == 1 and rel(out2["u"], out["u"]) <= 1e-10


def test_high_valence_fan_mesh(built):
    """A node shared by 40 elements: more incident elements than the slot words kept in registers."""
    n = 40
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    xy = np.concatenate([[[0.0, 0.0]], np.stack([np.cos(ang), np.sin(ang)], axis=1),
                         1.8 * np.stack([np.cos(ang), np.sin(ang)], axis=1)])
    tri = [[0, 1 + k, 1 + (k + 1) % n] for k in range(n)]
    tri += [[1 + k, 1 + n + k, 1 + n + (k + 1) % n] for k in range(n)]
    tri += [[1 + k, 1 + n + (k + 1) % n, 1 + (k + 1) % n] for k in range(n)]
    m = meshgen.Mesh(xy, np.array(tri, dtype=np.int32), "fan")
    p = meshgen.apply_boundary_rules(m, [meshgen.BoundaryRule("hold", x_max=-1.2, ux=0.0, uy=0.0),
                                         meshgen.BoundaryRule("pull", x_min=1.2, ux=0.01, fy=0.0)])
    ref = oracle_run(p)
    for variant in (0, 1):
        with Context(device=0, op_variant=variant, tile_nodes=256) as c:
            out = c.solve(p)
        assert rel(out["u"], ref["u"]) <= TOL_U


@pytest.mark.parametrize("valence", [6, 7, 8, 9, 14, 15, 16, 17])
def test_assembly_fast_paths_at_their_row_length_limits(built, valence, monkeypatch):
    """A hub of `valence` triangles has valence + 1 blocks in its K row.  The assembly keeps rows of up to 8 blocks in LDS
    accumulators (longer ones are finished by the whole workgroup), the pattern kernel keeps up to 16 distinct columns in
    registers (longer rows send the whole pattern to the sort-based construction): K, K_ff and b must be bit-identical to
    the oracle's on both sides of both limits, and identical between the two pattern constructions."""
    n = valence
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False) + 0.1
    xy = np.concatenate([[[0.0, 0.0]], np.stack([np.cos(ang), np.sin(ang)], axis=1),
                         2.0 * np.stack([np.cos(ang + np.pi / n), np.sin(ang + np.pi / n)], axis=1)])
    tri = [[0, 1 + k, 1 + (k + 1) % n] for k in range(n)]
    tri += [[1 + k, 1 + n + k, 1 + (k + 1) % n] for k in range(n)]
    m = meshgen.shuffle(meshgen.Mesh(xy, np.array(tri, dtype=np.int32), f"hub{n}"), 5)
    p = meshgen.apply_boundary_rules(m, [meshgen.BoundaryRule("hold", x_max=-1.5, ux=0.0, uy=0.0),
                                         meshgen.BoundaryRule("pull", x_min=1.5, ux=0.01, fy=0.0)])
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    A, bo = oracle.reduce_system(K, p.u_known, p.u_in, p.f_in)
    assert np.diff(K.rowptr).max() == 2 * (valence + 1)
    got = {}
    # the assembly fed from the CG tiles (default) at both tile sizes, round 2's kernel on caller-numbered gathers, and the
    # sort-based pattern
    for how in ("incidence", "ctile512", "tiles", "sort"):
        if how == "sort":
            monkeypatch.delenv("MAG_TUNE_ASSEMBLY")
            monkeypatch.setenv("MAG_TUNE_PATTERN_SORT", "1")
        if how == "tiles":
            monkeypatch.setenv("MAG_TUNE_ASSEMBLY", "tiles")
        with Context(device=0, tile_nodes=512 if how == "ctile512" else 0) as c:
            c.upload_problem(p)
            rowptr, col, val = c.assemble_csr()
            rp, cf, vf, b = c.reduce_system()
        assert np.array_equal(rowptr.astype(np.int64), K.rowptr) and np.array_equal(col, K.col), how
        assert np.array_equal(val, K.val), how
        assert np.array_equal(rp.astype(np.int64), A.rowptr) and np.array_equal(cf, A.col) and np.array_equal(vf, A.val)
        assert np.array_equal(b, bo)
        got[how] = val
    assert np.array_equal(got["incidence"], got["sort"]) and np.array_equal(got["incidence"], got["tiles"])


@pytest.mark.parametrize("how", [None, "tiles"])
def test_assembly_of_an_element_that_lists_a_node_twice(built, how, monkeypatch):
    """solver.rs:299-325 adds an element's nine blocks in label order wherever they land; an element (n0, n1, n0) lands two
    of a row's three blocks on the SAME entry (and has zero area: its contributions are inf / NaN, as in the reference).
    The tile kernels take their label-ordered path for it; everything else in K stays bit-identical to the oracle."""
    if how:
        monkeypatch.setenv("MAG_TUNE_ASSEMBLY", how)
    m = meshgen.plate(9)
    conn = m.conn.copy()
    conn[17, 2] = conn[17, 0]
    conn[40, 1] = conn[40, 2]
    p = meshgen.config_fixed_left_pull_right(meshgen.Mesh(m.xy, conn, "degenerate"))
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    with Context(device=0) as c:
        c.upload_problem(p)
        rowptr, col, val = c.assemble_csr()
    assert np.array_equal(rowptr.astype(np.int64), K.rowptr) and np.array_equal(col, K.col)
    bad = ~np.isfinite(K.val)
    assert bad.any() and not bad.all()
    assert np.array_equal(~np.isfinite(val), bad)              # the same entries are poisoned
    assert np.array_equal(val[~bad], K.val[~bad])              # and every other entry has the oracle's bits


@pytest.mark.parametrize("which,scale", [("hole1m", 1.0), ("plate100k", 1.0), ("plate4m", 1.0)])
def test_full_size_properties(built, which, scale):
    """BASELINE-size meshes, where the oracle does not finish in seconds: size-independent properties.
    (1) K is symmetric: x.(Ky) == y.(Kx); (2) rigid translations are in the null space of the unmasked K;
    (3) the returned u satisfies K_ff u_f = b to round-off, checked with the operator itself;
    (4) the solve is linear in the load: solve(2 delta) == 2 solve(delta)."""
    p = meshgen.baseline_problem(which, scale)
    n = 2 * p.mesh.num_nodes
    rng = np.random.default_rng(2024)
    with Context(device=0, stop_mode=MAG_STOP_REL, tol=1e-10) as c:
        c.upload_problem(p)
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        Kx, Ky = c.apply_operator(x), c.apply_operator(y)
        assert abs(x @ Ky - y @ Kx) <= 1e-11 * np.linalg.norm(x) * np.linalg.norm(Ky)
        tx = np.tile([1.0, 0.0], n // 2)
        ty = np.tile([0.0, 1.0], n // 2)
        scale_k = np.abs(Kx).max() / np.abs(x).max()
        assert np.abs(c.apply_operator(tx)).max() <= 1e-9 * scale_k
        assert np.abs(c.apply_operator(ty)).max() <= 1e-9 * scale_k
        out = c.solve(p)
        assert out["converged"] == 1
        free = p.u_known == 0
        res = c.apply_operator(out["u"])          # K u (unmasked): on free rows must equal f_in
        r = (res - p.f_in)[free]
        fs = np.abs(c.apply_operator(np.where(free, 0.0, out["u"]))).max()  # scale of K_fk u_k
        assert np.linalg.norm(r) <= 1e-8 * max(fs, np.abs(p.f_in).max()) * np.sqrt(free.sum())
        p2 = meshgen.Problem(p.mesh, p.u_known, 2.0 * p.u_in, 2.0 * p.f_in, p.youngs_modulus, p.poisson_ratio,
                             p.part_thickness)
        out2 = c.solve(p2)
        assert rel(out2["u"], 2.0 * out["u"]) <= 1e-8


def test_fp32_leg_of_config5(built):
    """mag_options.precision = 1 (BASELINE config 5's fp32 leg): converges to a loose relative tolerance and lands
    within fp32-level distance of the oracle; it is not expected to meet the 1e-8 bar."""
    p = meshgen.config_fixed_left_pull_right(meshgen.shuffle(meshgen.multi_hole(120, 3, 0.25), 2))
    ref = oracle_run(p)
    with Context(device=0, precision=1, stop_mode=MAG_STOP_REL, tol=1e-7) as c:
        out = c.solve(p)
    assert out["converged"] == 1
    err = rel(out["u"], ref["u"])
    assert 1e-9 < err < 5e-4, err
    k = p.u_known == 1
    assert np.array_equal(out["u"][k], p.u_in[k])
    # the fp32 recurrence residual keeps falling, the true error does not: a tighter tolerance buys nothing
    with Context(device=0, precision=1, stop_mode=MAG_STOP_REL, tol=1e-11, max_iter=20000) as c:
        tight = c.solve(p, allow_not_converged=True)
    assert rel(tight["u"], ref["u"]) > 1e-8
    with Context(device=0, stop_mode=MAG_STOP_REL, tol=1e-11) as c:
        f64 = c.solve(p)
    assert rel(f64["u"], ref["u"]) < rel(tight["u"], ref["u"])


def test_pathological_hub_falls_back_to_the_gather_operator(built):
    """One node shared by 3000 elements: the hub's tile needs far more halo nodes than fit its LDS image, so the
    library must switch to the global-gather operator (and the two-launch iteration) by itself -- no error, same u."""
    n = 3000
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    ring1 = np.stack([np.cos(ang), np.sin(ang)], axis=1)
    ring2 = 1.5 * ring1
    xy = np.concatenate([[[0.0, 0.0]], ring1, ring2])
    tri = [[0, 1 + k, 1 + (k + 1) % n] for k in range(n)]
    tri += [[1 + k, 1 + n + k, 1 + n + (k + 1) % n] for k in range(n)]
    tri += [[1 + k, 1 + n + (k + 1) % n, 1 + (k + 1) % n] for k in range(n)]
    m = meshgen.shuffle(meshgen.Mesh(xy, np.array(tri, dtype=np.int32), "hub"), 1)
    p = meshgen.apply_boundary_rules(m, [meshgen.BoundaryRule("hold", x_max=-1.2, ux=0.0, uy=0.0),
                                         meshgen.BoundaryRule("pull", x_min=1.2, ux=0.01, fy=0.0)])
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    A, b = oracle.reduce_system(K, p.u_known, p.u_in, p.f_in)
    _, _, _, hist_ref = oracle.cg(A, b, max_iter=30, hist_len=30)
    x = np.random.default_rng(8).standard_normal(K.n)
    # (a 3000-valence node pads its tile's slot table to 3000 slots: correct but slow, so no run to convergence here)
    for tile in (512, 256):
        with Context(device=0, tile_nodes=tile, max_iter=30, history_len=30) as c:
            out = c.solve(p, allow_not_converged=True)
            hist = c.history(30)
            y = c.apply_operator(x, masked=False)
        assert out["max_tile_halo"] > 2016 - tile and out["lds_operator"] == 0 and out["iterations"] == 30
        assert np.allclose(hist, hist_ref, rtol=1e-8)
        assert rel(y, K.spmv(x)) < 1e-12


def test_device_resident_inputs_and_outputs(built):
    """mag_problem.memory / mag_result.memory = MAG_MEM_DEVICE: pointers into HBM owned by the caller (allocated here
    with hipMalloc through ctypes -- no torch in this process, see bench.py on import order)."""
    import ctypes as C

    from magnetite_amd import _lib
    L = _lib.lib()  # pulls in libamdhip64.so.7
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    H2D, D2H = 1, 2

    def to_dev(a):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), a.nbytes) == 0
        assert hip.hipMemcpy(p, a.ctypes.data, a.nbytes, H2D) == 0
        return p

    def from_dev(p, n, dtype=np.float64):
        out = np.empty(n, dtype=dtype)
        assert hip.hipMemcpy(out.ctypes.data, p, out.nbytes, D2H) == 0
        return out

    p = PROBLEMS["plate_shuffled"]
    ref = oracle_run(p)
    N, E = p.mesh.num_nodes, p.mesh.num_elements
    with Context(device=0) as c:
        bufs = [to_dev(a) for a in (p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in)]
        outs = [to_dev(np.zeros(n)) for n in (2 * N, 2 * N, E)]
        prob = _lib.Problem(N, E, *[b.value for b in bufs], p.youngs_modulus, p.poisson_ratio, p.part_thickness,
                            _lib.MAG_MEM_DEVICE, 0)
        res = _lib.Result(*[o.value for o in outs], _lib.MAG_MEM_DEVICE, 0)
        rc = L.mag_solve(c._h, C.byref(prob), C.byref(res))
        assert rc == 0, L.mag_last_error(c._h)
        u, f, s = from_dev(outs[0], 2 * N), from_dev(outs[1], 2 * N), from_dev(outs[2], E)
        for b in bufs + outs:
            hip.hipFree(b)
    assert rel(u, ref["u"]) <= TOL_U
    k = p.u_known == 1
    assert np.array_equal(f[~k], p.f_in[~k])
    assert np.isfinite(s).all()


def _random_delaunay_problem(seed):
    """Unstructured mesh: random points in a rectangle (plus a boundary frame) triangulated by Delaunay, random node
    numbering, clamped strip on one side, a random mix of prescribed displacements / forces elsewhere."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    lx, ly = rng.uniform(0.5, 3.0), rng.uniform(0.5, 3.0)
    n = int(rng.integers(300, 2500))
    k = int(np.sqrt(n) / 2) + 3
    frame = np.concatenate([np.stack([np.linspace(0, lx, k), np.zeros(k)], 1), np.stack([np.linspace(0, lx, k), np.full(k, ly)], 1),
                            np.stack([np.zeros(k), np.linspace(0, ly, k)], 1)[1:-1], np.stack([np.full(k, lx), np.linspace(0, ly, k)], 1)[1:-1]])
    pts = np.concatenate([frame, rng.uniform([0.02 * lx, 0.02 * ly], [0.98 * lx, 0.98 * ly], size=(n, 2))])
    tri = Delaunay(pts).simplices
    a = pts[tri]
    area = 0.5 * ((a[:, 1, 0] - a[:, 0, 0]) * (a[:, 2, 1] - a[:, 0, 1]) - (a[:, 2, 0] - a[:, 0, 0]) * (a[:, 1, 1] - a[:, 0, 1]))
    tri = tri[np.abs(area) > 1e-7 * lx * ly]  # drop slivers on the frame
    area = area[np.abs(area) > 1e-7 * lx * ly]
    tri[area < 0] = tri[area < 0][:, ::-1]
    m = meshgen.shuffle(meshgen.Mesh(pts, tri.astype(np.int32), f"delaunay{seed}"), seed)
    rules = [meshgen.BoundaryRule("clamp", x_max=0.05 * lx, ux=0.0, uy=0.0)]
    if seed % 2:
        rules.append(meshgen.BoundaryRule("pull", x_min=0.95 * lx, ux=rng.uniform(-1e-3, 1e-3), fy=rng.uniform(-50, 50)))
    else:
        rules.append(meshgen.BoundaryRule("push", y_min=0.9 * ly, x_min=0.5 * lx, fx=rng.uniform(-100, 100), uy=rng.uniform(-1e-3, 1e-3)))
    rules.append(meshgen.BoundaryRule("spot", x_min=0.4 * lx, x_max=0.6 * lx, y_min=0.4 * ly, y_max=0.6 * ly,
                                      fx=rng.uniform(-1e4, 1e4), fy=rng.uniform(-1e4, 1e4)))
    return meshgen.apply_boundary_rules(m, rules, youngs_modulus=10.0 ** rng.uniform(6, 11),
                                        poisson_ratio=rng.uniform(0.05, 0.45), part_thickness=rng.uniform(0.1, 2.0))


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_unstructured_meshes(built, seed):
    p = _random_delaunay_problem(seed)
    ref = oracle_run(p)
    with Context(device=0, tile_nodes=(256, 512, 1024)[seed % 3], cg_variant=1 if seed % 4 else 0) as c:
        c.upload_problem(p)
        assert np.array_equal(c.element_stiffness(), oracle.element_stiffness_all(
            p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness))
        # mixed valences (5 ... 9+): rows on both sides of the assembly's 8-block accumulators, pattern from the incidence lists
        rowptr, col, val = c.assemble_csr()
        K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
        assert np.array_equal(rowptr.astype(np.int64), K.rowptr) and np.array_equal(col, K.col) and np.array_equal(val, K.val)
        rp, cf, vf, b = c.reduce_system()
        A, bo = oracle.reduce_system(K, p.u_known, p.u_in, p.f_in)
        assert np.array_equal(cf, A.col) and np.array_equal(vf, A.val) and np.array_equal(b, bo)
        out = c.solve(p)
    assert out["converged"] == 1
    assert rel(out["u"], ref["u"]) <= TOL_U
    # near the absolute threshold the residual wanders at round-off level: a few per cent of slack on the count
    assert abs(out["iterations"] - ref["iterations"]) <= max(5, ref["iterations"] // 20)


def test_empty_and_inconsistent_inputs_are_errors(built):
    from magnetite_amd._lib import MAG_ERR_STATE
    p = PROBLEMS["plate12"]
    with Context(device=0) as c:
        with pytest.raises(MagnetiteError) as ei:       # nothing uploaded yet
            c.run()
        assert ei.value.code == MAG_ERR_STATE
        with pytest.raises(MagnetiteError) as ei:       # empty mesh
            c.upload(np.zeros(0), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.uint8), np.zeros(0), np.zeros(0),
                     1.0, 0.3, 1.0)
        assert ei.value.code == MAG_ERR_BAD_ARGS
        with pytest.raises(MagnetiteError):             # boundary arrays of the wrong length
            c.upload(p.xy_flat, p.conn_flat, p.u_known[:-2], p.u_in, p.f_in, 1.0, 0.3, 1.0)
        with pytest.raises(MagnetiteError) as ei:       # nu = 1 makes D singular (division by 1 - nu^2)
            c.upload(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, 1.0, 1.0, 1.0)
        assert ei.value.code == MAG_ERR_BAD_ARGS
        c.upload_problem(p)                             # the context is still usable afterwards
        c.run()
        with pytest.raises(MagnetiteError):             # history was not requested
            c.history(4)
        assert c.stats()["converged"] == 1


def test_fused_beta_expansion_tracks_the_exact_recurrence(built):
    """cg_variant 1 expands |r_new|^2 = r.r + 2 alpha r.q + alpha^2 q.q instead of reducing it a second time; over
    hundreds of iterations its cost history must stay on top of the two-launch variant's (same argmin recurrences
    with every dot product reduced directly) and of the oracle's."""
    p = meshgen.config_fixed_left_pull_right(meshgen.shuffle(meshgen.plate_with_holes(96), 4))
    n = 400
    hist = {}
    for variant in (0, 1):
        with Context(device=0, cg_variant=variant, history_len=n, max_iter=n) as c:
            out = c.solve(p, allow_not_converged=True)
            assert out["iterations"] == n
            hist[variant] = c.history(n)
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    A, b = oracle.reduce_system(K, p.u_known, p.u_in, p.f_in)
    _, _, _, href = oracle.cg(A, b, max_iter=n, hist_len=n)
    # CG amplifies rounding differences slowly: tight early, still close after 400 iterations
    assert np.allclose(hist[1][:50], hist[0][:50], rtol=1e-10)
    assert np.allclose(hist[1], hist[0], rtol=1e-6)
    assert np.allclose(hist[1][:50], href[:50], rtol=1e-10) and np.allclose(hist[1], href, rtol=1e-6)


# ---- opt-in preconditioner: no reference counterpart, checked against the oracle's own PCG ----
@pytest.mark.parametrize("kind", [1, 2], ids=["jacobi", "block-jacobi"])
@pytest.mark.parametrize("tile", [256, 512])
def test_preconditioned_cg_matches_oracle_pcg(built, kind, tile):
    m = meshgen.perturb(meshgen.shuffle(meshgen.plate_with_holes(72, 60, 1.2, 1.0), 7), 0.2, 2)
    p = meshgen.config_fixed_left_point_load(m)
    a = dict(xy=p.xy_flat, conn=p.conn_flat, u_known=p.u_known, u_in=p.u_in, f_in=p.f_in, youngs=p.youngs_modulus,
             nu=p.poisson_ratio, thickness=p.part_thickness)
    n = 15
    plain = oracle.run(**a, stop_mode=oracle.STOP_REL, tol=1e-10)
    ref = oracle.run(**a, stop_mode=oracle.STOP_REL, tol=1e-10, precond=kind, hist_len=n)
    with Context(device=0, tile_nodes=tile, preconditioner=kind, stop_mode=_lib.MAG_STOP_REL, tol=1e-10,
                 history_len=n) as c:
        out = c.solve(p)
        hist = c.history(n)
        st = c.stats()
    assert st["converged"] == 1
    # same M (fp32 inverse blocks built with the same operations) and the same recurrences -> the residual history
    # starts on top of the oracle's.  Only the start is comparable: on this distorted mesh the true-residual history
    # of ANY two summation orders (numpy vs C, same textbook PCG) differs by O(1) after ~40 iterations.
    assert np.allclose(hist[:n], ref["history"][:n], rtol=1e-9), np.max(np.abs(hist[:n] / ref["history"][:n] - 1))
    assert abs(out["iterations"] - ref["iterations"]) <= max(5, ref["iterations"] // 20)
    assert out["iterations"] < 0.96 * plain["iterations"]         # the point of it on a distorted mesh (-7 % / -15 %)
    for key in ("u", "f", "stress"):
        scale = np.linalg.norm(plain[key])
        assert np.linalg.norm(out[key] - ref[key]) <= 1e-8 * scale, key
        assert np.linalg.norm(out[key] - plain[key]) <= 1e-7 * scale, key   # same solution as the reference's plain CG


def test_preconditioner_default_off_and_rejected_where_unsupported(built):
    o = _lib.Options()
    _lib.lib().mag_default_options(C.byref(o))
    assert o.preconditioner == 0
    p = meshgen.config_fixed_left_pull_right(meshgen.plate(24, 24))
    for bad in (dict(cg_variant=0), dict(precision=1), dict(cg_operator=_lib.MAG_OP_CSR), dict(op_variant=1)):
        with Context(device=0, preconditioner=2, **bad) as c:
            with pytest.raises(MagnetiteError) as e:
                c.solve(p)
            assert "preconditioner" in str(e.value)


def test_preconditioned_full_size_property(built):
    """1M-triangle benchmark mesh (two tiles per workgroup): the preconditioned solve satisfies K_ff u_f = b to the
    requested relative residual under the PLAIN operator (size-independent), and lands on plain CG's solution.  On
    this uniform mesh Jacobi scaling cannot help -- the iteration count is allowed to be somewhat higher."""
    p = meshgen.baseline_problem("hole1m")
    free = p.u_known == 0
    with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8) as c:
        base = c.solve(p)
    with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8, preconditioner=2) as c:
        out = c.solve(p)
        st = c.stats()
        assert st["converged"] == 1
        r = (c.apply_operator(out["u"]) - p.f_in)[free]
        assert np.linalg.norm(r) <= 1.01e-8 * st["rhs_norm"] + 1e-12 * np.abs(p.f_in).max()
    assert out["iterations"] <= 1.3 * base["iterations"]
    assert rel(out["u"], base["u"]) <= 1e-6


def test_multi_tile_loops_against_the_oracle(built, monkeypatch):
    """MAG_TUNE_GRID caps the persistent grid so that every workgroup walks several tiles on a mesh the oracle
    finishes in seconds (normally only the >= 1M-node meshes do, and those are checked by properties only)."""
    monkeypatch.setenv("MAG_TUNE_GRID", "3")
    p = meshgen.config_fixed_left_pull_right(meshgen.perturb(meshgen.shuffle(meshgen.plate_with_holes(80), 11), 0.2, 5))
    a = dict(xy=p.xy_flat, conn=p.conn_flat, u_known=p.u_known, u_in=p.u_in, f_in=p.f_in, youngs=p.youngs_modulus,
             nu=p.poisson_ratio, thickness=p.part_thickness)
    for kind in (0, 2):
        ref = oracle.run(**a, precond=kind)
        # LDS-DMA kernels (default) and the per-lane-record kernels (MAG_TUNE_DMA=0; tile 1024 always uses them)
        for dma, tile in (("1", 256), ("1", 512), ("0", 256), ("0", 512), ("1", 1024)):
            monkeypatch.setenv("MAG_TUNE_DMA", dma)
            with Context(device=0, tile_nodes=tile, preconditioner=kind) as c:
                out = c.solve(p)
                assert c.stats()["num_tiles"] >= 6
            assert rel(out["u"], ref["u"]) <= TOL_U, (kind, dma, tile)
            assert abs(out["iterations"] - ref["iterations"]) <= max(5, ref["iterations"] // 20), (kind, dma, tile)


def test_ring_table_on_irregular_node_stars(built):
    """The tile-local table stores every node's incident triangles as a walk over its neighbours (k_ring16).
    Operator parity on node stars that are not a single closed fan: open boundary fans, two fans meeting in one
    node (bow-tie), an edge shared by three triangles (non-manifold), mixed CCW / CW orientation, shuffled element
    order, and a 40-fan (ring longer than the register-resident words, built by the > 32 fallback)."""
    rng = np.random.default_rng(5)
    base = meshgen.plate(6, 5)
    xy = [tuple(v) for v in base.xy]
    tri = [list(t) for t in base.conn]

    def add(pt):
        xy.append(pt)
        return len(xy) - 1

    # bow-tie: a second, separate fan hanging off node 0 (corner of the plate), not edge-connected to the first
    a, b = add((-0.3, -0.1)), add((-0.1, -0.3))
    tri.append([0, a, b])
    # non-manifold edge: a third triangle on an interior edge of the plate
    e0, e1 = tri[7][0], tri[7][1]
    c = add((0.5 * (xy[e0][0] + xy[e1][0]) + 0.01, 0.5 * (xy[e0][1] + xy[e1][1]) + 0.013))
    tri.append([e0, e1, c])
    # 40-fan around a new hub, attached to the plate through one node
    hub = add((2.0, 2.0))
    ring = [add((2.0 + 0.4 * np.cos(t), 2.0 + 0.4 * np.sin(t))) for t in np.linspace(0, 2 * np.pi, 40, endpoint=False)]
    for k in range(40):
        tri.append([hub, ring[k], ring[(k + 1) % 40]])
    tri.append([base.num_nodes - 1, ring[25], ring[24]])
    tri = np.array(tri, dtype=np.int32)
    flip = rng.random(len(tri)) < 0.3                      # mixed orientation: CW elements get negative stiffness
    tri[flip] = tri[flip][:, ::-1]
    tri = tri[rng.permutation(len(tri))]
    m = meshgen.Mesh(np.array(xy, dtype=np.float64), np.ascontiguousarray(tri), "zoo")
    p = meshgen.apply_boundary_rules(m, [meshgen.BoundaryRule("hold", x_max=1e-9, ux=0.0, uy=0.0)])
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    x = rng.standard_normal(2 * m.num_nodes)
    want = K.spmv(x)
    for tile in (256, 512):
        for variant in (0, 1):
            with Context(device=0, tile_nodes=tile, op_variant=variant) as c:
                c.upload_problem(p)
                got = c.apply_operator(x, masked=False)
            assert rel(got, want) < 1e-12, (tile, variant)


@pytest.mark.parametrize("variant", [1, 0])
def test_degenerate_element_is_reported_not_hung(built, variant):
    """A zero-area triangle makes B (solver.rs:204-230: division by 2A) and hence K non-finite; the reference would
    feed NaNs to argmin.  The library must end the solve within a few launches and say so (MAG_ERR_NOT_CONVERGED,
    stats.breakdown), never spin to max_iter or fault."""
    m = meshgen.plate(8, 8)
    xy = m.xy.copy()
    conn = m.conn.copy()
    a, b, _ = conn[10]
    xy = np.vstack([xy, 0.5 * (xy[a] + xy[b])])             # a node exactly on an existing edge ...
    conn = np.vstack([conn, [a, b, len(xy) - 1]])            # ... and a triangle of area 0 on it
    p = meshgen.config_fixed_left_pull_right(meshgen.Mesh(xy, np.ascontiguousarray(conn, dtype=np.int32), "degenerate"))
    with Context(device=0, cg_variant=variant) as c:
        with pytest.raises(MagnetiteError) as e:
            c.solve(p)
        st = c.stats()
    assert "non-finite" in str(e.value)
    assert st["breakdown"] == 1 and st["converged"] == 0 and st["iterations"] <= 3


def test_repeated_solves_do_not_grow_device_memory(built):
    """Buffers are grow-only and owned by the context: alternating problem sizes and option sets for 40 solves must
    leave the free device memory where it was after the first round, and destroying the context returns it."""
    L = _lib.lib()
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()

    def free_bytes():
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    probs = [meshgen.config_fixed_left_pull_right(meshgen.plate_with_holes(n)) for n in (40, 64, 24)]
    with Context(device=0) as warm:                      # runtime pools, code objects, rocPRIM state: allocated once
        warm.solve(probs[1])
    base = free_bytes()
    c = Context(device=0)
    for p in probs:
        c.solve(p)
    after_first = free_bytes()
    for k in range(40):
        c.options.history_len = 0
        c.solve(probs[k % 3])
    assert free_bytes() >= after_first - (1 << 20)
    c.close()
    assert free_bytes() >= base - (8 << 20)
    assert L.mag_version() == 4


def test_timing_hooks_report_plausible_launch_times(built):
    """mag_time_operator / mag_time_spmv (the HIP-event figures bench.py turns into roofline fractions): they need an
    uploaded problem, not a solve (bench.py's HBM-resident leg times the 16M-triangle mesh without its 20 000
    iterations); they return positive per-launch times, the whole-iteration kernel costs more than the plain SpMV and
    neither is faster than the 8 TB/s HBM peak allows for the bytes it must move."""
    p = meshgen.config_fixed_left_pull_right(meshgen.plate_with_holes(300))
    with Context(device=0, stop_mode=MAG_STOP_REL, tol=1e-6) as c:
        with pytest.raises(MagnetiteError):
            c.time_operator(10)                   # nothing uploaded
        c.upload_problem(p)
        cold_it, cold_mv = c.time_operator(10), c.time_spmv(10)   # straight after the upload: symbolic phase only
        assert cold_it > 0 and cold_mv > 0
        c.run()
        us_it = min(c.time_operator(50) for _ in range(3)) * 1e3   # best of three: a busy box must not fail the test
        assert 0.5 < cold_it * 1e3 / us_it < 2.0
        us_mv = min(c.time_spmv(50) for _ in range(3)) * 1e3
        u1 = c.download()[0]
        c.run()                                  # the hooks run on scratch state: a later solve is unaffected
        assert np.array_equal(c.download()[0], u1)
    E, N = p.mesh.num_elements, p.mesh.num_nodes
    assert us_it > us_mv > 0
    assert (12 * E + 50 * N) / (us_mv * 1e-6) < 8e12 and 177 * N / (us_it * 1e-6) < 8e12
    for variant in (0,):
        with Context(device=0, cg_variant=variant, stop_mode=MAG_STOP_REL, tol=1e-6) as c:
            c.upload_problem(p)
            c.run()
            assert c.time_operator(20) > 0


# ---- on-chip single-launch CG (cg_variant 2, the default when the mesh fits the chip) ----
def test_on_chip_cg_matches_the_oracle(built, monkeypatch):
    """k_cg_persist: the whole solve in one launch, workgroups meeting at a grid barrier every iteration.  The library
    picks it by itself only for 3-4 tiles of 512 nodes per CU (0.5-1M triangles); MAG_TUNE_PERSIST_MIN_K=1 lets small
    meshes reach it: one workgroup, a few workgroups, workgroups with unequal tile counts, both tile sizes, every stop
    mode, the iteration cap, the cost history and the zero right-hand side."""
    monkeypatch.setenv("MAG_TUNE_PERSIST_MIN_K", "1")
    cases = [("plate12", meshgen.config_fixed_left_pull_right(meshgen.plate(12)), 512),
             ("hole48", meshgen.config_fixed_left_pull_right(meshgen.shuffle(meshgen.plate_with_holes(48), 7)), 256),
             ("hole150", meshgen.config_fixed_left_point_load(meshgen.perturb(meshgen.plate_with_holes(150), 0.2, 4)), 512),
             ("cw", meshgen.config_fixed_left_pull_right(meshgen.clockwise(meshgen.plate(20))), 512)]
    for name, p, tile in cases:
        ref = oracle_run(p, hist_len=50)
        with Context(device=0, tile_nodes=tile, history_len=50) as c:
            out = c.solve(p)
            st = c.stats()
            hist = c.history(min(50, out["iterations"]))
        assert st["cg_kernel"] == 2, name
        assert out["converged"] == 1 and abs(out["iterations"] - ref["iterations"]) <= max(3, ref["iterations"] // 50), name
        for key in ("u", "f", "stress"):
            assert rel(out[key], ref[key]) <= TOL_U, (name, key)
        k = min(len(hist), len(ref["history"]), 30)
        assert np.allclose(hist[:k], ref["history"][:k], rtol=1e-8), name
    p = cases[2][1]
    for mode, tol in ((MAG_STOP_REL, 1e-9), (MAG_STOP_RNORM_SQ, 1e-6)):
        ref = oracle_run(p, stop_mode={MAG_STOP_REL: oracle.STOP_REL, MAG_STOP_RNORM_SQ: oracle.STOP_RNORM_SQ}[mode], tol=tol)
        with Context(device=0, tile_nodes=512, stop_mode=mode, tol=tol) as c:
            out = c.solve(p)
            assert c.stats()["cg_kernel"] == 2
        assert abs(out["iterations"] - ref["iterations"]) <= 5 and rel(out["u"], ref["u"]) <= TOL_U
    with Context(device=0, tile_nodes=512, max_iter=37) as c:
        out = c.solve(p, allow_not_converged=True)
        assert c.stats()["cg_kernel"] == 2 and out["iterations"] == 37 and out["converged"] == 0
    z = meshgen.Problem(p.mesh, p.u_known, 0.0 * p.u_in, 0.0 * p.f_in, p.youngs_modulus, p.poisson_ratio, p.part_thickness)
    with Context(device=0, tile_nodes=512) as c:
        out = c.solve(z)
        assert out["iterations"] == 0 and out["converged"] == 1 and not out["u"].any()


def test_on_chip_cg_is_chosen_for_the_benchmark_mesh_and_agrees_with_streaming(built):
    """1M-triangle benchmark mesh: the default picks the on-chip kernel (4 tiles of 512 nodes per CU); the streaming
    kernel (cg_variant 1) takes the same number of iterations and lands on the same displacements."""
    p = meshgen.baseline_problem("hole1m")
    with Context(device=0, stop_mode=MAG_STOP_REL, tol=1e-8) as c:
        a = c.solve(p)
        assert c.stats()["cg_kernel"] == 2
        b = c.solve(p)
    assert np.array_equal(a["u"], b["u"])                      # bitwise reproducible
    with Context(device=0, stop_mode=MAG_STOP_REL, tol=1e-8, cg_variant=1) as c:
        s = c.solve(p)
        assert c.stats()["cg_kernel"] == 1
    assert abs(a["iterations"] - s["iterations"]) <= 2
    assert rel(a["u"], s["u"]) <= 1e-9


def test_on_chip_cg_falls_back_when_the_grid_barrier_times_out(built, monkeypatch):
    """A workgroup that does not see every arrival within its spin budget sets the timeout word and leaves (the grid
    is not fully resident: GPU shared with another process).  MAG_TUNE_PERSIST_SPIN=0 makes every wait give up at
    once: the solve must still come back correct, through the streaming kernels; the context then streams for 8 solves
    and tries the on-chip kernel again (a long-lived caller on a once-shared GPU gets the fast path back); a second
    failure doubles the wait."""
    monkeypatch.setenv("MAG_TUNE_PERSIST_MIN_K", "1")
    monkeypatch.setenv("MAG_TUNE_PERSIST_SPIN", "0")
    p = meshgen.config_fixed_left_pull_right(meshgen.plate_with_holes(96))
    ref = oracle_run(p)
    with Context(device=0, tile_nodes=512) as c:
        out = c.solve(p)
        assert c.stats()["cg_kernel"] == 1
        assert rel(out["u"], ref["u"]) <= TOL_U and out["iterations"] == ref["iterations"]
        assert c.stats()["persist_timeout"] == 1
        monkeypatch.delenv("MAG_TUNE_PERSIST_SPIN")
        for _ in range(7):                                     # streams, without paying another spin budget
            out = c.solve(p)
            assert c.stats()["cg_kernel"] == 1 and c.stats()["persist_timeout"] == 0
        first = out
        out = c.solve(p)                                       # the 8th solve after the failure: on-chip again
        assert c.stats()["cg_kernel"] == 2 and c.stats()["persist_timeout"] == 0
        assert rel(out["u"], ref["u"]) <= TOL_U and rel(out["u"], first["u"]) <= 1e-9
        monkeypatch.setenv("MAG_TUNE_PERSIST_SPIN", "0")       # shared again: fails, ...
        c.solve(p)
        assert c.stats()["cg_kernel"] == 1 and c.stats()["persist_timeout"] == 1
        monkeypatch.delenv("MAG_TUNE_PERSIST_SPIN")
        c.solve(p)
        assert c.stats()["cg_kernel"] == 1                     # ... and waits again
    with Context(device=0, tile_nodes=512) as c:
        c.solve(p)
        assert c.stats()["cg_kernel"] == 2


def test_on_chip_cg_with_rings_longer_than_its_registers(built):
    """A node of valence 40 (its row becomes 80 single-triangle entries) and one of valence 12 in 512-node tiles: the
    on-chip kernel keeps ten ring entries per node in registers and walks the rest of the tile's longest row from
    the table in memory."""
    n = 40
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    xy = np.concatenate([[[0.0, 0.0]], np.stack([np.cos(ang), np.sin(ang)], axis=1),
                         1.8 * np.stack([np.cos(ang), np.sin(ang)], axis=1)])
    tri = [[0, 1 + k, 1 + (k + 1) % n] for k in range(n)]
    tri += [[1 + k, 1 + n + k, 1 + n + (k + 1) % n] for k in range(n)]
    tri += [[1 + k, 1 + n + (k + 1) % n, 1 + (k + 1) % n] for k in range(n)]
    rules = [meshgen.BoundaryRule("hold", x_max=-1.2, ux=0.0, uy=0.0), meshgen.BoundaryRule("pull", x_min=1.2, ux=0.01, fy=0.0)]
    p40 = meshgen.apply_boundary_rules(meshgen.Mesh(xy, np.array(tri, dtype=np.int32), "fan40"), rules)
    m12 = 12
    ang = np.linspace(0, 2 * np.pi, m12, endpoint=False)
    xy = np.concatenate([[[0.0, 0.0]], np.stack([np.cos(ang), np.sin(ang)], axis=1),
                         1.8 * np.stack([np.cos(ang), np.sin(ang)], axis=1)])
    tri = [[0, 1 + k, 1 + (k + 1) % m12] for k in range(m12)]
    tri += [[1 + k, 1 + m12 + k, 1 + m12 + (k + 1) % m12] for k in range(m12)]
    tri += [[1 + k, 1 + m12 + (k + 1) % m12, 1 + (k + 1) % m12] for k in range(m12)]
    p12 = meshgen.apply_boundary_rules(meshgen.Mesh(xy, np.array(tri, dtype=np.int32), "fan12"), rules)
    for p in (p40, p12):
        ref = oracle_run(p)
        with Context(device=0, tile_nodes=512) as c:
            out = c.solve(p)
            assert c.stats()["cg_kernel"] == 2
        assert rel(out["u"], ref["u"]) <= TOL_U and abs(out["iterations"] - ref["iterations"]) <= 3


def _plate_with_a_hub(nx):
    """plate(nx) with ONE interior node of valence 12: each of its six triangles (v, a, b) is split at the midpoint of
    the ring edge a-b (the midpoints hang on the outer side: non-conforming, which neither solver minds)."""
    m = meshgen.plate(nx)
    v = (nx // 3) * (nx + 1) + nx // 3
    conn, xy = m.conn.tolist(), m.xy.tolist()
    out = []
    for tri in conn:
        if v not in tri:
            out.append(tri)
            continue
        k = tri.index(v)
        a, b = tri[(k + 1) % 3], tri[(k + 2) % 3]
        xy.append([(xy[a][0] + xy[b][0]) / 2, (xy[a][1] + xy[b][1]) / 2])
        mid = len(xy) - 1
        out += [[v, a, mid], [v, mid, b]]
    return meshgen.Mesh(np.array(xy), np.array(out, dtype=np.int32), f"hub_plate_{nx}")


@pytest.mark.parametrize("case", ["perturbed_hole", "hub"])
def test_on_chip_sibling_tiles_on_irregular_meshes(built, case):
    """Two 512-node tiles per workgroup (more than 256 tiles): the tiles of a workgroup read each other's LDS slots instead of
    keeping halo copies of each other's nodes, and nodes only siblings read publish nothing.  Shuffled caller numbering,
    perturbed coordinates and a stair-step hole; and a mesh with a valence-12 node, whose tile has rows longer than the
    registers hold: that tile keeps its halo copies (its siblings must still publish for it) while the others do not."""
    if case == "hub":
        mesh = meshgen.shuffle(meshgen.perturb(_plate_with_a_hub(372), 0.15, 3), 11)
    else:
        mesh = meshgen.shuffle(meshgen.perturb(meshgen.plate_with_holes(385), 0.2, 5), 9)
    p = meshgen.config_fixed_left_pull_right(mesh)
    assert p.mesh.num_nodes > 256 * 512  # two tiles per workgroup
    ref = oracle_run(p, stop_mode=oracle.STOP_REL, tol=1e-9)
    with Context(device=0, stop_mode=MAG_STOP_REL, tol=1e-9) as c:
        out = c.solve(p)
        st = c.stats()
        again = c.solve(p)
    assert st["cg_kernel"] == 2 and st["num_tiles"] > 256 and out["converged"] == 1
    assert abs(out["iterations"] - ref["iterations"]) <= max(3, ref["iterations"] // 200)
    assert rel(out["u"], ref["u"]) <= TOL_U and np.array_equal(out["u"], again["u"])
    with Context(device=0, stop_mode=MAG_STOP_REL, tol=1e-9, cg_variant=1) as c:
        streamed = c.solve(p)
    assert rel(out["u"], streamed["u"]) <= 1e-9


def test_one_context_alternates_between_streaming_and_on_chip_solves(built):
    """The same context solves a mesh the chip cannot hold (more than 524 288 nodes: streamed, one fused launch per iteration
    replayed from a graph), one it keeps on chip, a small one (on chip as well since round 4: every mesh one GPU can hold), and
    the first again: tile tables, graph and kernel choice all follow the problem, results equal fresh contexts'."""
    big = meshgen.config_fixed_left_pull_right(meshgen.plate_with_holes(760))        # 577k nodes: streamed
    mid = meshgen.config_fixed_left_pull_right(meshgen.plate_with_holes(300))        # 90k nodes: on chip
    small = meshgen.config_fixed_left_pull_right(meshgen.plate_with_holes(60))       # 3.6k nodes: on chip
    assert big.mesh.num_nodes > 524288
    kw = dict(stop_mode=MAG_STOP_REL, tol=1e-8)
    fresh = {}
    for name, p in (("big", big), ("mid", mid), ("small", small)):
        with Context(device=0, **kw) as c:
            fresh[name] = (c.solve(p), c.stats()["cg_kernel"])
    assert fresh["big"][1] == 1 and fresh["mid"][1] == 2 and fresh["small"][1] == 2
    with Context(device=0, **kw) as c:
        for name, p in (("big", big), ("mid", mid), ("small", small), ("big", big), ("mid", mid)):
            out = c.solve(p)
            assert c.stats()["cg_kernel"] == fresh[name][1]
            assert np.array_equal(out["u"], fresh[name][0]["u"]) and out["iterations"] == fresh[name][0]["iterations"]


def test_small_meshes_run_the_on_chip_kernel_by_default(built):
    """Round 4: on one GPU the on-chip kernel takes every mesh it can hold -- the reference's own examples are a few thousand
    triangles (3.85 against 6.5 us per iteration for the streamed kernels at 1k-27k triangles, and no graph to instantiate in
    the first solve).  One tile and a half, a handful of tiles, the tensile fixture: oracle parity with the default options."""
    for p in (meshgen.config_fixed_left_pull_right(meshgen.plate_with_holes(24)),
              meshgen.config_fixed_left_point_load(meshgen.shuffle(meshgen.plate(48), 2)),
              meshgen.config_fixed_left_pull_right(meshgen.frontal_like(30, 0.4, 5))):
        ref = oracle_run(p)
        with Context(device=0) as c:
            out = c.solve(p)
            st = c.stats()
        assert st["cg_kernel"] == 2 and st["edge_blocks"] in (1, 2), p.mesh.name
        assert out["converged"] == 1 and abs(out["iterations"] - ref["iterations"]) <= max(3, ref["iterations"] // 50)
        for key in ("u", "f", "stress"):
            assert rel(out[key], ref[key]) <= TOL_U, (p.mesh.name, key)


def _disc(m, rings):
    """A disc triangulated ring by ring: ring k carries m * k nodes, so the centre is a CLOSED fan of valence m and every
    other interior node has valence 6; boundary nodes carry open fans of three or four triangles."""
    pts, start = [[0.0, 0.0]], [0]
    for k in range(1, rings + 1):
        start.append(len(pts))
        for i in range(m * k):
            a = 2 * np.pi * i / (m * k)
            pts.append([k * np.cos(a), k * np.sin(a)])
    tri = []
    for k in range(rings):
        n_in, n_out = max(1, m * k), m * (k + 1)
        inner = lambda i: start[k] + (i % n_in if k > 0 else 0)
        outer = lambda i: start[k + 1] + i % n_out
        for s in range(m):                    # sector s: k inner edges, k + 1 outer edges
            for j in range(k + 1):
                o = s * (k + 1) + j
                i = s * k + j
                tri.append([inner(min(i, s * k + k) if k else 0), outer(o), outer(o + 1)])
                if j < k:
                    tri.append([inner(i), outer(o + 1), inner(i + 1)])
    return meshgen.Mesh(np.array(pts), np.array(tri, dtype=np.int32), f"disc{m}x{rings}")


@pytest.mark.parametrize("case", ["plate", "hole_perturbed", "clockwise", "disc3", "disc4", "disc5", "tile256", "tile256_k8"])
def test_on_chip_edge_blocks_match_the_oracle_and_the_triangle_walk(built, case, monkeypatch):
    """Meshes whose nodes all carry ONE fan of at most six entries run the on-chip kernel's edge-block instantiation
    (mag_stats.edge_blocks): symmetric 2 x 2 blocks per ring entry built once per solve by k_edge_blocks, antisymmetric
    parts telescoped per fan.  Closed fans of valence 3, 4, 5 (disc centres; the closing entry sits inside the blocks) and 6
    (the closing triangle is folded onto entry 0), open fans at the boundary, clockwise elements, perturbed coordinates,
    both tile sizes.  Checked against the oracle and against the triangle walk of the same library."""
    monkeypatch.setenv("MAG_TUNE_PERSIST_MIN_K", "1")
    tile = 512
    if case == "plate":
        p = meshgen.config_fixed_left_point_load(meshgen.plate(40))
    elif case == "hole_perturbed":
        p = meshgen.config_fixed_left_pull_right(meshgen.shuffle(meshgen.perturb(meshgen.plate_with_holes(120), 0.2, 4), 3))
    elif case == "clockwise":
        p = meshgen.config_fixed_left_pull_right(meshgen.clockwise(meshgen.plate_with_holes(64)))
    elif case == "tile256":
        p, tile = meshgen.config_fixed_left_pull_right(meshgen.plate_with_holes(96)), 256
    elif case == "tile256_k8":  # eight 256-node tiles per workgroup: sibling slots reach seven tile images up and down
        monkeypatch.setenv("MAG_TUNE_PERSIST_K", "8")
        p, tile = meshgen.config_fixed_left_pull_right(meshgen.shuffle(meshgen.plate_with_holes(110), 5)), 256
    else:
        m = int(case[-1])
        mesh = meshgen.perturb(_disc(m, 24), 0.1, m)
        rules = [meshgen.BoundaryRule("hold", x_max=-14.0, ux=0.0, uy=0.0), meshgen.BoundaryRule("pull", x_min=14.0, ux=0.01, fy=0.0)]
        p = meshgen.apply_boundary_rules(mesh, rules)
    ref = oracle_run(p)
    with Context(device=0, tile_nodes=tile) as c:
        out = c.solve(p)
        st = c.stats()
        monkeypatch.setenv("MAG_TUNE_PERSIST_TRIANGLES", "1")
        walk = c.solve(p)
        st_walk = c.stats()
        monkeypatch.delenv("MAG_TUNE_PERSIST_TRIANGLES")
        again = c.solve(p)
    assert st["cg_kernel"] == 2 and st["edge_blocks"] == 1, case
    assert st_walk["cg_kernel"] == 2 and st_walk["edge_blocks"] == 0, case
    assert out["converged"] == 1 and abs(out["iterations"] - ref["iterations"]) <= max(3, ref["iterations"] // 50), case
    for key in ("u", "f", "stress"):
        assert rel(out[key], ref[key]) <= TOL_U, (case, key)
    assert rel(out["u"], walk["u"]) <= 1e-9 and abs(out["iterations"] - walk["iterations"]) <= 2, case
    assert np.array_equal(out["u"], again["u"])              # bitwise reproducible


def _poisson_delaunay(n_pts, seed):
    """Random points in the unit square plus a boundary frame, Delaunay-triangulated: valences from 3 to 12 and more."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    m = int(np.sqrt(n_pts))
    t = np.linspace(0.0, 1.0, m + 1)
    frame = np.concatenate([np.stack([t, 0 * t], 1), np.stack([t, 0 * t + 1], 1), np.stack([0 * t[1:-1], t[1:-1]], 1),
                            np.stack([0 * t[1:-1] + 1, t[1:-1]], 1)])
    pts = np.concatenate([frame, rng.uniform(0.5 / m, 1 - 0.5 / m, size=(n_pts, 2))])
    tri = Delaunay(pts).simplices.astype(np.int64)
    a = pts[tri]
    area = 0.5 * ((a[:, 1, 0] - a[:, 0, 0]) * (a[:, 2, 1] - a[:, 0, 1]) - (a[:, 2, 0] - a[:, 0, 0]) * (a[:, 1, 1] - a[:, 0, 1]))
    keep = np.abs(area) > 1e-12
    tri, area = tri[keep], area[keep]
    tri[area < 0] = tri[area < 0][:, ::-1]
    return meshgen.Mesh(pts, np.ascontiguousarray(tri.astype(np.int32)), f"poisson_{n_pts}_{seed}")


@pytest.mark.parametrize("case", ["frontal", "frontal_shuffled_cw", "frontal_two_tiles_per_wg", "disc8", "disc9", "disc12", "poisson",
                                  "structured_forced"])
def test_on_chip_edge_blocks_with_overflow_match_the_oracle_and_the_triangle_walk(built, case, monkeypatch):
    """Round 4: meshes whose rows are single fans of ANY length -- what gmsh's frontal mesher hands solver::run
    (mesher.rs:501-506): a quarter of the nodes with seven or more neighbours -- run the edge-block kernel with the blocks
    beyond the six in registers as records in an LDS pool (mag_stats.edge_blocks == 2; k_edge_blocks_ovf,
    ring_walk_blocks_ovf).  Jittered-lattice Delaunay meshes (valence 4-10), a hub of valence 12, disc centres of valence 8
    and 9 (closed fans folded onto block 0), random-point Delaunay (valence up to 12+), shuffled numbering, clockwise
    elements, two tiles per workgroup (sibling slots in the pool's entries), and a structured mesh forced through the
    overflow instantiation (no overflow records at all: the compact LDS layout alone).  Against the oracle and against the
    triangle walk of the same library."""
    monkeypatch.setenv("MAG_TUNE_PERSIST_MIN_K", "1")
    kw = {}
    if case == "frontal":
        p = meshgen.config_fixed_left_pull_right(meshgen.frontal_like(100, 0.4, 3))
    elif case == "frontal_shuffled_cw":
        p = meshgen.config_fixed_left_point_load(meshgen.clockwise(meshgen.shuffle(meshgen.frontal_like(90, 0.45, 5), 8)))
    elif case == "frontal_two_tiles_per_wg":
        p = meshgen.config_fixed_left_pull_right(meshgen.shuffle(meshgen.frontal_like(345, 0.4, 7), 2))
        assert p.mesh.num_nodes > 256 * 512
        kw = dict(stop_mode=MAG_STOP_REL, tol=1e-9)
    elif case.startswith("disc"):
        m = int(case[4:])
        mesh = meshgen.perturb(_disc(m, 12), 0.05, 2)
        p = meshgen.apply_boundary_rules(mesh, [meshgen.BoundaryRule("hold", x_max=-7.0, ux=0.0, uy=0.0),
                                                meshgen.BoundaryRule("pull", x_min=7.0, ux=0.01, fy=0.0)])
    elif case == "poisson":
        p = meshgen.config_fixed_left_pull_right(_poisson_delaunay(9000, 4))
    else:
        monkeypatch.setenv("MAG_TUNE_PERSIST_FORCE_OVERFLOW", "1")
        p = meshgen.config_fixed_left_pull_right(meshgen.plate_with_holes(96))
    ref = oracle_run(p, **({"stop_mode": oracle.STOP_REL, "tol": 1e-9} if kw else {}))
    with Context(device=0, tile_nodes=512, **kw) as c:
        out = c.solve(p)
        st = c.stats()
        monkeypatch.setenv("MAG_TUNE_PERSIST_TRIANGLES", "1")
        walk = c.solve(p)
        st_walk = c.stats()
        monkeypatch.delenv("MAG_TUNE_PERSIST_TRIANGLES")
        again = c.solve(p)
    assert st["cg_kernel"] == 2 and st["edge_blocks"] == 2, (case, st["edge_blocks"])
    assert st_walk["cg_kernel"] == 2 and st_walk["edge_blocks"] == 0, case
    assert out["converged"] == 1 and abs(out["iterations"] - ref["iterations"]) <= max(3, ref["iterations"] // 50), case
    for key in ("u", "f", "stress"):  # (a relative stop at 1e-9 leaves the reactions and stresses at ~1e-8)
        assert rel(out[key], ref[key]) <= (TOL_U if key == "u" or not kw else 1e-7), (case, key)
    # (the reference's stop rule runs to round-off, where the iteration count wobbles with the order of the sums)
    # (... and two solves stopped at a relative 1e-9 differ by about that much)
    assert rel(out["u"], walk["u"]) <= (1e-8 if kw else 1e-9), case
    assert abs(out["iterations"] - walk["iterations"]) <= max(3, ref["iterations"] // 50), case
    assert np.array_equal(out["u"], again["u"])              # bitwise reproducible
    if case == "structured_forced":  # the same blocks as the register-only instantiation: the same solution to rounding
        monkeypatch.delenv("MAG_TUNE_PERSIST_FORCE_OVERFLOW")
        with Context(device=0, tile_nodes=512) as c:
            plain = c.solve(p)
            assert c.stats()["edge_blocks"] == 1
        assert rel(out["u"], plain["u"]) <= 1e-12 and out["iterations"] == plain["iterations"]


@pytest.mark.parametrize("case", ["blocks_k2", "blocks_k3", "overflow_k3", "blocks_one_wg_1", "blocks_one_wg_2", "blocks_one_wg_3",
                                  "overflow_one_wg_1", "overflow_one_wg_2", "overflow_one_wg_3"])
def test_on_chip_node_slots_per_lane_follow_the_tiles_per_workgroup(built, case, monkeypatch):
    """A workgroup of the on-chip kernel that holds fewer than four 512-node tiles runs the instantiation with as many
    node slots per lane (k_cg_persist's NPTX; config 2 runs one tile per workgroup): two and three tiles per workgroup on
    a grid of several workgroups, and the single-workgroup instantiation at one, two and three tiles, for the edge-block
    kernel and its overflow variant.  (One tile per workgroup on a grid is what every mid-size case of the tests above
    runs; two tiles with overflow records is their `frontal_two_tiles_per_wg`.)  Against the oracle."""
    kind, rest = case.split("_", 1)
    if rest.startswith("k"):
        k = int(rest[1:])
        monkeypatch.setenv("MAG_TUNE_PERSIST_K", str(k))
        mesh = meshgen.shuffle(meshgen.plate_with_holes(120), 3) if kind == "blocks" else meshgen.frontal_like(110, 0.4, 9)
        one_wg = False
    else:
        k = int(rest[-1])
        n = {1: 20, 2: 29, 3: 36}[k]  # 441, 900, 1369 nodes (plate); the lattice of frontal_like about as many
        mesh = meshgen.plate(n) if kind == "blocks" else meshgen.frontal_like({1: 19, 2: 27, 3: 34}[k], 0.4, k)
        one_wg = True
    p = meshgen.config_fixed_left_pull_right(mesh)
    tiles = (p.mesh.num_nodes + 511) // 512
    assert (tiles == k) if one_wg else (tiles > 2 * k), (case, p.mesh.num_nodes)
    ref = oracle_run(p)
    with Context(device=0, tile_nodes=512) as c:
        out = c.solve(p)
        st = c.stats()
        again = c.solve(p)
    assert st["cg_kernel"] == 2 and st["edge_blocks"] == (1 if kind == "blocks" else 2), (case, st)
    assert st["tiles_per_workgroup"] == k, (case, st["tiles_per_workgroup"])
    assert out["converged"] == 1 and abs(out["iterations"] - ref["iterations"]) <= max(3, ref["iterations"] // 50), case
    for key in ("u", "f", "stress"):
        assert rel(out[key], ref[key]) <= TOL_U, (case, key)
    assert np.array_equal(out["u"], again["u"])


def test_on_chip_edge_blocks_are_refused_where_a_row_does_not_fit(built, monkeypatch):
    """A node whose triangles form TWO fans (a plate with one cell pair removed so that two corner cells touch only at a
    node; the hanging midpoints around the valence-12 hub of _plate_with_a_hub) keeps the whole mesh on the triangle walk
    (k_ring16's flag, bit 1); so does a single-fan mesh with the overflow instantiation switched off
    (MAG_TUNE_PERSIST_NO_OVERFLOW=1: round 3's all-or-nothing rule)."""
    monkeypatch.setenv("MAG_TUNE_PERSIST_MIN_K", "1")
    xy, tri, cx, cy = meshgen._grid(64, 64, 1.0, 1.0)  # cells (20, 20) and (21, 21) removed: node (21, 21) keeps two cells
    keep = np.ones(cx.shape[0], dtype=bool)             # that touch only at the node -- two fans
    keep[[20 * 64 + 20, 21 * 64 + 21]] = False
    two_fans = meshgen._compact(xy, tri, keep, "pinched_plate")
    for mesh, env in ((two_fans, None), (_plate_with_a_hub(60), None), (meshgen.frontal_like(80, 0.4, 2), "MAG_TUNE_PERSIST_NO_OVERFLOW")):
        if env:
            monkeypatch.setenv(env, "1")
        p = meshgen.config_fixed_left_pull_right(mesh)
        ref = oracle_run(p)
        with Context(device=0, tile_nodes=512) as c:
            out = c.solve(p)
            st = c.stats()
        assert st["cg_kernel"] == 2 and st["edge_blocks"] == 0, mesh.name
        assert rel(out["u"], ref["u"]) <= TOL_U, mesh.name
        if env:
            monkeypatch.delenv(env)


def _plate_with_removed_cells(nx, seed, rects, p_random):
    """plate(nx) minus axis-parallel rectangles of cells and, with probability p_random each, single cells: rectangles keep
    every node's triangles in one fan (the mesh stays eligible for edge blocks); a single removed cell can leave a node
    whose two remaining corner cells touch only at the node -- two fans -- which must send the mesh to the triangle walk."""
    rng = np.random.default_rng(seed)
    xy, tri, cx, cy = meshgen._grid(nx, nx, 1.0, 1.0)
    keep = np.ones(cx.shape[0], dtype=bool)
    for _ in range(rects):
        x0, y0 = rng.uniform(0.1, 0.7, 2)
        w, h = rng.uniform(0.05, 0.2, 2)
        keep &= ~((cx > x0) & (cx < x0 + w) & (cy > y0) & (cy < y0 + h))
    if p_random > 0:
        inner = (cx > 0.1) & (cx < 0.9) & (cy > 0.1) & (cy < 0.9)
        keep &= ~(inner & (rng.uniform(size=cx.shape[0]) < p_random))
    return meshgen._compact(xy, tri, keep, f"cells_{nx}_{seed}")


@pytest.mark.parametrize("seed", range(6))
def test_on_chip_kernel_choice_on_random_cell_patterns(built, seed, monkeypatch):
    """Random structured meshes: rectangles of removed cells (every node star stays one fan: edge blocks) and, for the odd
    seeds, scattered single cells as well (nodes with two fans appear: triangle walk).  Either way the oracle's answer."""
    monkeypatch.setenv("MAG_TUNE_PERSIST_MIN_K", "1")
    mesh = _plate_with_removed_cells(72 + 8 * seed, seed, rects=2 + seed % 3, p_random=0.01 if seed % 2 else 0.0)
    p = meshgen.config_fixed_left_pull_right(meshgen.shuffle(mesh, seed))
    ref = oracle_run(p)
    with Context(device=0, tile_nodes=512) as c:
        out = c.solve(p)
        st = c.stats()
    assert st["cg_kernel"] == 2, seed
    if seed % 2 == 0:
        assert st["edge_blocks"] == 1, seed
    assert out["converged"] == 1 and abs(out["iterations"] - ref["iterations"]) <= max(3, ref["iterations"] // 50), seed
    for key in ("u", "f", "stress"):
        assert rel(out[key], ref[key]) <= TOL_U, (seed, key)
