"""Hand-derived known answers for the ASSEMBLED system: K (solver.rs:290-331), the K_ff partition and b
(solver.rs:365-404,427-432) and the exact-zero drop (solver.rs:126-137), on a 2-element and an 8-element mesh.

The reference ships no vectors (PARITY UNPINNED), and the oracle and the HIP kernels restate the same matrix
products; what pins both here is tests/closed_form.py: literal matrices worked out by hand plus the textbook block
formula in exact rational arithmetic -- no B, no D, no gemm.  All inputs are dyadic, so every correct evaluation
order gives these bits: the comparisons are array_equal, for the oracle (CPU) and for mag_assemble_csr /
mag_reduce_system (GPU).
"""
import numpy as np
import pytest

import closed_form as cf
import oracle

CASES = {
    "sq2": (cf.SQ2_XY, cf.SQ2_TRIS, cf.SQ2_MATERIAL, cf.SQ2_U_KNOWN, cf.SQ2_U_IN, cf.SQ2_F_IN),
    "g8": (cf.G8_XY, cf.G8_TRIS, cf.G8_MATERIAL, cf.G8_U_KNOWN, cf.G8_U_IN, cf.G8_F_IN),
}


def flat(case):
    xy, tris, mat, u_known, u_in, f_in = CASES[case]
    return (np.array([[float(x), float(y)] for x, y in xy]).reshape(-1), np.array(tris, dtype=np.int32).reshape(-1),
            float(mat["nu"]), float(mat["youngs"]), float(mat["thick"]), np.array(u_known, dtype=np.uint8),
            np.array([float(v) for v in u_in]), np.array([float(v) for v in f_in]))


def expected(case):
    xy, tris, mat, u_known, u_in, f_in = CASES[case]
    K = cf.assemble(xy, tris, **mat)
    Kff, b = cf.reduce(K, u_known, u_in, f_in)
    assert cf.exactly_representable(K) and cf.exactly_representable(Kff) and cf.exactly_representable(b)
    return cf.to_float(K), cf.to_float(Kff), cf.to_float(b), cf.csr_of_dense(Kff)


def dense_of(rowptr, col, val, n):
    A = np.zeros((n, n))
    for r in range(n):
        A[r, col[rowptr[r]:rowptr[r + 1]]] = val[rowptr[r]:rowptr[r + 1]]
    return A


def test_literals_agree_with_the_rational_algebra():
    """the hand-worked literals of KAT-SQ2 and the spot values of KAT-G8 == the block formula in Fractions"""
    K, Kff, b, (rp, col, val) = expected("sq2")
    assert np.array_equal(K, np.array(cf.SQ2_K))
    assert np.array_equal(Kff, np.array(cf.SQ2_KFF))
    assert np.array_equal(b, np.array(cf.SQ2_B))
    assert rp.tolist() == cf.SQ2_KFF_ROWPTR and col.tolist() == cf.SQ2_KFF_COL and val.tolist() == cf.SQ2_KFF_VAL
    K8 = expected("g8")[0]
    for (r, c), v in cf.G8_SPOT.items():
        assert K8[r, c] == v
    assert np.array_equal(K8, K8.T) and np.all(K8.sum(axis=1) == 0.0)  # exact: dyadic entries


@pytest.mark.parametrize("case", list(CASES))
def test_oracle_dense_path_hits_the_closed_form(built, case):
    """the reference-faithful O(n^2) path: K_e -> dense scatter -> dense partition -> != 0 scan"""
    xy, conn, nu, youngs, t, u_known, u_in, f_in = flat(case)
    K, Kff, b, (rp, col, val) = expected(case)
    ke = oracle.element_stiffness_all(xy, conn, nu, youngs, t)
    for e, tri in enumerate(CASES[case][1]):
        assert np.array_equal(ke[e], cf.to_float(cf.element_stiffness(CASES[case][0], tri, **CASES[case][2])))
    Kd = oracle.assemble_dense(xy.size // 2, conn, ke)
    assert np.array_equal(Kd, K)
    Kff_d, b_d = oracle.partition_dense(Kd, u_known, u_in, f_in)
    assert np.array_equal(Kff_d, Kff)
    assert np.array_equal(b_d, b)
    A = oracle.sparsify_dense(Kff_d)
    assert np.array_equal(A.rowptr, rp) and np.array_equal(A.col, col) and np.array_equal(A.val, val)


@pytest.mark.parametrize("case", list(CASES))
def test_oracle_sparse_path_hits_the_closed_form(built, case):
    xy, conn, nu, youngs, t, u_known, u_in, f_in = flat(case)
    K, Kff, b, (rp, col, val) = expected(case)
    Ks = oracle.assemble_sparse(xy, conn, nu, youngs, t)
    assert np.array_equal(Ks.toarray(), K)
    A, bs = oracle.reduce_system(Ks, u_known, u_in, f_in)
    assert np.array_equal(A.rowptr, rp) and np.array_equal(A.col, col) and np.array_equal(A.val, val)
    assert np.array_equal(bs, b)


@pytest.mark.parametrize("case", list(CASES))
def test_closed_form_solution_is_what_the_oracle_solves_for(built, case):
    """K_ff x = b solved in rationals (Gaussian elimination on Fractions): the oracle's CG lands on it"""
    from fractions import Fraction as F
    xy_, tris, mat, u_known_, u_in_, f_in_ = CASES[case]
    Kff, b = cf.reduce(cf.assemble(xy_, tris, **mat), u_known_, u_in_, f_in_)
    n = len(b)
    M = [row[:] + [b[i]] for i, row in enumerate(Kff)]
    for c in range(n):
        p = next(r for r in range(c, n) if M[r][c] != 0)
        M[c], M[p] = M[p], M[c]
        M[c] = [v / M[c][c] for v in M[c]]
        for r in range(n):
            if r != c and M[r][c] != 0:
                M[r] = [a - M[r][c] * d for a, d in zip(M[r], M[c])]
    x = np.array([float(M[r][n]) for r in range(n)])
    xy, conn, nu, youngs, t, u_known, u_in, f_in = flat(case)
    out = oracle.run(xy, conn, u_known, u_in, f_in, youngs, nu, t, path="dense", tol=1e-13)
    u = np.array([float(v) for v in u_in_])
    u[np.array(u_known_) == 0] = x
    assert np.linalg.norm(out["u"] - u) <= 1e-12 * np.linalg.norm(u)
    # reactions by definition (solver.rs:456-469): f_i = K[i, :] . u on the prescribed DOFs
    K = cf.to_float(cf.assemble(xy_, tris, **mat))
    k = np.array(u_known_) == 1
    assert np.abs(out["f"][k] - (K @ u)[k]).max() <= 1e-12 * np.abs(K).max() * np.abs(u).max()


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES))
def test_library_hits_the_closed_form(built, case):
    """mag_element_stiffness, mag_assemble_csr, mag_reduce_system through the C ABI == the hand-derived system"""
    from magnetite_amd import Context
    xy, conn, nu, youngs, t, u_known, u_in, f_in = flat(case)
    K, Kff, b, (rp, col, val) = expected(case)
    with Context(device=0) as ctx:
        ctx.upload(xy, conn, u_known, u_in, f_in, youngs, nu, t)
        ke = ctx.element_stiffness()
        for e, tri in enumerate(CASES[case][1]):
            assert np.array_equal(ke[e], cf.to_float(cf.element_stiffness(CASES[case][0], tri, **CASES[case][2])))
        rowptr, c, v = ctx.assemble_csr()
        assert np.array_equal(dense_of(rowptr, c, v, K.shape[0]), K)
        # structural pattern (solver.rs:299-325 touches every (node, node) pair of an element), ascending columns
        for r in range(K.shape[0]):
            cols = c[rowptr[r]:rowptr[r + 1]]
            assert np.all(np.diff(cols) > 0)
        rp_g, col_g, val_g, b_g = ctx.reduce_system()
        assert np.array_equal(rp_g.astype(np.int64), rp) and np.array_equal(col_g, col) and np.array_equal(val_g, val)
        assert np.array_equal(b_g, b)
