"""CPU: the oracle against hand-derivable known answers.  The reference has no tests or golden
vectors of its own -- PARITY UNPINNED -- so these closed forms, derived from src/solver.rs:187-278, are the pins."""
import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

import oracle

I3 = np.arange(3, dtype=np.int32)


def ke(xy, tri=I3, nu=0.0, E=1.0, t=1.0):
    return oracle.element_stiffness_all(np.asarray(xy, dtype=np.float64), np.asarray(tri, dtype=np.int32), nu, E, t)[0]


def test_kat1_unit_right_triangle(built):
    xy = [[0, 0], [1, 0], [0, 1]]
    assert oracle.element_area(np.array(xy, dtype=float), I3) == 0.5
    B = oracle.strain_displacement(np.array(xy, dtype=float), I3, 0.5)
    assert np.array_equal(B, [[-1, 0, 1, 0, 0, 0], [0, -1, 0, 0, 0, 1], [-1, -1, 0, 1, 1, 0]])
    K = ke(xy)
    want = [[.75, .25, -.5, -.25, -.25, 0], [.25, .75, 0, -.25, -.25, -.5], [-.5, 0, .5, 0, 0, 0],
            [-.25, -.25, 0, .25, .25, 0], [-.25, -.25, 0, .25, .25, 0], [0, -.5, 0, 0, 0, .5]]
    assert np.array_equal(K, want)


def test_kat1cw_signed_area_makes_ke_negative(built):
    """mesher.rs:522-526 hands clockwise elements to the solver on fine meshes; the signed area (solver.rs:192)
    then flips the sign of K_e.  The hot path must not 'fix' this."""
    xy = np.array([[0, 0], [1, 0], [0, 1]], dtype=float)
    cw = np.array([0, 2, 1], dtype=np.int32)
    assert oracle.element_area(xy, cw) == -0.5
    K = ke(xy, cw)
    assert K[0, 0] == -0.75
    assert np.all(np.linalg.eigvalsh(K) <= 1e-15)
    # node-permuted negative of KAT-1
    perm = [0, 1, 4, 5, 2, 3]
    assert np.array_equal(K, -ke(xy)[np.ix_(perm, perm)])


def test_kat2_aluminium(built):
    xy = [[0, 0], [2, 0], [0, 1]]
    D = oracle.stress_strain(0.33, 69e9)
    assert D[0, 0] == 7.7432386937492981e10 and D[0, 1] == 2.5552687689372684e10 and D[2, 2] == 2.5939849624060146e10
    assert D[0, 2] == 0 and D[2, 0] == 0 and D[1, 1] == D[0, 0]
    K = ke(xy, nu=0.33, E=69e9, t=0.5)
    assert K[0, 0] == 2.2648973179216698e10
    assert K[1, 1] == 4.1958674671754005e10
    assert K[0, 1] == 1.2873134328358208e10
    assert np.abs(K - K.T).max() <= 1e-6 * np.abs(K).max() * 1e-9
    assert np.abs(K.sum(axis=1)).max() <= 1e-12 * np.abs(K).max()


coords = st.floats(min_value=-50.0, max_value=50.0, allow_nan=False, allow_infinity=False)


@settings(max_examples=200, deadline=None)
@given(st.lists(coords, min_size=6, max_size=6), st.floats(0.0, 0.49), st.floats(1.0, 1e11), st.floats(0.01, 10.0))
def test_ke_properties(built, c, nu, E, t):
    xy = np.array(c).reshape(3, 2)
    A = oracle.element_area(xy, I3)
    if abs(A) < 1e-3:
        return
    K = ke(xy, nu=nu, E=E, t=t)
    s = np.abs(K).max()
    assert np.abs(K - K.T).max() <= 1e-9 * s           # symmetric
    assert np.abs(K.sum(axis=1)).max() <= 1e-9 * s     # rigid translation in x+y together ...
    assert np.abs(K[:, 0::2].sum(axis=1)).max() <= 1e-9 * s  # ... and in x and y separately
    assert np.abs(K[:, 1::2].sum(axis=1)).max() <= 1e-9 * s
    # reversing orientation negates K_e (up to the node permutation)
    Kr = ke(xy, np.array([0, 2, 1], dtype=np.int32), nu=nu, E=E, t=t)
    perm = [0, 1, 4, 5, 2, 3]
    assert np.abs(Kr + K[np.ix_(perm, perm)]).max() <= 1e-9 * s
    # definiteness follows the sign of the signed area
    ev = np.linalg.eigvalsh(0.5 * (K + K.T)) * np.sign(A)
    assert ev.min() >= -1e-9 * s


def test_stress_sign_quirk(built):
    """solver.rs:524-530: sign = -1 when sx+sy < 1.0 (not < 0.0)."""
    xy = np.array([[0, 0], [1, 0], [0, 1]], dtype=float)
    conn = np.array([0, 1, 2], dtype=np.int32)
    # uniform strain eps_x = e: u = (e x, 0); E=1, nu=0 => sx = e, sy = 0
    for e, sign in ((0.5, -1.0), (0.999, -1.0), (1.0, 1.0), (2.0, 1.0), (-2.0, -1.0)):
        u = np.array([0, 0, e, 0, 0, 0], dtype=float)
        s = oracle.stress(xy, conn, u, 0.0, 1.0)
        assert s[0] == pytest.approx(sign * abs(e), rel=1e-15)
