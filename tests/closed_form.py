"""Closed-form (exact rational) plane-stress CST algebra for the hand-derived known-answer tests.

Independent of BOTH oracle/magnetite_oracle.c and magnetite_amd/csrc/exact.hip: no matrix products, no B or D
matrices -- the textbook entry formula of K_e = A t B^T D B written out per 2x2 node block (under-the-hood.md:541-555
states the product, solver.rs:204-250 the entries of B and D):

    beta  = (y1-y2, y2-y0, y0-y1)          gamma = (x2-x1, x0-x2, x1-x0)          A = signed area (solver.rs:187-193)
    c = E / (1 - nu^2)                     h = (1 - nu) / 2
    K_e[2a  ][2b  ] = c t / (4A) * (beta_a beta_b   + h gamma_a gamma_b)
    K_e[2a  ][2b+1] = c t / (4A) * (nu beta_a gamma_b + h gamma_a beta_b)
    K_e[2a+1][2b  ] = c t / (4A) * (nu gamma_a beta_b + h beta_a gamma_b)
    K_e[2a+1][2b+1] = c t / (4A) * (gamma_a gamma_b + h beta_a beta_b)

evaluated in `fractions.Fraction`, scattered by definition (solver.rs:299-325: K[2 n_a + i][2 n_b + j] += ...),
partitioned by definition (solver.rs:365-404: rows = DOFs with known force, columns = DOFs with unknown displacement;
solver.rs:427-432: b = f_known - K_fk u_known) and sparsified by definition (solver.rs:128-136: drop entries == 0).
The KAT meshes use dyadic inputs, so every intermediate is exactly representable in fp64 and ANY correct evaluation
order must reproduce these values bit for bit.
"""
from fractions import Fraction as F

import numpy as np


def element_stiffness(xy, tri, youngs, nu, thick):
    (x0, y0), (x1, y1), (x2, y2) = [xy[i] for i in tri]
    area = F(1, 2) * (x0 * (y1 - y2) + x1 * (y2 - y0) + x2 * (y0 - y1))
    beta = [y1 - y2, y2 - y0, y0 - y1]
    gamma = [x2 - x1, x0 - x2, x1 - x0]
    c, h = youngs / (1 - nu * nu), (1 - nu) / 2
    s = c * thick / (4 * area)
    K = [[F(0)] * 6 for _ in range(6)]
    for a in range(3):
        for b in range(3):
            K[2 * a][2 * b] = s * (beta[a] * beta[b] + h * gamma[a] * gamma[b])
            K[2 * a][2 * b + 1] = s * (nu * beta[a] * gamma[b] + h * gamma[a] * beta[b])
            K[2 * a + 1][2 * b] = s * (nu * gamma[a] * beta[b] + h * beta[a] * gamma[b])
            K[2 * a + 1][2 * b + 1] = s * (gamma[a] * gamma[b] + h * beta[a] * beta[b])
    return K


def assemble(xy, tris, youngs, nu, thick):
    n = 2 * len(xy)
    K = [[F(0)] * n for _ in range(n)]
    for tri in tris:
        ke = element_stiffness(xy, tri, youngs, nu, thick)
        for a in range(3):
            for b in range(3):
                for i in range(2):
                    for j in range(2):
                        K[2 * tri[a] + i][2 * tri[b] + j] += ke[2 * a + i][2 * b + j]
    return K


def reduce(K, u_known, u_in, f_in):
    """(K_ff dense, b) in compact ascending-DOF numbering of the unknown displacements."""
    n = len(K)
    free = [i for i in range(n) if not u_known[i]]
    known = [i for i in range(n) if u_known[i]]
    Kff = [[K[r][c] for c in free] for r in free]
    b = [f_in[r] - sum((K[r][c] * u_in[c] for c in known), F(0)) for r in free]
    return Kff, b


def to_float(M):
    a = np.array([[float(v) for v in row] for row in M]) if M and isinstance(M[0], list) else np.array([float(v) for v in M])
    return a


def exactly_representable(M):
    """every Fraction in M is a dyadic rational that fp64 holds exactly"""
    flat = [v for row in M for v in row] if M and isinstance(M[0], list) else list(M)
    return all(F(float(v)) == v for v in flat)


def csr_of_dense(A):
    """solver.rs:126-137: row-major scan, entries != 0 kept, ascending columns."""
    rowptr, col, val = [0], [], []
    for row in A:
        for c, v in enumerate(row):
            if v != 0:
                col.append(c)
                val.append(float(v))
        rowptr.append(len(col))
    return np.array(rowptr, dtype=np.int64), np.array(col, dtype=np.int32), np.array(val)


# ---------------------------------------------------------------------------------------------------------------
# KAT-SQ2: the unit square cut along the diagonal into two CCW triangles, E = 1, nu = 0, t = 1.
#   nodes 0 (0,0)  1 (1,0)  2 (1,1)  3 (0,1);  elements (0,1,2), (0,2,3); both areas 1/2, so c t/(4A) = 1/2, h = 1/2.
#   element (0,1,2): beta = (-1, 1, 0), gamma = (0, -1, 1);  element (0,2,3): beta = (0, 1, -1), gamma = (-1, 0, 1).
# Worked by hand from the block formula above, e.g.
#   K[0][0] = 1/2 (1 + 0) + 1/2 (0 + 1/2)            = 3/4      (both elements touch node 0)
#   K[0][1] = 1/2 (0 + 1/2 * 0 * -1) + 1/2 (0 + 1/2 * -1 * 0) = 0   (structurally present, numerically zero)
#   K[2][4] = element (0,1,2) only, a=1, b=2: 1/2 (1*0 + 1/2 * -1 * 1) = -1/4
#   K[0][4] = (0,1,2): 1/2 (-1*0 + 1/2*0*1) = 0  plus (0,2,3) with a=0, b=1: 1/2 (0*1 + 1/2*-1*0) = 0
SQ2_XY = [(F(0), F(0)), (F(1), F(0)), (F(1), F(1)), (F(0), F(1))]
SQ2_TRIS = [(0, 1, 2), (0, 2, 3)]
SQ2_MATERIAL = dict(youngs=F(1), nu=F(0), thick=F(1))
SQ2_K = [
    [.75, 0, -.5, 0, 0, -.25, -.25, .25],
    [0, .75, .25, -.25, -.25, 0, 0, -.5],
    [-.5, .25, .75, -.25, -.25, 0, 0, 0],
    [0, -.25, -.25, .75, .25, -.5, 0, 0],
    [0, -.25, -.25, .25, .75, 0, -.5, 0],
    [-.25, 0, 0, -.5, 0, .75, .25, -.25],
    [-.25, 0, 0, 0, -.5, .25, .75, -.25],
    [.25, -.5, 0, 0, 0, -.25, -.25, .75],
]
# boundary set: node 0 pinned (ux = uy = 0), node 3 on a roller (ux = 0, fy = 0), node 1 loaded (fx = 2, fy = 0),
# node 2 pulled (ux = 1/2 prescribed, fy = -1).  Known displacements: DOFs 0, 1, 4, 6; unknown: 2, 3, 5, 7.
SQ2_U_KNOWN = [1, 1, 0, 0, 1, 0, 1, 0]
SQ2_U_IN = [F(0), F(0), F(0), F(0), F(1, 2), F(0), F(0), F(0)]
SQ2_F_IN = [F(0), F(0), F(2), F(0), F(0), F(-1), F(0), F(0)]
# K_ff = rows/cols (2, 3, 5, 7) of SQ2_K;  b = f - K[:, 4] * 1/2 on those rows:
#   b[2] = 2 - (-1/4)(1/2) = 17/8,  b[3] = 0 - (1/4)(1/2) = -1/8,  b[5] = -1 - 0 = -1,  b[7] = 0 - 0 = 0
SQ2_KFF = [
    [.75, -.25, 0, 0],
    [-.25, .75, -.5, 0],
    [0, -.5, .75, -.25],
    [0, 0, -.25, .75],
]
SQ2_B = [2.125, -0.125, -1.0, 0.0]
# after dropping the exact zeros (solver.rs:131): tridiagonal
SQ2_KFF_ROWPTR = [0, 2, 5, 8, 10]
SQ2_KFF_COL = [0, 1, 0, 1, 2, 1, 2, 3, 2, 3]
SQ2_KFF_VAL = [.75, -.25, -.25, .75, -.5, -.5, .75, -.25, -.25, .75]


# ---------------------------------------------------------------------------------------------------------------
# KAT-G8: 3 x 3 nodes on a grid of pitch 2, eight CCW triangles (each cell cut along its rising diagonal),
#   E = 15, nu = 1/4, t = 1/2  =>  c = 15 / (15/16) = 16, h = 3/8, every area 2, c t/(4A) = 1: all entries are small
#   integers or halves.  Node id = 3 * row + column.  D = [[16, 4, 0], [4, 16, 0], [0, 0, 6]] (solver.rs:240-250).
G8_XY = [(F(2 * i), F(2 * j)) for j in range(3) for i in range(3)]
G8_TRIS = []
for _j in range(2):
    for _i in range(2):
        _n00 = 3 * _j + _i
        G8_TRIS += [(_n00, _n00 + 1, _n00 + 4), (_n00, _n00 + 4, _n00 + 3)]
G8_MATERIAL = dict(youngs=F(15), nu=F(1, 4), thick=F(1, 2))
# boundary set in the reference's pattern (examples/tensile-example/input.json:10-33): left column fixed, right
# column ux = 1/4 with fy = 0, one interior load.
G8_U_KNOWN = [0] * 18
G8_U_IN = [F(0)] * 18
G8_F_IN = [F(0)] * 18
for _n in (0, 3, 6):
    G8_U_KNOWN[2 * _n] = G8_U_KNOWN[2 * _n + 1] = 1
for _n in (2, 5, 8):
    G8_U_KNOWN[2 * _n] = 1
    G8_U_IN[2 * _n] = F(1, 4)
G8_F_IN[2 * 4 + 1] = F(-3)
# Hand-checked entries of the assembled K (node 4 is the interior node, touched by six triangles):
#   the six triangles at node 4 contribute beta_a^2 + h gamma_a^2 with (beta_a, gamma_a) =
#   (2,-2)->4+3/2, (0,2)->3/2, (-2,0)->4, (-2,2)->4+3/2, (0,-2)->3/2, (2,0)->4   => K[8][8] = 22
#   and gamma_a^2 + h beta_a^2: 4+3/2, 4, 3/2, 4+3/2, 4, 3/2                     => K[9][9] = 22
#   K[8][9] = sum of (nu + h) beta_a gamma_a = 5/8 * (-4 + 0 + 0 - 4 + 0 + 0)     = -5
G8_SPOT = {(8, 8): 22.0, (9, 9): 22.0, (8, 9): -5.0, (9, 8): -5.0}
