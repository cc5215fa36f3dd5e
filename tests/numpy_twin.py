"""Independent numpy restatement of src/solver.rs (dense, textbook order) used only to cross-check the C oracle.

Written separately from oracle/magnetite_oracle.c on purpose: matrix products go through numpy (different
summation order), the solve goes through a direct factorisation -- agreement to round-off with the oracle's
element-ordered scatter and CG is evidence that both restate the same algorithm.
"""
import numpy as np


def element_area(xy, tri):  # solver.rs:187-193
    (x0, y0), (x1, y1), (x2, y2) = xy[tri[0]], xy[tri[1]], xy[tri[2]]
    return 0.5 * (x0 * (y1 - y2) + x1 * (y2 - y0) + x2 * (y0 - y1))


def strain_displacement(xy, tri, area):  # solver.rs:204-230
    (x0, y0), (x1, y1), (x2, y2) = xy[tri[0]], xy[tri[1]], xy[tri[2]]
    b = [y1 - y2, y2 - y0, y0 - y1]
    g = [x2 - x1, x0 - x2, x1 - x0]
    B = np.array([[b[0], 0, b[1], 0, b[2], 0], [0, g[0], 0, g[1], 0, g[2]], [g[0], b[0], g[1], b[1], g[2], b[2]]],
                 dtype=np.float64)
    return B / (2.0 * area)


def stress_strain(nu, youngs):  # solver.rs:240-250
    return np.array([[1, nu, 0], [nu, 1, 0], [0, 0, (1 - nu) / 2]], dtype=np.float64) * (youngs / (1 - nu ** 2))


def element_stiffness(xy, tri, nu, youngs, t):  # solver.rs:263-278
    A = element_area(xy, tri)
    B = strain_displacement(xy, tri, A)
    return (B.T @ stress_strain(nu, youngs)) @ B * A * t


def total_stiffness(xy, conn, nu, youngs, t):  # solver.rs:290-331
    n = 2 * xy.shape[0]
    K = np.zeros((n, n))
    for tri in conn:
        ke = element_stiffness(xy, tri, nu, youngs, t)
        dofs = np.array([[2 * i, 2 * i + 1] for i in tri]).reshape(-1)
        K[np.ix_(dofs, dofs)] += ke
    return K


def solve(xy, conn, u_known, u_in, f_in, youngs, nu, t):
    """solver.rs:412-487 with a direct solve in place of CG, then solver.rs:496-535."""
    K = total_stiffness(xy, conn, nu, youngs, t)
    free = np.where(u_known == 0)[0]
    known = np.where(u_known == 1)[0]
    b = f_in[free] - K[np.ix_(free, known)] @ u_in[known]  # solver.rs:365-404,427-432
    u = np.array(u_in, dtype=np.float64)
    u[free] = np.linalg.solve(K[np.ix_(free, free)], b)
    f = np.array(f_in, dtype=np.float64)
    f[known] = K[known] @ u  # solver.rs:456-469
    stress = np.empty(len(conn))
    D = stress_strain(nu, youngs)
    for e, tri in enumerate(conn):
        ue = np.array([[u[2 * i], u[2 * i + 1]] for i in tri]).reshape(-1)
        s = D @ strain_displacement(xy, tri, element_area(xy, tri)) @ ue
        stress[e] = np.hypot(s[0], s[1]) * (-1.0 if s[0] + s[1] < 1.0 else 1.0)  # the `< 1.0` quirk, solver.rs:524-530
    return dict(K=K, b=b, u=u, f=f, stress=stress, free=free, known=known)


def argmin_cg(A, b, target, max_iter, squared=False):
    """argmin 0.10 `ConjugateGradient` under `Executor`, as the reference configures it (solver.rs:139-176), restated from
    the published algorithm with numpy vector operations: x0 = 0, r0 = -(b - A x0), p0 = -r0, and per iteration
    q = A p, alpha = r.r / p.q, x += alpha p, r += alpha q, beta = r'.r' / r.r, p = -r' + beta p, cost = |r'| (or r'.r').
    The executor stops once best_cost <= target or after max_iter iterations and returns best_param, the lowest-cost
    iterate.  Returns (best_param, iterations, cost history)."""
    x = np.zeros_like(b)
    r = -(b - A @ x)
    p = -r
    rtr = float(r @ r)
    best, x_best, hist = np.inf, x.copy(), []
    it = 0
    while it < max_iter and not best <= target:
        q = A @ p
        alpha = rtr / float(p @ q)
        x = x + alpha * p
        r = r + alpha * q
        rtr_new = float(r @ r)
        p = -r + (rtr_new / rtr) * p
        rtr = rtr_new
        cost = abs(rtr) if squared else np.sqrt(rtr)
        hist.append(cost)
        it += 1
        if cost < best:
            best, x_best = cost, x.copy()
    return x_best, it, np.array(hist)
