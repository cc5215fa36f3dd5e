"""CPU check of the algebra behind the on-chip CG kernel's edge blocks (magnetite_amd/csrc/cg_device.h: ring_blocks,
ring_walk_blocks): a triangle's force on its corner a is linear in the other corners' values relative to a; folded per ring
entry the node's fan becomes one 2 x 2 block per neighbour; only the SYMMETRIC part of a block is kept, because the
antisymmetric part of one triangle's contribution is -kappa J for the entry before it and +kappa J for the entry after it,
kappa = (h - nu) c0 / 2, whatever the triangle's shape -- so over a fan they telescope to kappa J (u_last - u_first).
numpy restatement of fan_force_w (the triangle walk's arithmetic, solver.rs:204-250 behind it) against the block form, on
random closed and open fans."""
import numpy as np
import pytest

NU, C0 = 0.29, 3.7e9
H = (1.0 - NU) / 2.0
KAPPA = 0.5 * (H - NU) * C0


def fan_force_w(db, ub, dc, uc, wt):
    """cg_device.h fan_force_w: force on a of the triangle (a, b, c), everything relative to a; wt = c0 / (2A)."""
    ba, ga = db[1] - dc[1], dc[0] - db[0]
    ex = dc[1] * ub[0] - db[1] * uc[0]
    ey = db[0] * uc[1] - dc[0] * ub[1]
    g = (dc[1] * ub[1] - dc[0] * ub[0]) + (db[0] * uc[0] - db[1] * uc[1])
    sx, sy, tq = ex + NU * ey, NU * ex + ey, H * g
    return np.array([wt * (ba * sx + ga * tq), wt * (ga * sy + ba * tq)])


def random_fan(rng, triangles, closed):
    """neighbour positions relative to the centre, counter-clockwise; a closed fan repeats the first at the end"""
    span = 2 * np.pi if closed else rng.uniform(0.5, 1.2) * np.pi
    n = triangles if closed else triangles + 1
    ang = np.sort(rng.uniform(0, span, n)) if not closed else np.linspace(0, span, n, endpoint=False) + rng.uniform(-0.2, 0.2, n)
    pts = [rng.uniform(0.5, 1.5) * np.array([np.cos(a), np.sin(a)]) for a in np.sort(ang)]
    if closed:
        pts.append(pts[0])
    return pts


def blocks_of(pts):
    """ring_blocks: per entry the symmetric part of K_ab (triangle after the entry) + K_ac (triangle before it), from unit
    vectors through fan_force_w; and the antisymmetric coefficients it drops"""
    n = len(pts)
    kb, anti = np.zeros((n, 3)), np.zeros(n)
    z, e0, e1 = np.zeros(2), np.array([1.0, 0.0]), np.array([0.0, 1.0])
    for k in range(1, n):
        two_a = pts[k - 1][0] * pts[k][1] - pts[k][0] * pts[k - 1][1]
        wt = C0 / two_a
        b0, b1 = fan_force_w(pts[k - 1], e0, pts[k], z, wt), fan_force_w(pts[k - 1], e1, pts[k], z, wt)
        c0, c1 = fan_force_w(pts[k - 1], z, pts[k], e0, wt), fan_force_w(pts[k - 1], z, pts[k], e1, wt)
        kb[k - 1] += [b0[0], 0.5 * (b1[0] + b0[1]), b1[1]]
        kb[k] += [c0[0], 0.5 * (c1[0] + c0[1]), c1[1]]
        anti[k - 1] += 0.5 * (b1[0] - b0[1])
        anti[k] += 0.5 * (c1[0] - c0[1])
    return kb, anti


@pytest.mark.parametrize("closed", [True, False])
@pytest.mark.parametrize("triangles", [3, 4, 5, 6])
def test_blocks_reproduce_the_triangle_walk(closed, triangles):
    rng = np.random.default_rng(100 * triangles + closed)
    for _ in range(20):
        pts = random_fan(rng, triangles, closed)
        us = [rng.normal(size=2) for _ in pts]
        if closed:
            us[-1] = us[0]
        walk = np.zeros(2)
        for k in range(1, len(pts)):
            two_a = pts[k - 1][0] * pts[k][1] - pts[k][0] * pts[k - 1][1]
            walk += fan_force_w(pts[k - 1], us[k - 1], pts[k], us[k], C0 / two_a)
        kb, anti = blocks_of(pts)
        # one triangle: -kappa for the entry before it, +kappa for the entry after it, whatever its shape
        expect = np.zeros(len(pts))
        expect[:-1] -= KAPPA
        expect[1:] += KAPPA
        assert np.allclose(anti, expect, rtol=1e-12, atol=1e-12 * abs(KAPPA))
        f = np.zeros(2)
        for j, u in enumerate(us):
            f += [kb[j, 0] * u[0] + kb[j, 1] * u[1], kb[j, 1] * u[0] + kb[j, 2] * u[1]]
        a = us[-1] - us[0]  # telescoped antisymmetric parts: kappa J (u_last - u_first); zero for a closed fan
        f += [KAPPA * a[1], -KAPPA * a[0]]
        assert np.allclose(f, walk, rtol=1e-11, atol=1e-11 * np.abs(walk).max())
        if closed:
            assert not a.any()


def test_closing_triangle_folds_onto_entry_zero():
    """A closed fan of valence 6 is seven ring entries, the last the first node again: its block added to block 0 leaves six
    blocks and nothing to telescope -- what k_edge_blocks stores for an interior node of a structured mesh."""
    rng = np.random.default_rng(7)
    pts = random_fan(rng, 6, True)
    us = [rng.normal(size=2) for _ in pts]
    us[-1] = us[0]
    kb, _ = blocks_of(pts)
    folded = kb[:6].copy()
    folded[0] += kb[6]
    f6 = sum(np.array([folded[j, 0] * us[j][0] + folded[j, 1] * us[j][1], folded[j, 1] * us[j][0] + folded[j, 2] * us[j][1]])
             for j in range(6))
    walk = np.zeros(2)
    for k in range(1, 7):
        two_a = pts[k - 1][0] * pts[k][1] - pts[k][0] * pts[k - 1][1]
        walk += fan_force_w(pts[k - 1], us[k - 1], pts[k], us[k], C0 / two_a)
    assert np.allclose(f6, walk, rtol=1e-11, atol=1e-11 * np.abs(walk).max())


def overflow_layout(pts, nb=6):
    """k_edge_blocks_ovf's placement (round 4): a row of n ring entries is streamed one triangle at a time -- block j is
    K_ac of triangle j + K_ab of triangle j + 1 --; a CLOSED fan folds its closing triangle into blocks n - 2 and 0 (n - 1
    blocks); blocks below nb stay in registers, the rest become pool records (block, ring entry); an OPEN fan of more than nb
    entries keeps its LAST block in register position nb - 1 and its middle blocks nb - 1 .. n - 2 in the pool, so that u_first
    and u_last are register entries.  Returns (register blocks with their entries, pool records with their entries, closed)."""
    n = len(pts)
    closed = n >= 3 and np.allclose(pts[-1], pts[0])
    kb, _ = blocks_of(pts)
    if closed:  # the closing triangle's K_ac lands in block n - 1 of blocks_of: fold it onto entry 0
        kb = kb.copy()
        kb[0] += kb[n - 1]
        nblk = n - 1
    else:
        nblk = n
    regs, pool = [(np.zeros(3), 0)] * nb, []
    regs = list(regs)
    open_long = not closed and n > nb
    for j in range(nblk):
        dest = j
        if open_long and j >= nb - 1:
            dest = nb - 1 if j == n - 1 else j + 1
        if dest < nb:
            regs[dest] = (kb[j], j)
        else:
            pool.append((dest - nb, kb[j], j))
    pool = [(b, e) for _, b, e in sorted(pool, key=lambda t: t[0])]
    return regs, pool, closed


@pytest.mark.parametrize("closed", [True, False])
@pytest.mark.parametrize("triangles", [2, 5, 6, 7, 8, 9, 12])
def test_overflow_layout_reproduces_the_triangle_walk(closed, triangles):
    """Rows of ANY length through six register blocks + pool records (ring_walk_blocks_ovf): the register blocks, the
    telescoped antisymmetric part from register entries 0 and 5 only, then the pool records in order -- equal to the triangle
    walk for closed fans of valence 3-12 and open fans of up to 12 triangles."""
    if closed and triangles < 3:
        pytest.skip("a closed fan has at least three triangles")
    rng = np.random.default_rng(1000 * triangles + closed)
    for _ in range(10):
        pts = random_fan(rng, triangles, closed)
        us = [rng.normal(size=2) for _ in pts]
        if closed:
            us[-1] = us[0]
        walk = np.zeros(2)
        for k in range(1, len(pts)):
            two_a = pts[k - 1][0] * pts[k][1] - pts[k][0] * pts[k - 1][1]
            walk += fan_force_w(pts[k - 1], us[k - 1], pts[k], us[k], C0 / two_a)
        regs, pool, is_closed = overflow_layout(pts)
        assert is_closed == closed
        n = len(pts)
        nblk = n - 1 if closed else n
        assert len(pool) == max(0, nblk - 6)
        # register entries: blocks the row does not have are zero blocks on a repeated entry (the last one)
        ent = [e for _, e in regs]
        for j in range(6):
            if j >= min(nblk, 6) and not (not closed and n > 6):
                ent[j] = min(n - 1, nblk - 1) if nblk > 0 else 0
        f = np.zeros(2)
        for (b, _), e in zip(regs, ent):
            u = us[e]
            f += [b[0] * u[0] + b[1] * u[1], b[1] * u[0] + b[2] * u[1]]
        if not closed:  # kappa J (u_last - u_first), both from the registers: entry 0 and register entry 5
            a = us[ent[5]] - us[ent[0]]
            assert ent[5] == n - 1 or n <= 6
            if n <= 6:
                a = us[n - 1] - us[0]
            f += [KAPPA * a[1], -KAPPA * a[0]]
        for b, e in pool:
            u = us[e]
            f += [b[0] * u[0] + b[1] * u[1], b[1] * u[0] + b[2] * u[1]]
        assert np.allclose(f, walk, rtol=1e-10, atol=1e-10 * np.abs(walk).max())
