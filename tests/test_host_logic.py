"""CPU: host-side logic -- boundary-rule stamping (mesher.rs:815-930 semantics), flattening of the
Node/Element/Option model (datatypes.rs:1-29), synthetic mesh generators."""
import numpy as np
import pytest

from magnetite_amd import Element, MagnetiteError, Node, Vertex, meshgen
from magnetite_amd.solver import flatten


def areas(m):
    a = m.xy[m.conn]
    return 0.5 * ((a[:, 1, 0] - a[:, 0, 0]) * (a[:, 2, 1] - a[:, 0, 1]) - (a[:, 2, 0] - a[:, 0, 0]) * (a[:, 1, 1] - a[:, 0, 1]))


def test_region_test_is_strict_and_later_rules_win():
    m = meshgen.plate(4, 4, 4.0, 4.0)  # nodes on integer coordinates 0..4
    rules = [meshgen.BoundaryRule("a", x_min=0.0, x_max=2.0, ux=1.0, uy=2.0),       # strict: x in (0,2) => x == 1 only
             meshgen.BoundaryRule("b", x_min=0.5, x_max=1.5, y_min=2.5, fx=5.0, uy=7.0)]  # overrides a on x==1, y>2.5
    p = meshgen.apply_boundary_rules(m, rules)
    k = p.u_known.reshape(-1, 2)
    x, y = m.xy[:, 0], m.xy[:, 1]
    on_a = (x == 1.0)
    on_b = on_a & (y > 2.5)
    assert np.array_equal(k[:, 0] == 1, on_a & ~on_b)       # b sets fx => ux no longer prescribed there
    assert np.array_equal(k[:, 1] == 1, on_a)
    assert np.all(p.u_in.reshape(-1, 2)[on_b][:, 1] == 7.0) and np.all(p.f_in.reshape(-1, 2)[on_b][:, 0] == 5.0)
    assert np.all(p.u_in.reshape(-1, 2)[on_a & ~on_b] == [1.0, 2.0])
    # defaults elsewhere: ux=uy=None, fx=fy=Some(0.0) (mesher.rs:615-624)
    assert not k[~on_a].any() and not p.f_in.reshape(-1, 2)[~on_a].any()
    # nodes exactly on a region edge are NOT selected (mesher.rs:915-918 uses > and <)
    assert not k[x == 0.0].any() and not k[x == 2.0].any()


@pytest.mark.parametrize("kw,msg", [
    (dict(x_min=2, x_max=1, ux=0.0, uy=0.0), "x_target_min greater than x_target_max"),
    (dict(y_min=2, y_max=1, ux=0.0, uy=0.0), "y_target_min greater than y_target_max"),
    (dict(uy=0.0), "under-constrained in x-axis"),
    (dict(ux=0.0), "under-constrained in y-axis"),
    (dict(ux=0.0, fx=1.0, uy=0.0), "over-constrained in x-axis"),
    (dict(ux=0.0, uy=0.0, fy=1.0), "over-constrained in y-axis"),
])
def test_rule_validation_messages(kw, msg):
    """mesher.rs:871-900"""
    with pytest.raises(ValueError, match=msg):
        meshgen.apply_boundary_rules(meshgen.plate(2), [meshgen.BoundaryRule("r", **kw)])


def test_flatten_options_to_soa_and_back():
    nodes = [Node(Vertex(0.0, 0.0), ux=0.0, uy=0.0, fx=None, fy=None), Node(Vertex(1.0, 0.0)),
             Node(Vertex(0.0, 1.0), ux=None, uy=0.5, fx=3.0, fy=None)]
    els = [Element([0, 1, 2])]
    xy, conn, k, u, f = flatten(nodes, els)
    assert xy.tolist() == [0, 0, 1, 0, 0, 1] and conn.tolist() == [0, 1, 2] and conn.dtype == np.int32
    assert k.tolist() == [1, 1, 0, 0, 0, 1] and u.tolist() == [0, 0, 0, 0, 0, 0.5] and f.tolist() == [0, 0, 0, 0, 3, 0]
    nodes[1].ux = 1.0  # both ux and fx set: solver.rs:431 would panic
    with pytest.raises(MagnetiteError, match="Solver error"):
        flatten(nodes, els)
    nodes[1].ux, nodes[1].fx = None, None  # neither
    with pytest.raises(MagnetiteError):
        flatten(nodes, els)


def test_generators():
    m = meshgen.plate(224)
    assert m.num_elements == 100352 and m.num_nodes == 50625  # BASELINE config 2 / SURVEY 8
    assert np.all(areas(m) > 0)
    h = meshgen.plate_with_holes(40)
    assert np.all(areas(h) > 0) and h.num_elements < 2 * 40 * 40
    assert h.conn.max() == h.num_nodes - 1 and len(np.unique(h.conn)) == h.num_nodes  # no orphan nodes
    c = h.xy[h.conn].mean(axis=1)
    assert np.all((c[:, 0] - 0.5) ** 2 + (c[:, 1] - 0.5) ** 2 > 0.14 ** 2)
    n = meshgen.grid_for_triangles(1e6, np.pi * 0.15 ** 2)
    assert abs(2 * n * n * (1 - np.pi * 0.15 ** 2) - 1e6) < 5e3
    mh = meshgen.multi_hole(64, 4, 0.25)
    assert np.all(areas(mh) > 0) and mh.num_elements < 2 * 64 * 64
    s = meshgen.shuffle(h, 9)
    assert np.allclose(np.sort(areas(s)), np.sort(areas(h)), rtol=0, atol=1e-18)
    assert np.all(areas(meshgen.perturb(h, 0.2)) > 0)
    assert np.all(areas(meshgen.clockwise(h)) < 0)
    # check_ccw quirk (mesher.rs:522-526): area < 1.0 => reversed, so a fine CCW mesh comes out clockwise
    assert np.all(areas(meshgen.check_ccw(h)) < 0)
    big = meshgen.plate(2, 2, 10.0, 10.0)  # areas 12.5 >= 1 stay CCW
    assert np.all(areas(meshgen.check_ccw(big)) > 0)


def test_baseline_configs_bc_counts():
    p = meshgen.config_fixed_left_point_load(meshgen.plate(10))
    assert p.u_known.sum() == 2 * 11 and np.count_nonzero(p.f_in) == 1
    q = meshgen.config_fixed_left_pull_right(meshgen.plate(10))
    assert q.u_known.sum() == 2 * 11 + 11
    k = q.u_known.reshape(-1, 2)
    right = q.mesh.xy[:, 0] > 1 - 1e-9
    assert np.all(k[right] == [1, 0]) and np.all(q.u_in.reshape(-1, 2)[right][:, 0] == 1e-3)


def test_polygon_mesher_is_deterministic_and_valid():
    import os
    outline = np.loadtxt(os.path.join(os.path.dirname(__file__), "golden", "tensile_outline.csv"), delimiter=",", skiprows=1)
    a, b = meshgen.polygon_mesh(outline, 0.6), meshgen.polygon_mesh(outline, 0.6)
    assert np.array_equal(a.xy, b.xy) and np.array_equal(a.conn, b.conn)
    assert np.all(areas(a) > 0)
    poly_area = 0.5 * abs(np.dot(outline[:, 0], np.roll(outline[:, 1], -1)) - np.dot(outline[:, 1], np.roll(outline[:, 0], -1)))
    assert areas(a).sum() == pytest.approx(poly_area, rel=1e-9)


def test_frontal_like_mesh_is_deterministic_valid_and_has_gmsh_like_valences():
    """meshgen.frontal_like, the stand-in for what gmsh's frontal mesher hands solver::run (mesher.rs:501-506; round 4):
    deterministic for (n, jitter, seed), conforming and CCW, every node used, and at jitter 0.4 a valence histogram with about a
    quarter of the nodes at five, half at six, a quarter at seven or more -- the rows the on-chip kernel's overflow edge blocks
    exist for.  Its oracle solution is a valid CST solve (patch-like pull: ux grows along x)."""
    a, b = meshgen.frontal_like(60, 0.4, 7), meshgen.frontal_like(60, 0.4, 7)
    assert np.array_equal(a.xy, b.xy) and np.array_equal(a.conn, b.conn)
    assert not np.array_equal(a.xy, meshgen.frontal_like(60, 0.4, 8).xy)
    tri = a.xy[a.conn]
    area = 0.5 * ((tri[:, 1, 0] - tri[:, 0, 0]) * (tri[:, 2, 1] - tri[:, 0, 1]) -
                  (tri[:, 2, 0] - tri[:, 0, 0]) * (tri[:, 1, 1] - tri[:, 0, 1]))
    assert np.all(area > 0) and abs(area.sum() - 1.0) < 1e-12  # CCW, and the triangles tile the unit square
    assert np.array_equal(np.unique(a.conn), np.arange(a.num_nodes))
    # conforming: every interior edge is shared by exactly two triangles, boundary edges by one
    e = np.sort(np.concatenate([a.conn[:, [0, 1]], a.conn[:, [1, 2]], a.conn[:, [2, 0]]]), axis=1)
    _, cnt = np.unique(e, axis=0, return_counts=True)
    assert set(cnt.tolist()) <= {1, 2}
    val = np.bincount(a.conn.reshape(-1), minlength=a.num_nodes)
    frac = np.bincount(val, minlength=12) / a.num_nodes
    assert 0.15 < frac[5] < 0.32 and 0.40 < frac[6] < 0.60 and 0.18 < frac[7:].sum() < 0.32 and val.max() <= 12
    p = meshgen.baseline_problem("frontal1m", scale=0.05)
    assert p.mesh.num_elements > 2000 and int((p.u_known == 1).sum()) > 50
