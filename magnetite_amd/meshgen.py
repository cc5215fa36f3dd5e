"""Synthetic CST meshes + boundary-rule stamping for the BASELINE.json configs.

The reference meshes with an external `gmsh` process (mesher.rs:481-506), which
this image does not have; BASELINE.json's configs 2-5 are synthetic anyway.
Generators are deterministic (no RNG unless `shuffle_seed` is given) and emit
the flat arrays of include/magnetite_hip.h: xy[2N] f64, conn[3E] i32.

Boundary stamping follows mesher.rs:
  * defaults ux=uy=None, fx=fy=Some(0.0)              (mesher.rs:615-624)
  * region test is STRICT on all four sides            (mesher.rs:915-918)
  * every matching rule overwrites all four targets, later rules win
                                                       (mesher.rs:920-925)
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

# examples/*/input.json: aluminium, all three examples use these values
ALU = dict(youngs_modulus=69e9, poisson_ratio=0.33, part_thickness=0.5)


@dataclass
class Mesh:
    xy: np.ndarray    # (N,2) f64
    conn: np.ndarray  # (E,3) i32, counter-clockwise (positive signed area)
    name: str = "mesh"

    @property
    def num_nodes(self):
        return self.xy.shape[0]

    @property
    def num_elements(self):
        return self.conn.shape[0]


@dataclass
class BoundaryRule:
    """datatypes.rs:31-52 BoundaryRegion + BoundaryTarget (None == json null)."""
    name: str
    x_min: float = -np.finfo(np.float64).max  # f64::MIN (mesher.rs:838-841)
    x_max: float = np.finfo(np.float64).max
    y_min: float = -np.finfo(np.float64).max
    y_max: float = np.finfo(np.float64).max
    ux: Optional[float] = None
    uy: Optional[float] = None
    fx: Optional[float] = None
    fy: Optional[float] = None

    def validate(self):
        """mesher.rs:871-900; messages match the reference's Input errors."""
        if self.x_min > self.x_max:
            raise ValueError(f"Input error: Boundary '{self.name}' has x_target_min greater than x_target_max")
        if self.y_min > self.y_max:
            raise ValueError(f"Input error: Boundary '{self.name}' has y_target_min greater than y_target_max")
        for ax, f, u in (("x", self.fx, self.ux), ("y", self.fy, self.uy)):
            if f is None and u is None:
                raise ValueError(f"Input error: Boundary '{self.name}' is under-constrained in {ax}-axis")
            if f is not None and u is not None:
                raise ValueError(f"Input error: Boundary '{self.name}' is over-constrained in {ax}-axis")


@dataclass
class Problem:
    """Flat SoA problem as it crosses the C ABI (include/magnetite_hip.h mag_problem)."""
    mesh: Mesh
    u_known: np.ndarray  # (2N,) u8: 1 => displacement prescribed, 0 => force prescribed
    u_in: np.ndarray     # (2N,) f64, read where u_known==1
    f_in: np.ndarray     # (2N,) f64, read where u_known==0
    youngs_modulus: float = ALU["youngs_modulus"]
    poisson_ratio: float = ALU["poisson_ratio"]
    part_thickness: float = ALU["part_thickness"]
    meta: dict = field(default_factory=dict)

    @property
    def xy_flat(self):
        return np.ascontiguousarray(self.mesh.xy, dtype=np.float64).reshape(-1)

    @property
    def conn_flat(self):
        return np.ascontiguousarray(self.mesh.conn, dtype=np.int32).reshape(-1)


def apply_boundary_rules(mesh, rules, **material):
    """mesher.rs:815-930 apply_boundary_conditions on flat arrays."""
    N = mesh.num_nodes
    x, y = mesh.xy[:, 0], mesh.xy[:, 1]
    # Option state per node/axis: known-u flag + values; defaults mesher.rs:615-624
    u_known = np.zeros((N, 2), dtype=np.uint8)
    u_in = np.zeros((N, 2))
    f_in = np.zeros((N, 2))
    for r in rules:
        r.validate()
        cand = (x > r.x_min) & (x < r.x_max) & (y > r.y_min) & (y < r.y_max)
        for ax, (u, f) in enumerate(((r.ux, r.fx), (r.uy, r.fy))):
            if u is not None:
                u_known[cand, ax] = 1
                u_in[cand, ax] = u
                f_in[cand, ax] = 0.0
            else:
                u_known[cand, ax] = 0
                u_in[cand, ax] = 0.0
                f_in[cand, ax] = f
    mat = dict(ALU)
    mat.update(material)
    return Problem(mesh, u_known.reshape(-1), u_in.reshape(-1), f_in.reshape(-1), **mat)


def _grid(nx, ny, lx, ly, x0=0.0, y0=0.0):
    xs = x0 + np.arange(nx + 1, dtype=np.float64) * (lx / nx)
    ys = y0 + np.arange(ny + 1, dtype=np.float64) * (ly / ny)
    X, Y = np.meshgrid(xs, ys)  # row-major: node id = j*(nx+1)+i
    xy = np.stack([X.reshape(-1), Y.reshape(-1)], axis=1)
    i = np.arange(nx, dtype=np.int64)[None, :]
    j = np.arange(ny, dtype=np.int64)[:, None]
    n00 = (j * (nx + 1) + i).reshape(-1)
    n10, n01, n11 = n00 + 1, n00 + nx + 1, n00 + nx + 2
    # two CCW triangles per cell: (00,10,11) and (00,11,01)
    tri = np.empty((nx * ny, 2, 3), dtype=np.int32)
    tri[:, 0, 0], tri[:, 0, 1], tri[:, 0, 2] = n00, n10, n11
    tri[:, 1, 0], tri[:, 1, 1], tri[:, 1, 2] = n00, n11, n01
    cx = (x0 + (i + 0.5) * (lx / nx)) + 0 * j
    cy = (y0 + (j + 0.5) * (ly / ny)) + 0 * i
    return xy, tri, cx.reshape(-1), cy.reshape(-1)


def _compact(xy, tri_cells, keep, name):
    conn = tri_cells[keep].reshape(-1, 3)
    used = np.zeros(xy.shape[0], dtype=bool)
    used[conn.reshape(-1)] = True
    remap = np.cumsum(used, dtype=np.int64) - 1
    return Mesh(np.ascontiguousarray(xy[used]), remap[conn].astype(np.int32), name)


def plate(nx, ny=None, lx=1.0, ly=None, name=None):
    """Structured rectangular plate, 2*nx*ny CCW triangles (configs 2 and 4)."""
    ny = nx if ny is None else ny
    ly = lx * ny / nx if ly is None else ly
    xy, tri, _, _ = _grid(nx, ny, lx, ly)
    return Mesh(xy, tri.reshape(-1, 3), name or f"plate_{nx}x{ny}")


def plate_with_holes(nx, ny=None, lx=1.0, ly=None, holes=((0.5, 0.5, 0.15),), name=None):
    """Plate with circular holes cut cell-wise (stair-step boundary; config 3 / 5).

    holes: (cx, cy, r) in units of lx / ly / min(lx,ly).  A cell is removed when
    its centre lies inside a hole; unused nodes are dropped and renumbered.
    """
    ny = nx if ny is None else ny
    ly = lx * ny / nx if ly is None else ly
    xy, tri, cx, cy = _grid(nx, ny, lx, ly)
    keep = np.ones(cx.shape[0], dtype=bool)
    s = min(lx, ly)
    for hx, hy, hr in holes:
        keep &= (cx - hx * lx) ** 2 + (cy - hy * ly) ** 2 > (hr * s) ** 2
    return _compact(xy, tri, keep, name or f"plate_holes_{nx}x{ny}_{len(holes)}")


def multi_hole(nx, k=4, r=0.25, **kw):
    """k x k holes on a regular lattice, radius r in units of the lattice pitch (config 5)."""
    holes = [((a + 0.5) / k, (b + 0.5) / k, r / k) for b in range(k) for a in range(k)]
    return plate_with_holes(nx, holes=holes, name=f"multihole_{nx}_{k}x{k}", **kw)


def grid_for_triangles(target, hole_fraction=0.0):
    """Cells per side so that 2*n*n*(1-hole_fraction) ~= target triangles."""
    return max(1, int(round(np.sqrt(target / 2.0 / (1.0 - hole_fraction)))))


def shuffle(mesh, seed):
    """Random node renumbering + element reordering + per-element corner rotation.

    gmsh numbering carries no spatial order the solver may rely on; parity tests
    use this to make sure nothing depends on the generators' row-major order.
    Rotation keeps orientation (signed area unchanged).
    """
    rng = np.random.default_rng(seed)
    N, E = mesh.num_nodes, mesh.num_elements
    newid = rng.permutation(N)          # old -> new
    xy = np.empty_like(mesh.xy)
    xy[newid] = mesh.xy
    conn = newid[mesh.conn]
    rot = rng.integers(0, 3, size=E)
    idx = (np.arange(3)[None, :] + rot[:, None]) % 3
    conn = np.take_along_axis(conn, idx, axis=1)
    conn = conn[rng.permutation(E)]
    return Mesh(xy, conn.astype(np.int32), mesh.name + f"_shuf{seed}")


def perturb(mesh, amount, seed=12345):
    """move nodes by uniform +-amount*h (keeps CCW for amount <= 0.2)."""
    rng = np.random.default_rng(seed)
    a = mesh.xy[mesh.conn]
    h = np.sqrt(np.abs(np.mean(0.5 * ((a[:, 1, 0] - a[:, 0, 0]) * (a[:, 2, 1] - a[:, 0, 1]) -
                                      (a[:, 2, 0] - a[:, 0, 0]) * (a[:, 1, 1] - a[:, 0, 1]))) * 2.0))
    d = rng.uniform(-amount * h, amount * h, size=mesh.xy.shape)
    lo, hi = mesh.xy.min(axis=0), mesh.xy.max(axis=0)
    edge = (np.abs(mesh.xy - lo) < 1e-12 * (hi - lo)).any(axis=1) | (np.abs(mesh.xy - hi) < 1e-12 * (hi - lo)).any(axis=1)
    d[edge] = 0.0  # outer boundary stays put so edge rules still catch it
    return Mesh(mesh.xy + d, mesh.conn.copy(), mesh.name + "_pert")


def frontal_like(n, jitter=0.4, seed=1, lx=1.0, name=None):
    """Unstructured stand-in for what `gmsh geom.geo -2` hands solver::run (mesher.rs:501-506; gmsh is not installed):
    an equilateral lattice of pitch lx / n on the unit square (straight boundary frame), interior points moved by uniform
    +-jitter pitches, Delaunay-triangulated (scipy), CCW.  At jitter 0.4 the valence histogram is that of a frontal
    mesh: half the nodes have six neighbours, a quarter five, a quarter seven or more (3 % eight, a few nine or ten).
    2 n^2 * 1.155 triangles; deterministic for a given (n, jitter, seed)."""
    from scipy.spatial import Delaunay
    h = lx / n
    rows = int(round(lx / (h * np.sqrt(3.0) / 2.0)))
    hy = lx / rows
    rng = np.random.default_rng(seed)
    pts = []
    for j in range(rows + 1):
        xs = np.arange(n + 1) * h if j % 2 == 0 else np.concatenate([[0.0], (np.arange(n) + 0.5) * h, [lx]])
        p = np.stack([xs, np.full_like(xs, j * hy)], axis=1)
        d = rng.uniform(-jitter * h, jitter * h, size=p.shape)
        inner = (p[:, 0] > 1e-12) & (p[:, 0] < lx - 1e-12) & (j > 0) & (j < rows)
        p[inner] += d[inner]
        pts.append(p)
    xy = np.concatenate(pts)
    tri = Delaunay(xy).simplices.astype(np.int64)
    a = xy[tri]
    area = 0.5 * ((a[:, 1, 0] - a[:, 0, 0]) * (a[:, 2, 1] - a[:, 0, 1]) - (a[:, 2, 0] - a[:, 0, 0]) * (a[:, 1, 1] - a[:, 0, 1]))
    tri = tri[np.abs(area) > 1e-9 * h * h]  # (collinear frame points can leave a sliver of zero area)
    area = area[np.abs(area) > 1e-9 * h * h]
    flip = area < 0
    tri[flip] = tri[flip][:, ::-1]
    return Mesh(np.ascontiguousarray(xy), np.ascontiguousarray(tri.astype(np.int32)), name or f"frontal_{n}_j{jitter}_s{seed}")


def clockwise(mesh):
    """Reverse every element (what mesher.rs:522-526 check_ccw does to all elements
    of a fine mesh: signed area < 1.0 => reversed => K negative semidefinite)."""
    return Mesh(mesh.xy.copy(), np.ascontiguousarray(mesh.conn[:, ::-1]), mesh.name + "_cw")


# ------------------------------------------------------------ BASELINE configs


def config_fixed_left_point_load(mesh, load=1e6, **material):
    """Config 2: fixed-left (ux=uy=0 on x<eps) + point load fx on the node nearest (L, L/2)."""
    x, y = mesh.xy[:, 0], mesh.xy[:, 1]
    lx, ly = x.max() - x.min(), y.max() - y.min()
    eps = 1e-9 * max(lx, ly)
    rules = [BoundaryRule("restraint", x_max=x.min() + eps, ux=0.0, uy=0.0)]
    p = apply_boundary_rules(mesh, rules, **material)
    tgt = np.array([x.max(), y.min() + 0.5 * ly])
    node = int(np.argmin(((mesh.xy - tgt) ** 2).sum(axis=1)))
    p.f_in[2 * node] = load
    p.meta = dict(config="fixed-left + point-load-right", load_node=node, load=load)
    return p


def config_fixed_left_pull_right(mesh, delta=None, **material):
    """Config 3: left edge fixed, right edge ux=delta, fy=0 (the tensile-example pattern,
    examples/tensile-example/input.json:10-33)."""
    x, y = mesh.xy[:, 0], mesh.xy[:, 1]
    lx, ly = x.max() - x.min(), y.max() - y.min()
    eps = 1e-9 * max(lx, ly)
    delta = 1e-3 * lx if delta is None else delta
    rules = [BoundaryRule("restraint", x_max=x.min() + eps, ux=0.0, uy=0.0),
             BoundaryRule("load", x_min=x.max() - eps, ux=delta, fy=0.0)]
    p = apply_boundary_rules(mesh, rules, **material)
    p.meta = dict(config="fixed-left + ux=delta right", delta=delta)
    return p


def baseline_problem(which, scale=1.0):
    """BASELINE.json configs[1..4] as single-GPU problems (`scale` shrinks for CPU tests)."""
    if which == "plate100k":
        return config_fixed_left_point_load(plate(max(2, int(224 * scale))))
    if which == "hole1m":
        n = grid_for_triangles(1e6 * scale * scale, np.pi * 0.15 ** 2)
        return config_fixed_left_pull_right(plate_with_holes(n))
    if which == "frontal1m":  # round 4: the unstructured stand-in for a gmsh mesh (frontal_like), 1 006 602 triangles
        return config_fixed_left_pull_right(frontal_like(max(4, int(660 * scale)), 0.4, 1))
    if which == "plate4m":
        return config_fixed_left_pull_right(plate(max(2, int(1414 * scale))))
    if which == "multihole16m":
        n = grid_for_triangles(16e6 * scale * scale, np.pi * 0.25 ** 2)
        return config_fixed_left_pull_right(multi_hole(n, 4, 0.25))
    raise KeyError(which)


def check_ccw(mesh):
    """mesher.rs:522-526 as written: an element whose SIGNED area is < 1.0 gets its node list reversed.
    (Looks like a typo for < 0.0; on a fine mesh every CCW element has area < 1 and ends up clockwise.)"""
    a = mesh.xy[mesh.conn]
    area = 0.5 * (a[:, 0, 0] * (a[:, 1, 1] - a[:, 2, 1]) + a[:, 1, 0] * (a[:, 2, 1] - a[:, 0, 1]) +
                  a[:, 2, 0] * (a[:, 0, 1] - a[:, 1, 1]))
    conn = mesh.conn.copy()
    flip = area < 1.0
    conn[flip] = conn[flip][:, ::-1]
    return Mesh(mesh.xy.copy(), np.ascontiguousarray(conn), mesh.name + "_checkccw")


def _inside(poly, pts):
    """Even-odd ray casting, vectorised: poly (M,2) closed implicitly, pts (K,2)."""
    x, y = pts[:, 0][:, None], pts[:, 1][:, None]
    x0, y0 = poly[:, 0][None, :], poly[:, 1][None, :]
    x1, y1 = np.roll(poly[:, 0], -1)[None, :], np.roll(poly[:, 1], -1)[None, :]
    cond = (y0 > y) != (y1 > y)
    with np.errstate(divide="ignore", invalid="ignore"):
        xint = x0 + (y - y0) * (x1 - x0) / (y1 - y0)
    return (np.sum(cond & (x < xint), axis=1) % 2) == 1


def polygon_mesh(outline, h, name="polygon"):
    """Unstructured CCW triangulation of a simple polygon (stand-in for `gmsh geom.geo -2`, mesher.rs:501-506):
    boundary points every <= h along the outline + a staggered interior lattice, Delaunay, triangles whose
    centroid is outside the polygon dropped.  Deterministic."""
    from scipy.spatial import Delaunay
    outline = np.asarray(outline, dtype=np.float64)
    bpts = []
    for a, b in zip(outline, np.roll(outline, -1, axis=0)):
        n = max(1, int(np.ceil(np.linalg.norm(b - a) / h)))
        for k in range(n):
            bpts.append(a + (b - a) * (k / n))
    bpts = np.array(bpts)
    lo, hi = outline.min(axis=0), outline.max(axis=0)
    ys = np.arange(lo[1] + 0.5 * h, hi[1], h * np.sqrt(3) / 2)
    ipts = []
    for j, yv in enumerate(ys):
        xs = np.arange(lo[0] + (0.5 if j % 2 else 0.25) * h, hi[0], h)
        ipts.append(np.stack([xs, np.full_like(xs, yv)], axis=1))
    ipts = np.concatenate(ipts)
    ipts = ipts[_inside(outline, ipts)]
    # keep interior points at least 0.5 h away from every boundary point
    d2 = ((ipts[:, None, :] - bpts[None, :, :]) ** 2).sum(axis=2).min(axis=1)
    ipts = ipts[d2 > (0.5 * h) ** 2]
    pts = np.concatenate([bpts, ipts])
    tri = Delaunay(pts).simplices.astype(np.int64)
    cen = pts[tri].mean(axis=1)
    tri = tri[_inside(outline, cen)]
    a = pts[tri]
    area = 0.5 * ((a[:, 1, 0] - a[:, 0, 0]) * (a[:, 2, 1] - a[:, 0, 1]) - (a[:, 2, 0] - a[:, 0, 0]) * (a[:, 1, 1] - a[:, 0, 1]))
    tri = tri[np.abs(area) > 1e-9 * h * h]
    area = area[np.abs(area) > 1e-9 * h * h]
    tri[area < 0] = tri[area < 0][:, ::-1]
    used = np.zeros(len(pts), dtype=bool)
    used[tri.reshape(-1)] = True
    remap = np.cumsum(used) - 1
    return Mesh(np.ascontiguousarray(pts[used]), remap[tri].astype(np.int32), name)
