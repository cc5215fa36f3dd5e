"""Generates the committed golden fixtures under tests/golden/ (run from the repo root: python tests/golden/make_fixtures.py).

PARITY UNPINNED: the reference has no tests/fixtures and cannot be built here (Rust), so these vectors are the
CPU oracle's outputs (oracle/magnetite_oracle.c, which follows src/solver.rs line by line) on small inputs --
they pin the oracle against drift and give the HIP path fixed numbers to hit, they are not reference outputs.

  tensile.npz   examples/tensile-example: outline = tensile_outline.csv (the example's vertices.csv, input data),
                meshed here at h=0.6 (no gmsh in this image), check_ccw as the reference's mesher applies it
                (mesher.rs:522-526), BCs and material of the example's input.json:10-33,3-5.
  plate.npz     6x4 plate, shuffled numbering, fixed-left + point load (BASELINE config 2 pattern), plus K_e, the
                CSR of K and the reduced system.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from magnetite_amd import meshgen  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def solve(p, hist=32):
    return oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                      p.part_thickness, path="dense", hist_len=hist)


def tensile_problem():
    outline = np.loadtxt(os.path.join(HERE, "tensile_outline.csv"), delimiter=",", skiprows=1)
    mesh = meshgen.check_ccw(meshgen.polygon_mesh(outline, 0.6, "tensile"))
    rules = [meshgen.BoundaryRule("restraint", x_min=-12, x_max=-10, ux=0.0, uy=0.0),
             meshgen.BoundaryRule("load", x_min=10, x_max=12, ux=3.0, fy=0.0)]
    return meshgen.apply_boundary_rules(mesh, rules, youngs_modulus=69e9, poisson_ratio=0.33, part_thickness=0.5)


def plate_problem():
    return meshgen.config_fixed_left_point_load(meshgen.shuffle(meshgen.plate(6, 4, 3.0, 2.0), 5))


def save(name, p, extra=None):
    r = solve(p)
    d = dict(xy=p.mesh.xy, conn=p.mesh.conn, u_known=p.u_known, u_in=p.u_in, f_in=p.f_in,
             material=np.array([p.youngs_modulus, p.poisson_ratio, p.part_thickness]),
             u=r["u"], f=r["f"], stress=r["stress"], iterations=np.int64(r["iterations"]),
             final_cost=np.float64(r["final_cost"]), history=r["history"], n_free=np.int64(r["n_free"]),
             nnz_ff=np.int64(r["nnz_ff"]))
    d.update(extra or {})
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, "N", p.mesh.num_nodes, "E", p.mesh.num_elements, "iters", r["iterations"], "cost", r["final_cost"])


if __name__ == "__main__":
    save("tensile.npz", tensile_problem())
    p = plate_problem()
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    A, b = oracle.reduce_system(K, p.u_known, p.u_in, p.f_in)
    save("plate.npz", p, dict(ke=oracle.element_stiffness_all(p.xy_flat, p.conn_flat, p.poisson_ratio,
                                                              p.youngs_modulus, p.part_thickness),
                              K_rowptr=K.rowptr, K_col=K.col, K_val=K.val, A_rowptr=A.rowptr, A_col=A.col,
                              A_val=A.val, b=b))
