"""Generates tests/golden/fullsize_<workload>.npz: the CPU oracle's solution of a BASELINE-size workload, SAMPLED.

  python tests/golden/make_fullsize_fixtures.py hole1m plate4m [multihole16m]

PARITY UNPINNED: the reference cannot run any of these sizes (dense n x n K, solver.rs:295-296) and cannot be built
here (Rust); these are outputs of oracle/magnetite_oracle.c (sparse path: the reference's K_e arithmetic, `+=` order,
CSR SpMV in ascending column order, argmin's CG recurrences, 1 thread) on exactly the meshes bench.py builds, at
bench.py's stop rule (relative residual 1e-8, BASELINE config 3: "CG to 1e-8").  The oracle needs ~1 min (1M
triangles) to ~15 min (4M) per solve, so the GPU suite cannot call it live at these sizes; the fixture keeps what a
parity test needs: the iteration count, the final cost, the norms of u / reaction forces / stress, and the values of
u, f and stress at fixed pseudo-random positions (seed below) -- a few thousand each, a few hundred KB in all.

multihole16m uses the oracle's OpenMP CG (orc_cg_parallel: same recurrences, dot products reduced per thread) because
the serial one needs hours; the fixture records which one ran.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from magnetite_amd import meshgen  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED, NSAMPLE, TOL = 20261004, 4096, 1e-8
# frontal1m (round 4, the unstructured stand-in for a gmsh mesh): at a relative 1e-8 its solution is only determined to ~2e-8
# (two correct solves that stop a few iterations apart differ by that much: the triangle walk and the edge blocks of the same
# library do), so a 1e-8 parity bar needs a tighter stop -- its fixture is taken at 1e-10
TOLS = {"frontal1m": 1e-10}


def sample_indices(n, k=NSAMPLE, seed=SEED):
    return np.sort(np.random.default_rng(seed).choice(n, size=min(k, n), replace=False)).astype(np.int64)


def main(names):
    for name in names:
        TOL = TOLS.get(name, globals()["TOL"])
        p = meshgen.baseline_problem(name)
        N, E = p.mesh.num_nodes, p.mesh.num_elements
        t0 = time.time()
        if name == "multihole16m":
            K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
            A, b = oracle.reduce_system(K, p.u_known, p.u_in, p.f_in)
            x, it, cost = oracle.cg_parallel(A, b, stop_mode=oracle.STOP_REL, tol=TOL, threads=os.cpu_count())
            u = p.u_in.copy()
            u[p.u_known == 0] = x
            f = p.f_in.copy()
            k = p.u_known == 1
            f[k] = K.spmv(u)[k]
            s = oracle.stress(p.xy_flat, p.conn_flat, u, p.poisson_ratio, p.youngs_modulus)
            ref = dict(u=u, f=f, stress=s, iterations=it, final_cost=cost)
            solver = f"orc_cg_parallel ({os.cpu_count()} threads)"
        else:
            ref = oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                             p.part_thickness, path="sparse", stop_mode=oracle.STOP_REL, tol=TOL)
            solver = "orc_run_sparse (1 thread)"
        dt = time.time() - t0
        iu, ie = sample_indices(2 * N), sample_indices(E)
        known = p.u_known == 1
        out = os.path.join(HERE, f"fullsize_{name}.npz")
        np.savez_compressed(
            out, workload=name, num_nodes=N, num_elements=E, rel_tol=TOL, solver=solver,
            iterations=ref["iterations"], final_cost=ref["final_cost"], oracle_seconds=dt,
            u_norm=np.linalg.norm(ref["u"]), f_known_norm=np.linalg.norm(ref["f"][known]),
            stress_norm=np.linalg.norm(ref["stress"]), u_absmax=np.abs(ref["u"]).max(),
            dof_idx=iu, u_at=ref["u"][iu], f_at=ref["f"][iu], elem_idx=ie, stress_at=ref["stress"][ie],
            # mesh identity: a test that rebuilds the workload must get exactly this mesh
            xy_checksum=float(np.sum(p.xy_flat * np.arange(1, 2 * N + 1) % 7.0)),
            conn_checksum=int(np.sum(p.conn_flat.astype(np.int64) * (np.arange(3 * E) % 11 + 1))))
        print(f"{name}: E={E} N={N} iterations={ref['iterations']} cost={ref['final_cost']:.3e} "
              f"{dt:.0f} s [{solver}] -> {out} ({os.path.getsize(out) / 1024:.0f} KB)", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or ["hole1m", "plate4m"])
