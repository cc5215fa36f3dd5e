"""-m gpu: oracle parity AT BASELINE SIZE (BASELINE.json configs 2-4; config 5 when its fixture is committed).

The oracle (oracle/magnetite_oracle.c, 1 thread) needs 1-15 minutes per solve at these sizes, so its solutions are
committed SAMPLED under tests/golden/fullsize_<workload>.npz (generator: tests/golden/make_fullsize_fixtures.py --
iteration count, final cost, norms, and u / f / stress at 4096 fixed pseudo-random positions each).  The library
rebuilds the identical mesh (checksums below), solves with bench.py's stop rule (relative residual 1e-8) and must land
on the oracle's numbers with BOTH the on-chip kernel (cg_variant 2, where the mesh fits the chip) and the streaming
kernel (cg_variant 1).  plate100k is small enough to call the oracle live and compare whole vectors.

Bars: nodal displacements <= 1e-8 relative L2 (north_star); reactions and stress are first differences of two
iterative solutions stopped at a 1e-8 residual, i.e. amplified by 1/h: 1e-7 there.  PARITY UNPINNED against reference
outputs (none exist); these pin the HIP path to the oracle at full size.
"""
import os

import numpy as np
import pytest

from magnetite_amd import Context, _lib, meshgen

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_U, TOL_DERIVED = 1e-8, 1e-7


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def fixture(name):
    path = os.path.join(GOLDEN, f"fullsize_{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not committed (generate with tests/golden/make_fullsize_fixtures.py {name})")
    return np.load(path, allow_pickle=False)


# (frontal1m, round 4: the unstructured stand-in for a gmsh mesh -- on chip with overflow edge blocks, and streamed)
CASES = [("hole1m", 2), ("hole1m", 1), ("plate4m", 1), ("multihole16m", 1), ("frontal1m", 2), ("frontal1m", 1)]


@pytest.mark.parametrize("name,variant", CASES)
def test_sampled_oracle_solution_at_baseline_size(built, name, variant):
    fx = fixture(name)
    p = meshgen.baseline_problem(name)
    N, E = p.mesh.num_nodes, p.mesh.num_elements
    assert (N, E) == (int(fx["num_nodes"]), int(fx["num_elements"]))
    assert float(np.sum(p.xy_flat * np.arange(1, 2 * N + 1) % 7.0)) == float(fx["xy_checksum"])
    assert int(np.sum(p.conn_flat.astype(np.int64) * (np.arange(3 * E) % 11 + 1))) == int(fx["conn_checksum"])
    with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=float(fx["rel_tol"]), cg_variant=variant) as c:
        out = c.solve(p)
    assert out["converged"] == 1 and out["cg_kernel"] == variant
    # same recurrences, same stop rule: the iteration counts agree (the serial oracle's to the iteration; the OpenMP
    # oracle used for 16M reduces its dot products in a different order)
    slack = 0 if "1 thread" in str(fx["solver"]) else max(2, int(fx["iterations"]) // 1000)
    if name == "frontal1m":  # the kernels add a node's triangles in another order than the oracle's rows (edge blocks; ring
        slack = max(2, int(fx["iterations"]) // 100)  # order): the count wobbles at the threshold (the residual is not monotone)
    assert abs(int(out["iterations"]) - int(fx["iterations"])) <= slack, (out["iterations"], int(fx["iterations"]))
    iu, ie = fx["dof_idx"], fx["elem_idx"]
    assert rel(out["u"][iu], fx["u_at"]) <= TOL_U
    assert abs(np.linalg.norm(out["u"]) - float(fx["u_norm"])) <= TOL_U * float(fx["u_norm"])
    assert np.abs(out["u"]).max() == pytest.approx(float(fx["u_absmax"]), rel=1e-9)
    known = p.u_known == 1
    assert np.array_equal(out["u"][known], p.u_in[known])
    # reactions: compare on the scale of the reaction vector (most sampled DOFs carry f = f_in exactly)
    assert np.abs(out["f"][iu] - fx["f_at"]).max() <= TOL_DERIVED * float(fx["f_known_norm"])
    assert abs(np.linalg.norm(out["f"][known]) - float(fx["f_known_norm"])) <= TOL_DERIVED * float(fx["f_known_norm"])
    assert rel(out["stress"][ie], fx["stress_at"]) <= TOL_DERIVED
    assert abs(np.linalg.norm(out["stress"]) - float(fx["stress_norm"])) <= TOL_DERIVED * float(fx["stress_norm"])


@pytest.mark.parametrize("variant", [2, 1, 0])
def test_plate100k_against_the_live_oracle(built, variant):
    """BASELINE config 2 (100 352 triangles, fixed left, point load right), default reference stop rule (absolute 1e-4)."""
    import oracle
    p = meshgen.baseline_problem("plate100k")
    ref = oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                     p.part_thickness, path="sparse")
    with Context(device=0, cg_variant=variant) as c:
        out = c.solve(p)
    assert out["converged"] == 1 and out["cg_kernel"] == variant
    assert abs(int(out["iterations"]) - ref["iterations"]) <= max(3, ref["iterations"] // 50)
    assert rel(out["u"], ref["u"]) <= TOL_U
    assert rel(out["stress"], ref["stress"]) <= TOL_DERIVED
    k = p.u_known == 1
    assert np.array_equal(out["f"][~k], p.f_in[~k])
    assert rel(out["f"][k], ref["f"][k]) <= TOL_DERIVED


@pytest.mark.parametrize("name", ["hole1m", "plate4m"])
def test_assembled_matrix_at_baseline_size_is_the_oracles_bit_for_bit(built, name, monkeypatch):
    """K (solver.rs:290-331) of the BASELINE meshes, assembled from the CG tiles (k_assemble_ctile, the default) and by
    round 2's kernel on caller-numbered gathers (MAG_TUNE_ASSEMBLY=tiles): pattern and values equal to the oracle's
    sparse restatement bit for bit (the oracle needs ~1-3 s for the assembly alone, no solve)."""
    import oracle
    p = meshgen.baseline_problem(name)
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness, threads=8)
    for how in (None, "tiles"):
        if how:
            monkeypatch.setenv("MAG_TUNE_ASSEMBLY", how)
        with Context(device=0) as c:
            c.upload_problem(p)
            rowptr, col, val = c.assemble_csr()
        assert np.array_equal(rowptr.astype(np.int64), K.rowptr) and np.array_equal(col, K.col), how
        assert np.array_equal(val, K.val), how
