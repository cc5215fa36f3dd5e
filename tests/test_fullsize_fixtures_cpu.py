"""CPU: the committed BASELINE-size fixtures (tests/golden/fullsize_*.npz) belong to the meshes the generators build
today -- same sizes, same checksums -- and carry what tests/test_fullsize_parity_gpu.py reads.  (The 16M-triangle mesh
takes ~10 s to build here; its identity is checked on the GPU side only.)"""
import os

import numpy as np
import pytest

from magnetite_amd import meshgen

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["hole1m", "plate4m", "multihole16m", "frontal1m"])
def test_fixture_is_complete_and_matches_its_mesh(name):
    fx = np.load(os.path.join(GOLDEN, f"fullsize_{name}.npz"), allow_pickle=False)
    for key in ("workload", "num_nodes", "num_elements", "rel_tol", "solver", "iterations", "final_cost", "u_norm",
                "f_known_norm", "stress_norm", "u_absmax", "dof_idx", "u_at", "f_at", "elem_idx", "stress_at",
                "xy_checksum", "conn_checksum"):
        assert key in fx.files, key
    N, E = int(fx["num_nodes"]), int(fx["num_elements"])
    # (frontal1m, the unstructured mesh, is taken at 1e-10: at 1e-8 its solution is only determined to ~2e-8)
    assert str(fx["workload"]) == name and float(fx["rel_tol"]) == (1e-10 if name == "frontal1m" else 1e-8) and int(fx["iterations"]) > 1000
    assert fx["dof_idx"].shape == fx["u_at"].shape == fx["f_at"].shape == (4096,) and fx["dof_idx"].max() < 2 * N
    assert fx["elem_idx"].shape == fx["stress_at"].shape == (4096,) and fx["elem_idx"].max() < E
    assert np.all(np.diff(fx["dof_idx"]) > 0) and np.all(np.isfinite(fx["u_at"])) and np.all(np.isfinite(fx["stress_at"]))
    assert float(fx["final_cost"]) > 0 and float(fx["u_norm"]) > 0
    if name == "multihole16m":
        return
    p = meshgen.baseline_problem(name)
    assert (p.mesh.num_nodes, p.mesh.num_elements) == (N, E)
    assert float(np.sum(p.xy_flat * np.arange(1, 2 * N + 1) % 7.0)) == float(fx["xy_checksum"])
    assert int(np.sum(p.conn_flat.astype(np.int64) * (np.arange(3 * E) % 11 + 1))) == int(fx["conn_checksum"])
    # prescribed displacements come back untouched in the oracle's sample too
    k = p.u_known[fx["dof_idx"]] == 1
    assert np.array_equal(fx["u_at"][k], p.u_in[fx["dof_idx"]][k])
