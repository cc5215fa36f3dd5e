"""CPU: the maintainer-runnable pin (tests/golden/dump_reference.rs -> scripts/compare_reference_dump.py).  The Rust side
cannot be compiled here (no cargo): what CAN be checked is that its inputs are the fixture's, that the Rust source names
the reference's functions with the signatures solver.rs has, and that the comparison script accepts a dump with the
oracle's own numbers in the dump's format and rejects a perturbed one."""
import json
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import compare_reference_dump as crd  # noqa: E402


def test_dump_inputs_are_the_fixture():
    g = np.load(os.path.join(GOLD, "tensile.npz"))
    rows = [l.rstrip("\n").split(",") for l in open(os.path.join(GOLD, "tensile_nodes.csv"))][1:]
    assert len(rows) == g["xy"].shape[0]
    for i, r in enumerate(rows):
        assert float(r[0]) == g["xy"][i, 0] and float(r[1]) == g["xy"][i, 1]
        for d in (0, 1):
            u, f = r[2 + d], r[4 + d]
            assert (u == "") != (f == "")  # exactly one of displacement / force per DOF (mesher.rs:881-900)
            if g["u_known"][2 * i + d]:
                assert float(u) == g["u_in"][2 * i + d]
            else:
                assert float(f) == g["f_in"][2 * i + d]
    els = np.loadtxt(os.path.join(GOLD, "tensile_elements.csv"), delimiter=",", skiprows=1, dtype=np.int64)
    assert np.array_equal(els, g["conn"])


def test_rust_source_uses_the_references_private_api():
    src = open(os.path.join(GOLD, "dump_reference.rs")).read()
    for name in ("compute_element_stiffness_matrix", "build_total_stiffness_matrix", "build_col_vecs",
                 "build_known_unknown_matrices", "ConjugateGradientOperator", "TARGET_CG_COST", "MAX_CG_ITER", "run("):
        assert name in src, name
    assert "#[cfg(test)] #[path = \"dump_reference.rs\"] mod dump_reference;" in src  # the one line added to solver.rs
    assert re.search(r"use super::\*;", src)


def synthetic_dump(tmp_path, perturb=0.0, squared=False):
    """A dump in the Rust test's format, filled with the ORACLE's numbers: what the reference would print if the oracle
    restates it exactly."""
    import oracle
    g = np.load(os.path.join(GOLD, "tensile.npz"))
    E, nu, t = (float(v) for v in g["material"])
    xy, conn = g["xy"].reshape(-1), g["conn"].reshape(-1).astype(np.int32)
    ke = oracle.element_stiffness_all(xy, conn, nu, E, t)
    K = oracle.assemble_sparse(xy, conn, nu, E, t)
    A, b = oracle.reduce_system(K, g["u_known"], g["u_in"], g["f_in"])
    mode = oracle.STOP_RNORM_SQ if squared else oracle.STOP_RNORM
    ref = oracle.run(xy, conn, g["u_known"], g["u_in"], g["f_in"], E, nu, t, path="dense", stop_mode=mode)
    x1, _, _, _ = oracle.cg(A, b, max_iter=1)
    r1 = b - A.spmv(x1)
    rn = float(np.linalg.norm(r1))
    fmt = lambda v: "[" + ",".join("%r" % float(x) for x in v) + "]"
    u = ref["u"] * (1.0 + perturb)
    text = "running 1 test\nREFERENCE_DUMP_BEGIN\n{\n" + ",\n".join([
        '"num_nodes": %d, "num_elements": %d, "n_free": %d, "nnz_ff": %d' % (g["xy"].shape[0], g["conn"].shape[0], A.n, A.nnz),
        '"ke0": ' + fmt(ke[0].reshape(-1)), '"b": ' + fmt(b), '"cost_before_first_iteration": inf',
        '"cost_after_1_iteration": %r' % (rn * rn if squared else rn), '"residual_norm_after_1_iteration": %r' % rn,
        '"residual_norm_squared_after_1_iteration": %r' % (rn * rn),
        '"iterations": %d, "final_cost": %r, "best_cost": %r' % (ref["iterations"], ref["final_cost"], ref["final_cost"]),
        '"x_best": []', '"u": ' + fmt(u), '"f": ' + fmt(ref["f"]), '"stress": ' + fmt(ref["stress"])]) + \
        "\n}\nREFERENCE_DUMP_END\ntest result: ok.\n"
    path = tmp_path / "reference_dump.txt"
    path.write_text(text)
    return str(path)


def test_comparison_script_pins_and_rejects(built, tmp_path):
    v = crd.compare(crd.load_dump(synthetic_dump(tmp_path)))
    assert v["pinned"] and v["argmin_cost_is"] == "rnorm" and v["iterations_match"] and v["ke0_bit_exact"] and v["b_bit_exact"]
    assert v["argmin_reports_a_cost_before_the_first_iteration"] is False
    v = crd.compare(crd.load_dump(synthetic_dump(tmp_path, squared=True)))
    assert v["pinned"] and v["argmin_cost_is"] == "rnorm_sq" and v["oracle_stop_mode_used"] == "MAG_STOP_RNORM_SQ"
    bad = synthetic_dump(tmp_path, perturb=1e-6)
    assert crd.compare(crd.load_dump(bad))["pinned"] is False
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "compare_reference_dump.py"), bad],
                       capture_output=True, text=True)
    assert r.returncode == 1 and json.loads(r.stdout)["rel_l2_u"] > 1e-8
