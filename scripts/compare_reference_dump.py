"""Compare the output of tests/golden/dump_reference.rs -- the REFERENCE's own run of the tensile fixture, produced by a
maintainer with cargo -- with this repository's oracle (tests/golden/tensile.npz + oracle/ called live).

    python scripts/compare_reference_dump.py reference_dump.txt

Prints one JSON verdict and exits 0 when the oracle is pinned: K_e of element 0 and b bit for bit, the same iteration
count, u / f / stress within 1e-8 (two CG runs stopped by the same absolute rule), and says which of the two readings of
argmin's `cost` the reference uses (|r|: MAG_STOP_RNORM, the default here; |r|^2: MAG_STOP_RNORM_SQ) and whether argmin
reports a cost before the first iteration (deviations).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_dump(path):
    text = open(path).read()
    a, b = text.index("REFERENCE_DUMP_BEGIN") + len("REFERENCE_DUMP_BEGIN"), text.index("REFERENCE_DUMP_END")
    d = json.loads(text[a:b].replace("inf", "Infinity").replace("NaN", "NaN"))
    return {k: (np.array(v, dtype=np.float64) if isinstance(v, list) else v) for k, v in d.items()}


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def compare(d):
    import oracle
    g = np.load(os.path.join(ROOT, "tests", "golden", "tensile.npz"))
    E, nu, t = (float(v) for v in g["material"])
    xy, conn = g["xy"].reshape(-1), g["conn"].reshape(-1).astype(np.int32)
    out = {"mesh_matches": bool(d["num_nodes"] == g["xy"].shape[0] and d["num_elements"] == g["conn"].shape[0])}
    ke = oracle.element_stiffness_all(xy, conn, nu, E, t)
    out["ke0_bit_exact"] = bool(np.array_equal(ke[0].reshape(-1), d["ke0"]))
    K = oracle.assemble_sparse(xy, conn, nu, E, t)
    A, b = oracle.reduce_system(K, g["u_known"], g["u_in"], g["f_in"])
    out["n_free_matches"] = bool(A.n == d["n_free"])
    out["nnz_ff_matches"] = bool(A.nnz == d["nnz_ff"])
    out["b_bit_exact"] = bool(np.array_equal(b, d["b"]))
    # which reading of `cost`?
    c1, rn, rn2 = float(d["cost_after_1_iteration"]), float(d["residual_norm_after_1_iteration"]), float(
        d["residual_norm_squared_after_1_iteration"])
    near = lambda x, y: abs(x - y) <= 1e-9 * max(abs(y), 1e-300)
    out["argmin_cost_is"] = "rnorm" if near(c1, rn) else ("rnorm_sq" if near(c1, rn2) else "neither")
    out["argmin_reports_a_cost_before_the_first_iteration"] = bool(np.isfinite(d["cost_before_first_iteration"]))
    mode = {"rnorm": oracle.STOP_RNORM, "rnorm_sq": oracle.STOP_RNORM_SQ}.get(out["argmin_cost_is"], oracle.STOP_RNORM)
    ref = oracle.run(xy, conn, g["u_known"], g["u_in"], g["f_in"], E, nu, t, path="dense", stop_mode=mode)
    out["oracle_stop_mode_used"] = "MAG_STOP_RNORM" if mode == oracle.STOP_RNORM else "MAG_STOP_RNORM_SQ"
    out["iterations_reference"], out["iterations_oracle"] = int(d["iterations"]), int(ref["iterations"])
    out["iterations_match"] = bool(out["iterations_reference"] == out["iterations_oracle"])
    out["best_cost_reference"], out["final_cost_oracle"] = float(d["best_cost"]), float(ref["final_cost"])
    out["rel_l2_u"], out["rel_l2_f"] = rel(ref["u"], d["u"]), rel(ref["f"], d["f"])
    out["rel_l2_stress"] = rel(ref["stress"], d["stress"])
    out["pinned"] = bool(out["mesh_matches"] and out["ke0_bit_exact"] and out["b_bit_exact"] and out["n_free_matches"]
                         and out["nnz_ff_matches"] and out["argmin_cost_is"] != "neither" and out["iterations_match"]
                         and out["rel_l2_u"] <= 1e-8 and out["rel_l2_f"] <= 1e-7 and out["rel_l2_stress"] <= 1e-7)
    return out


if __name__ == "__main__":
    verdict = compare(load_dump(sys.argv[1]))
    print(json.dumps(verdict, indent=1))
    sys.exit(0 if verdict["pinned"] else 1)
