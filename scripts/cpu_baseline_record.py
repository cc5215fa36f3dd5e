"""CPU-baseline record -> profiles/r02_cpu_baseline.json.

Times oracle/magnetite_oracle.c -- the CPU restatement of src/solver.rs -- on the host this script runs on:
  1. the reference-faithful DENSE path (the O(n^2) steps of solver.rs:290-331,365-404,123-137,457-469 as written) on
     BASELINE config 1 (examples/tensile-example geometry, committed mesh) and on a 20k-triangle plate -- where the
     as-written reference stops scaling (dense K of 82 GB at config 2);
  2. the SPARSE restatement (same K_e arithmetic and `+=` order, CSR SpMV, argmin's CG, 1 thread) on config 1, the
     20k plate, config 2 (full, unscaled solve, the reference's own stop rule) and config 3 (full solve to relative 1e-8);
  3. the OpenMP CG on all cores for configs 2 and 3.
Nothing here is scaled or sampled: every number is a complete solve.  bench.py reads `config1_dense_s` from the file.

    python scripts/cpu_baseline_record.py [out.json]   # ~5 minutes, ~7 GB
"""
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import oracle  # noqa: E402
from magnetite_amd import meshgen  # noqa: E402
from make_fixtures import tensile_problem  # noqa: E402


def run(p, path, **kw):
    t0 = time.perf_counter()
    out = oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                     p.part_thickness, path=path, **kw)
    dt = time.perf_counter() - t0
    return {"seconds": dt, "iterations": out["iterations"], "elements": p.mesh.num_elements,
            "nodes": p.mesh.num_nodes, "elements_per_s": p.mesh.num_elements / dt}, out


def phases(p, threads, **kw):
    """sparse path with the phase split of the GPU timers: assembly + BC elimination, then the CG alone"""
    t0 = time.perf_counter()
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    A, b = oracle.reduce_system(K, p.u_known, p.u_in, p.f_in)
    t1 = time.perf_counter()
    if threads == 1:
        _, it, _, _ = oracle.cg(A, b, **kw)
    else:
        _, it, _ = oracle.cg_parallel(A, b, threads=threads, **kw)
    t2 = time.perf_counter()
    E = p.mesh.num_elements
    return {"threads": threads, "assembly_and_bc_s": t1 - t0, "cg_s": t2 - t1, "iterations": it,
            "assembly_elements_per_s": E / (t1 - t0), "cg_iterations_per_s": it / (t2 - t1),
            "elements_per_s": E / (t2 - t0)}


def main():
    # the GPU box gives one GPU's share of its host: 16 cores (bench.py's all-cores leg uses the same cap)
    cores = min(16, os.cpu_count() or 1)
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    rec = {"host": {"cpu": cpu, "cores_used": cores, "cores_visible": os.cpu_count(), "machine": platform.machine()},
           "what": "oracle/magnetite_oracle.c (CPU restatement of src/solver.rs), complete solves, nothing scaled"}
    tens = tensile_problem()
    d, od = run(tens, "dense")
    s, os_ = run(tens, "sparse")
    assert np.array_equal(od["u"], os_["u"])
    rec["config1_tensile_example"] = {"dense": d, "sparse": s}
    rec["config1_dense_s"] = d["seconds"]
    plate20k = meshgen.config_fixed_left_point_load(meshgen.plate(100))
    d, _ = run(plate20k, "dense")
    s, _ = run(plate20k, "sparse")
    d["dense_matrix_bytes"] = 8 * (2 * plate20k.mesh.num_nodes) ** 2
    rec["plate20k"] = {"dense": d, "sparse": s,
                       "note": "dense K of the next BASELINE size (config 2, 101 250 DOF) would need 82 GB, twice"}
    c2 = meshgen.baseline_problem("plate100k")
    rec["config2_plate100k"] = {"stop": "reference default: sqrt(r.r) <= 1e-4",
                                "sparse_1_thread": phases(c2, 1), f"sparse_{cores}_threads": phases(c2, cores)}
    c3 = meshgen.baseline_problem("hole1m")
    kw = dict(stop_mode=oracle.STOP_REL, tol=1e-8)
    rec["config3_hole1m"] = {"stop": "relative residual 1e-8 (bench.py)",
                             "sparse_1_thread": phases(c3, 1, **kw), f"sparse_{cores}_threads": phases(c3, cores, **kw)}
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02_cpu_baseline.json")
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
