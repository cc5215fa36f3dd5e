"""Where an iteration of the on-chip CG kernel (k_cg_persist) spends its time, measured inside the kernel.

Runs the 1M-triangle benchmark solve (BASELINE config 3, 5389 iterations) with the DIAGNOSTIC build of the library
(`make -C magnetite_amd/csrc stamps` -> libmagnetite_hip_stamps.so: lane 0 of every workgroup reads the 100 MHz constant
clock, s_memrealtime, at the phase boundaries of iterations 200..1199 and adds the differences up) and the
same solves with the product build (no stamp executes) for the un-instrumented time per iteration.  `--shapes 512,768` also
runs the 768 x 3 workgroup shape -- only in libraries built with -DMAG_PERSIST_768 (round 3's record,
profiles/r03_persist_phases.json, has both: 768 was slower and is not instantiated in the product).

    python scripts/persist_phases.py [--triangles] [out.json]   (on the GPU box; default profiles/r03_persist_phases.json)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASES = ["scalars_and_vector_updates", "workgroup_barrier", "ring_walks_and_q_publication", "workgroup_sums_and_record_publication",
          "wait_before_first_sweep", "sweeps_until_every_tag_matches", "record_reduction_and_barriers"]

WORKER = r"""
import json, os, sys
sys.path.insert(0, %r)
import bench
from magnetite_amd import Context, _lib
prob, desc = bench.build_problem(sys.argv[1], 1)
with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8, cg_variant=2) as c:
    c.upload_problem(prob)
    c.run(); c.run()
    st = c.stats()
print(json.dumps({k: st[k] for k in ("iterations", "ms_cg", "cg_kernel", "num_tiles", "persist_timeout", "edge_blocks", "tiles_per_workgroup")}))
""" % ROOT


TRIANGLES = "--triangles" in sys.argv  # the triangle-walk instantiation (what a mesh that does not qualify for edge blocks runs)


def run(workload, threads, stamps_file):
    env = dict(os.environ, MAG_TUNE_PERSIST_THREADS=str(threads))
    if TRIANGLES:
        env["MAG_TUNE_PERSIST_TRIANGLES"] = "1"
    if stamps_file:
        env["MAG_LIB_PATH"] = os.environ.get("MAG_STAMPS_LIB") or os.path.join(ROOT, "magnetite_amd", "libmagnetite_hip_stamps.so")
        env["MAG_TUNE_PERSIST_STAMPS"] = stamps_file
    r = subprocess.run([sys.executable, "-c", WORKER, workload], env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise SystemExit(r.stdout[-2000:] + r.stderr[-2000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    shapes = [int(x) for a in sys.argv[1:] if a.startswith("--shapes") for x in a.split("=")[1].split(",")] or [512]
    out_path = args[0] if args else os.path.join(ROOT, "profiles", "r03_persist_phases.json")
    out = {"what": __doc__.strip().splitlines()[0], "clock": "s_memrealtime, 100 MHz (10 ns ticks)",
           "iterations_stamped": "200..1199 of each solve, lane 0 of every workgroup",
           "instantiation": "triangle walk (MAG_TUNE_PERSIST_TRIANGLES=1)" if TRIANGLES else "default (edge blocks where the mesh qualifies)",
           "runs": []}
    workloads = [x for a in sys.argv[1:] if a.startswith("--workloads=") for x in a.split("=")[1].split(",")] or ["hole1m", "plate100k"]
    for workload in workloads:
        for threads in shapes:
            plain = run(workload, threads, None)
            f = f"/tmp/persist_stamps_{workload}_{threads}.csv"
            st = run(workload, threads, f)
            rows = [[int(v) for v in l.split(",")] for l in open(f) if l.strip()]
            rows = [r for r in rows if r[-1] > 0]
            detail = None
            if rows and len(rows[0]) >= 15:  # round 4's stamps: six detail intervals behind the eight phase words
                names = ["sums_wave_trees", "sums_barrier_wait_for_slowest_wave", "sums_chain_and_record_store", "deferred_x_update",
                         "after_own_sweeps_wait_for_other_waves", "record_reduction_and_its_barrier"]
                detail = {nm: sum(r[8 + k] / r[-1] * 0.01 for r in rows) / len(rows) for k, nm in enumerate(names)}
            n = len(rows)
            per_wg = [[r[k] / r[-1] * 0.01 for k in range(7)] for r in rows]  # us per iteration
            mean = [sum(w[k] for w in per_wg) / n for k in range(7)]
            lo = [min(w[k] for w in per_wg) for k in range(7)]
            hi = [max(w[k] for w in per_wg) for k in range(7)]
            sweeps = sum(r[7] / r[-1] for r in rows) / n
            d = {"workload": workload, "threads": threads, "node_slots_per_lane": st["tiles_per_workgroup"] if st["tiles_per_workgroup"] < 4 else (3 if threads == 768 else 4), "workgroups": n,
                 "iterations": plain["iterations"], "cg_kernel": plain["cg_kernel"], "edge_blocks": st["edge_blocks"],
                 "us_per_iteration_product_build": plain["ms_cg"] * 1e3 / plain["iterations"],
                 "us_per_iteration_stamped_build": st["ms_cg"] * 1e3 / st["iterations"],
                 "phases_us_mean_over_workgroups": dict(zip(PHASES, mean)),
                 "phases_us_min": dict(zip(PHASES, lo)), "phases_us_max": dict(zip(PHASES, hi)),
                 "phases_sum_us": sum(mean), "sweeps_per_iteration": sweeps,
                 "compute_us": sum(mean[:4]), "exchange_us": sum(mean[4:]), "detail_us_mean": detail}
            print(json.dumps(d), flush=True)
            out["runs"].append(d)
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
