"""Where an iteration of the MULTI-GPU on-chip CG kernel spends its time, measured inside the kernel -- as far as one GPU allows.

Two ranks share the one GPU of the development box at the real per-workgroup load (MAG_TUNE_PERSIST_K=4: four 512-node tiles
per workgroup, 123 compute workgroups + the exchange workgroup per rank, both grids co-resident), trade through same-device
inboxes, and run the diagnostic build (`make -C magnetite_amd/csrc stamps`): lane 0 of every compute workgroup stamps the
phases of iterations 200..1199 as on one GPU (scripts/persist_phases.py); the exchange workgroup (persist_comm_loop) stamps
its own three waits.  What this cannot show is the xGMI hop: every store lands in local memory here.

    python scripts/persist_phases_mg.py [out.json]         (on the GPU box)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASES = ["scalars_and_vector_updates", "workgroup_barrier", "ring_walks_and_q_publication", "workgroup_sums_and_record_publication",
          "wait_before_first_sweep", "sweeps_until_every_tag_matches", "sums_in_rank_order_and_barrier"]
DETAIL = ["sums_wave_trees", "sums_barrier_wait_for_slowest_wave", "sums_chain_and_record_store", "deferred_x_update"]
COMM = ["waiting_for_this_ranks_records", "summing_and_storing_the_rank_sum_into_every_inbox", "waiting_for_every_ranks_sum"]


def bench(env_extra, k):
    env = dict(os.environ, MAG_TUNE_PERSIST_K=str(k), **env_extra)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--partition", "strong", "--workload", "hole1m",
           "--exchange", "inboxes", "--no-cpu-baseline", "--no-hbm-resident", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not lines:
        raise SystemExit(r.stdout[-1500:] + r.stderr[-1500:])
    return json.loads(lines[-1])


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_persist_phases_mg.json")
    k = 4
    plain = bench({}, k)
    stamps = "/tmp/persist_stamps_mg.csv"
    st = bench({"MAG_LIB_PATH": os.path.join(ROOT, "magnetite_amd", "libmagnetite_hip_stamps.so"), "MAG_TUNE_PERSIST_STAMPS": stamps}, k)
    out = {"what": __doc__.strip().splitlines()[0], "clock": "s_memrealtime, 100 MHz", "tiles_per_workgroup": k,
           "us_per_iteration_product_build": plain["roofline"]["us_per_iteration"], "cg_kernel": plain["config"]["cg_kernel"],
           "exchange": plain["config"]["exchange"], "iterations": plain["cg_iterations"],
           "us_per_iteration_stamped_build": st["roofline"]["us_per_iteration"], "ranks": []}
    for rank in (0, 1):
        rows = [[int(v) for v in l.split(",")] for l in open(f"{stamps}.{rank}") if l.strip()]
        comm, rows = rows[-1], [r for r in rows[:-1] if r[-1] > 0]
        n = len(rows)
        us = lambda col: sum(r[col] / r[-1] * 0.01 for r in rows) / n
        d = {"rank": rank, "compute_workgroups": n, "phases_us_mean": {p: us(i) for i, p in enumerate(PHASES)},
             "sweeps_per_iteration": sum(r[7] / r[-1] for r in rows) / n,
             "detail_us_mean": {p: us(8 + i) for i, p in enumerate(DETAIL)},
             "exchange_workgroup_us_mean": ({p: comm[i] / comm[-1] * 0.01 for i, p in enumerate(COMM)} if comm[-1] > 0 else None)}
        d["phases_sum_us"] = sum(d["phases_us_mean"].values())
        print(json.dumps(d), flush=True)
        out["ranks"].append(d)
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
