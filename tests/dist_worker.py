"""Multi-rank worker for tests/test_distributed_gpu.py (launched with torch.distributed.run, gloo rendezvous).

mode callback : every rank drives the SAME GPU 0 through its own mag_ctx; the per-iteration all-reduces go through
                the library's host-callback transport -> gloo.  Exercises the whole distributed CG path (tile-range
                partition, interface list, exchange buffer, ghost recurrences, solution assembly) except
                the RCCL call itself, which needs one GPU per rank.
mode rccl1    : single rank, RCCL communicator of size 1, distributed protocol forced (MAG_TUNE_FORCE_DIST=1):
                exercises dlopen(librccl), ncclCommInitRank, ncclAllReduce on the library's stream.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="callback")
    ap.add_argument("--mesh", type=int, default=96)
    ap.add_argument("--variant", type=int, default=1)
    ap.add_argument("--tile", type=int, default=256)
    ap.add_argument("--precond", type=int, default=0, help="mag_options.preconditioner (checked against the oracle's PCG)")
    ap.add_argument("--window", type=int, default=0,
                    help="on-chip CG on every rank: 1 = exchange through a shared host-memory window, 2 = through "
                         "per-rank inboxes in device memory (HIP IPC)")
    ap.add_argument("--expect-kernel", type=int, default=2, help="with --window: mag_stats.cg_kernel every rank must report")
    ap.add_argument("--stream-inbox", action="store_true",
                    help="with --window 2: keep the streaming kernels (cg_variant 1); their per-iteration exchange then goes "
                         "through the inboxes (k_stream_exchange, mag_stats.exchange 3) instead of an all-reduce")
    ap.add_argument("--stacked", type=int, default=0, help="bench.py's weak-scaling geometry: plate-with-hole stacked N times")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist

    import oracle
    from magnetite_amd import Context, meshgen

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if a.stacked:
        holes = [(0.5, (k + 0.5) / a.stacked, 0.15) for k in range(a.stacked)]
        prob = meshgen.config_fixed_left_pull_right(
            meshgen.plate_with_holes(a.mesh, a.mesh * a.stacked, 1.0, float(a.stacked), holes=holes))
    else:
        prob = meshgen.config_fixed_left_pull_right(
            meshgen.shuffle(meshgen.plate_with_holes(a.mesh, 2 * a.mesh, 1.0, 2.0), 3))
    if a.mode == "rccl1":
        os.environ["MAG_TUNE_FORCE_DIST"] = "1"
        dist.init_process_group("gloo", rank=0, world_size=1, init_method="tcp://127.0.0.1:29533")
        with Context(device=0, tile_nodes=a.tile, cg_variant=a.variant, preconditioner=a.precond) as c:
            c.init_rccl_from_torch(dist, 0, 1)
            out = c.solve(prob)
    else:
        dist.init_process_group("gloo")

        def allreduce(arr):
            t = torch.from_numpy(arr)
            dist.all_reduce(t)

        shm = None
        if a.window == 1:
            from multiprocessing import shared_memory
            names = [None]
            if rank == 0:
                shm = shared_memory.SharedMemory(create=True, size=8 << 20)
                names[0] = shm.name
            dist.broadcast_object_list(names, src=0)
            if rank != 0:
                shm = shared_memory.SharedMemory(name=names[0])
                from multiprocessing import resource_tracker
                resource_tracker.unregister(shm._name, "shared_memory")  # rank 0 owns the segment (Python < 3.13)
        with Context(device=0, tile_nodes=a.tile, cg_variant=2 if (a.window and not a.stream_inbox) else a.variant,
                     preconditioner=a.precond) as c:
            c.init_callback(allreduce, rank, world)
            if shm is not None:
                c.set_window(shm)
            if a.window == 2:
                handles = [None] * world
                dist.all_gather_object(handles, c.create_inbox())
                c.open_inboxes(handles)
            out = c.solve(prob)
            kernel = c.stats()["cg_kernel"]
            if a.stream_inbox:
                assert c.stats()["exchange"] == 3, c.stats()["exchange"]
            if a.window:
                print(f"rank {rank}: on-chip across ranks: {out['iterations']} iterations, "
                      f"{c.stats()['ms_cg'] * 1e3 / max(1, out['iterations']):.2f} us per iteration", flush=True)
                out2 = c.solve(prob)                       # a second solve: new tags, same window
                assert np.array_equal(out2["u"], out["u"]) and c.stats()["cg_kernel"] == kernel
                if a.window == 1:
                    c.set_window(None)
                else:
                    dist.barrier()          # nobody closes an inbox another rank's kernel may still touch
                    c.close_inboxes()
        if a.window:
            print(f"rank {rank}: cg_kernel {kernel}", flush=True)
            assert kernel == a.expect_kernel, kernel
            dist.barrier()
            if shm is not None:
                shm.close()
                if rank == 0:
                    shm.unlink()
    ref = oracle.run(prob.xy_flat, prob.conn_flat, prob.u_known, prob.u_in, prob.f_in, prob.youngs_modulus,
                     prob.poisson_ratio, prob.part_thickness, path="sparse", precond=a.precond)
    err = np.linalg.norm(out["u"] - ref["u"]) / np.linalg.norm(ref["u"])
    os.environ.pop("MAG_TUNE_FORCE_DIST", None)
    with Context(device=0, tile_nodes=a.tile, cg_variant=a.variant, preconditioner=a.precond) as c1:
        single = c1.solve(prob)
    err1 = np.linalg.norm(out["u"] - single["u"]) / np.linalg.norm(single["u"])
    print(f"rank {rank}/{world} mode={a.mode} iters={out['iterations']} (single {single['iterations']}, oracle "
          f"{ref['iterations']}) rel-L2 vs oracle {err:.2e} vs single-rank {err1:.2e}", flush=True)
    ok = out["converged"] == 1 and err <= 1e-8 and err1 <= 1e-9 and abs(out["iterations"] - single["iterations"]) <= 5
    # every rank holds the full, identical solution
    if world > 1:
        t = torch.from_numpy(out["u"].copy())
        dist.broadcast(t, src=0)
        ok = ok and np.array_equal(t.numpy(), out["u"])
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
