#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, one JSON line on rank 0.

A "step" is one full pass of the hot path (mag_run: node ordering, K_e build, CSR assembly, BC elimination,
matrix-free CG to the configured tolerance, reactions, stress -- solver.rs:548-583) over a synthetic mesh
whose flat arrays are already resident in HBM (mag_upload happens before the timed region).

  N = 1 : BASELINE config 3, the configuration the metric is quoted on: ~1M-triangle plate with a hole,
          left edge fixed, right edge ux = delta, CG to relative residual 1e-8.
  N > 1 : weak scaling -- every GPU owns ~1M triangles of one global plate N times as tall (strips along y,
          one RCCL all-reduce of [halo residual | dot partials] per CG kernel pair).

value = elements of the global mesh * steps / wall time of the timed region (max over ranks).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def build_problem(workload, n_gpus):
    import numpy as np

    from magnetite_amd import meshgen
    if workload == "hole1m":
        n = meshgen.grid_for_triangles(1e6, np.pi * 0.15 ** 2)
        if n_gpus == 1:
            mesh = meshgen.plate_with_holes(n)
        else:  # the same plate stacked n_gpus times along y, one hole per unit square
            holes = [(0.5, (k + 0.5) / n_gpus, 0.15) for k in range(n_gpus)]
            mesh = meshgen.plate_with_holes(n, n * n_gpus, 1.0, float(n_gpus), holes=holes)
        return meshgen.config_fixed_left_pull_right(mesh), f"plate-with-hole {n}x{n * n_gpus} cells"
    if workload == "plate100k":
        return meshgen.config_fixed_left_point_load(meshgen.plate(224, 224 * n_gpus, 1.0, float(n_gpus))), "plate 224^2"
    if workload == "plate4m":
        return meshgen.config_fixed_left_pull_right(meshgen.plate(1414, 1414 * n_gpus, 1.0, float(n_gpus))), "plate 1414^2"
    if workload == "multihole16m":
        n = meshgen.grid_for_triangles(16e6, np.pi * 0.25 ** 2)
        return meshgen.config_fixed_left_pull_right(meshgen.multi_hole(n, 4, 0.25)), f"multi-hole {n}^2"
    raise SystemExit(f"unknown workload {workload}")


def cpu_baseline(prob, gpu_iterations, stop_mode, tol, sample_iters):
    """Oracle (C restatement of solver.rs, sparse path, 1 thread) timed on a bounded sample of the same workload:
    full K_e + CSR assembly + BC elimination, then `sample_iters` CG iterations; CG time is scaled to the
    iteration count the GPU needed (the oracle runs the same recurrence)."""
    import oracle
    t0 = time.perf_counter()
    K = oracle.assemble_sparse(prob.xy_flat, prob.conn_flat, prob.poisson_ratio, prob.youngs_modulus,
                               prob.part_thickness)
    A, b = oracle.reduce_system(K, prob.u_known, prob.u_in, prob.f_in)
    t1 = time.perf_counter()
    _, it, _, _ = oracle.cg(A, b, stop_mode=stop_mode, tol=tol, max_iter=sample_iters)
    t2 = time.perf_counter()
    per_iter = (t2 - t1) / max(it, 1)
    total = (t1 - t0) + per_iter * gpu_iterations
    E = prob.mesh.num_elements
    # all-cores variant of the CG (OpenMP rows + reductions); the reference itself is single-threaded
    cores = max(1, min(16, os.cpu_count() or 1))  # the GPU box's CPU share for one GPU
    t3 = time.perf_counter()
    _, itp, _ = oracle.cg_parallel(A, b, stop_mode=stop_mode, tol=tol, max_iter=sample_iters * 4, threads=cores)
    t4 = time.perf_counter()
    per_iter_par = (t4 - t3) / max(itp, 1)
    total_par = (t1 - t0) + per_iter_par * gpu_iterations
    return {
        "all_cores": {"value": E / total_par, "unit": "elements/s", "cores": cores, "kind": "port",
                      "cg_iters_per_s": 1.0 / per_iter_par,
                      "sample": f"same assembly (1 thread) + {itp} OpenMP CG iterations ({per_iter_par * 1e3:.2f} ms each)"},
        "value": E / total, "unit": "elements/s", "cores": 1, "kind": "port",
        "sample": f"oracle/magnetite_oracle.c on the same mesh: full K_e+CSR assembly+BC elimination "
                  f"({t1 - t0:.2f} s) + {it} CG iterations ({per_iter * 1e3:.2f} ms each), CG scaled to the "
                  f"{gpu_iterations} iterations of the GPU solve",
        "assembly_elements_per_s": E / (t1 - t0), "cg_iters_per_s": 1.0 / per_iter,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="hole1m")
    ap.add_argument("--tol", type=float, default=1e-8)
    ap.add_argument("--stop", default="rel", choices=["rel", "rnorm", "rnorm_sq"])
    ap.add_argument("--tile", type=int, default=0, help="0: library default (512 for >= 262144 nodes, else 256)")
    ap.add_argument("--check-every", type=int, default=64)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cg-variant", type=int, default=2,
                    help="2: on-chip single-launch CG when the mesh fits the chip (else as 1), 1: one fused launch per CG "
                         "iteration, 0: two launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-iters", type=int, default=1500,
                    help="CG iterations of the 1-thread CPU sample (default ~11 s at 1M triangles; x4 for the OpenMP leg)")
    ap.add_argument("--op-reps", type=int, default=400)
    ap.add_argument("--rehearse-dist", action="store_true",
                    help="world size 1 only: still create the nccl process group and the RCCL communicator and run "
                         "the distributed CG protocol (iteration kernel + one in-place all-reduce every iteration)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "inboxes", "allreduce", "host-window"],
                    help="N > 1: how the ranks exchange per CG iteration.  allreduce: streaming kernels + one RCCL "
                         "all-reduce per iteration; inboxes: on-chip CG on every rank, exchange through per-rank "
                         "inboxes in device memory (HIP IPC); auto (default): one untimed trial solve each way, the "
                         "faster is kept; host-window: the inbox protocol through shared host memory (slow, for tests)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="REHEARSAL of the N > 1 code path on a one-GPU box: every rank drives GPU 0 and the collectives "
                         "go through gloo (host callback) instead of RCCL; the printed line is marked, it is no result")
    args = ap.parse_args()

    import torch  # first: libmagnetite_hip.so then shares torch's HIP runtime and RCCL (same SONAMEs)
    import torch.distributed as dist

    from magnetite_amd import Context, _lib

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the solver has no CPU path")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.rehearse_dist
    if args.rehearse_dist:
        os.environ["MAG_TUNE_FORCE_DIST"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if use_dist and args.share_gpu:
        dist.init_process_group("gloo")
    elif use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    stop_mode = {"rel": _lib.MAG_STOP_REL, "rnorm": _lib.MAG_STOP_RNORM, "rnorm_sq": _lib.MAG_STOP_RNORM_SQ}[args.stop]
    prob, desc = build_problem(args.workload, world)
    E, N = prob.mesh.num_elements, prob.mesh.num_nodes

    ctx = Context(device=local_rank, stop_mode=stop_mode, tol=args.tol, tile_nodes=args.tile,
                  check_every=args.check_every, use_graph=0 if args.no_graph else 1, cg_variant=args.cg_variant)
    if use_dist and args.share_gpu:
        def host_allreduce(arr):
            dist.all_reduce(torch.from_numpy(arr))
        ctx.init_callback(host_allreduce, rank, world)
    elif use_dist:
        ctx.init_rccl_from_torch(dist, rank, world)
    shm = None
    exchange, autotune = ("RCCL all-reduce per iteration" if world > 1 else None), None
    cpu_or_gpu = "cpu" if args.share_gpu else "cuda"

    def all_agree(ok):
        t_ok = torch.tensor([1 if ok else 0], dtype=torch.int32, device=cpu_or_gpu)
        dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
        return int(t_ok.item()) == 1

    if world > 1 and args.cg_variant == 2 and args.exchange == "host-window":
        # EXPERIMENTAL: a window of shared host memory for the per-iteration exchange (mag_comm_set_window)
        from multiprocessing import shared_memory
        names = [None]
        if rank == 0:
            shm = shared_memory.SharedMemory(create=True, size=32 << 20)
            names[0] = shm.name
        dist.broadcast_object_list(names, src=0)
        ok = True
        try:
            if rank != 0:
                shm = shared_memory.SharedMemory(name=names[0])
                from multiprocessing import resource_tracker
                resource_tracker.unregister(shm._name, "shared_memory")  # rank 0 owns the segment (Python < 3.13)
            ctx.set_window(shm)
        except Exception as exc:
            print(f"rank {rank}: no host-memory window ({exc})", file=sys.stderr, flush=True)
            ok = False
        if all_agree(ok):
            exchange = "host-memory window (granules)"
        else:
            if ok:
                ctx.set_window(None)
            if shm is not None:
                shm.close()
                if rank == 0:
                    shm.unlink()
            shm = None
    ctx.upload_problem(prob)  # inputs resident in HBM before the timed region

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def open_inboxes():
        ok = True
        try:
            handles = [None] * world
            dist.all_gather_object(handles, ctx.create_inbox(32 << 20))
            ctx.open_inboxes(handles)
        except Exception as exc:  # e.g. no peer mapping between two GPUs: every rank must then do without
            print(f"rank {rank}: no device inboxes ({exc})", file=sys.stderr, flush=True)
            ok = False
        if not all_agree(ok):
            if ok:
                barrier()
                ctx.close_inboxes()
            return False
        return True

    def timed_solve():
        barrier()
        t_a = time.perf_counter()
        ctx.run()
        barrier()
        t = torch.tensor([time.perf_counter() - t_a], dtype=torch.float64, device=cpu_or_gpu)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if world > 1 and args.cg_variant == 2 and args.exchange in ("auto", "inboxes"):
        # Multi-GPU on-chip CG: every rank runs its share as one persistent launch and the per-iteration exchange goes
        # through per-rank inboxes in device memory (HIP IPC over xGMI) instead of a collective.  It cannot be
        # measured on the one-GPU development box, so the choice is made HERE, before anything is timed: one solve to
        # warm up and one timed solve each way, the faster exchange is kept (every rank sees the same two numbers).
        if open_inboxes():
            ctx.run()
            t_inbox, k_inbox = timed_solve(), int(ctx.stats()["cg_kernel"])
            barrier()
            ctx.close_inboxes()
            ctx.run()
            t_rccl = timed_solve()
            autotune = {"s_per_solve_inboxes": t_inbox, "kernel_with_inboxes": k_inbox, "s_per_solve_allreduce": t_rccl}
            if (args.exchange == "inboxes" or t_inbox < t_rccl) and k_inbox == 2 and open_inboxes():
                exchange = "per-rank inboxes in device memory (HIP IPC), on-chip CG on every rank"

    for _ in range(args.warmup):
        ctx.run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.run()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.share_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    st = ctx.stats()
    if args.tile == 0:  # what the library chose (mag_options.tile_nodes = 0): the tile size that gives its tile count
        args.tile = next(b for b in (256, 512, 1024) if (N + b - 1) // b == int(st["num_tiles"]))
    # HIP events on the library's own stream around op_reps back-to-back launches (mag_time_operator / mag_time_spmv)
    ms_op = ctx.time_operator(args.op_reps)   # the CG iteration kernel (dominant: one launch per iteration)
    ms_spmv = ctx.time_spmv(args.op_reps)     # the plain matrix-free SpMV y = M K M v
    u, _, _ = ctx.download()

    if rank == 0:
        iters = int(st["iterations"])
        ms_step = elapsed * 1e3 / args.steps
        Eloc, Nloc = E / world, N / world  # per-GPU share the operator kernel processes per launch
        spmv_bytes = 12.0 * Eloc + 50.0 * Nloc          # matrix-free SpMV
        iter_bytes = 12.0 * Eloc + 242.0 * Nloc         # full CG iteration (SpMV + 2 dots + 3 axpy)
        kind = int(st["cg_kernel"])  # what ran: 2 on-chip single launch, 1 one fused launch per iteration, 0 two launches
        if kind == 2:
            # ONE launch runs the whole CG phase (HIP events around it: stats ms_cg): its algorithmic bytes are the
            # per-iteration figure times the iterations it performed
            ms_op = st["ms_cg"]
            kernel_bytes = iter_bytes * iters
        else:
            kernel_bytes = iter_bytes if kind == 1 else (12.0 * Eloc + 50.0 * Nloc)
        achieved = kernel_bytes / (ms_op * 1e-3) / 1e9
        # mag_time_spmv applies the plain operator to the WHOLE mesh on every rank (it is not partitioned)
        spmv_bytes = 12.0 * E + 50.0 * N
        spmv_gbs = spmv_bytes / (ms_spmv * 1e-3) / 1e9
        asm_ms = st["ms_element"] + st["ms_assemble"] + st["ms_bc"]
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pmc):  # HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/README.md)
            pj = json.load(open(pmc))
            key = f"{args.workload}:tile{args.tile}:kernel{kind}"
            if key in pj and kind == 2:  # setup + per-iteration traffic, from two PMC runs of different length
                traffic = pj[key]["hbm_bytes_setup"] + pj[key]["hbm_bytes_per_iteration"] * iters
                traffic_src = "profiles/r01_pmc_summary.json:" + key
            elif key in pj:
                traffic, traffic_src = pj[key]["hbm_bytes_per_launch"], "profiles/r01_pmc_summary.json:" + key
        out = {
            "metric": "elements/sec assembly + CG iters/sec (achieved HBM GB/s), 1M-tri mesh",
            "value": E * args.steps / elapsed, "unit": "elements/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}, {E} triangles, {N} nodes, left edge fixed, "
                                   f"right edge ux=delta; full solver::run per step; CG stop={args.stop} tol={args.tol:g}",
                       "elements": E, "nodes": N, "tile_nodes": args.tile, "cg_variant": args.cg_variant, "cg_kernel": kind, "cg_stop": args.stop, "cg_tol": args.tol,
                       "parallelism": f"strips{world}" if world > 1 else "single",
                       "exchange": exchange, "exchange_autotune": autotune},
            "roofline": {"bound": "hbm",
                         "kernel": {2: "k_cg_persist<%d> (the whole CG solve in one launch: state resident in registers "
                                       "and LDS, grid-wide exchange by tagged granules every iteration)" % args.tile,
                                    1: "k_cg_fused_dma<%d> (whole CG iteration in one launch: r,x,p updates + "
                                       "matrix-free SpMV + 4 dot partials)" % args.tile,
                                    0: "k_operator_lds<%d> (matrix-free SpMV fused with p and x updates, p.q)"
                                       % args.tile}[kind],
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_launch": kernel_bytes,
                         "bytes_formula": {2: "(12E+242N) x iterations (full CG iteration, SURVEY 8d; one launch = one solve)",
                                           1: "12E+242N (full CG iteration, SURVEY 8d)",
                                           0: "12E+50N (SpMV, SURVEY 8d)"}[kind],
                         "us_per_launch": ms_op * 1e3,
                         "us_per_iteration": st["ms_cg"] * 1e3 / max(iters, 1),
                         "note": ("the state never leaves the chip: the fraction compares the bytes an unfused "
                                  "iteration would stream with the time taken, it is not HBM utilisation")
                                 if kind == 2 else None},
            "spmv": {"kernel": "k_operator_lds<%d,false> (plain matrix-free SpMV)" % args.tile, "achieved": spmv_gbs,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": spmv_gbs / HBM_PEAK_GBS,
                     "bytes_per_launch": spmv_bytes, "us_per_launch": ms_spmv * 1e3},
            "cg_iterations": iters, "cg_converged": int(st["converged"]), "cg_final_cost": st["final_cost"],
            "cg_iters_per_sec": iters / (st["ms_cg"] * 1e-3) if st["ms_cg"] > 0 else None,
            "cg_iteration_gbps": iter_bytes * iters / (st["ms_cg"] * 1e-3) / 1e9 if st["ms_cg"] > 0 else None,
            "assembly_elements_per_sec": Eloc / (asm_ms * 1e-3) if asm_ms > 0 else None,
            "phases_ms": {k: st[k] for k in ("ms_order", "ms_csr_symbolic", "ms_element", "ms_assemble", "ms_bc",
                                             "ms_cg", "ms_post", "ms_total")},
            "u_max": float(abs(u).max()),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prob, iters, stop_mode, args.tol, args.cpu_sample_iters)
            out["cpu_baseline"]["cores_available"] = os.cpu_count()
        else:
            out["cpu_baseline"] = None
        if args.share_gpu:
            out["rehearsal"] = "ranks share GPU 0, collectives through gloo: exercises the N > 1 code path, not a result"
        print(json.dumps(out), flush=True)

    if shm is not None:
        ctx.set_window(None)
    if world > 1:
        barrier()  # nobody frees an inbox another rank's kernel may still touch
    ctx.close()
    if shm is not None:
        barrier()
        shm.close()
        if rank == 0:
            shm.unlink()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
