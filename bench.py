#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, one JSON line on rank 0.

A "step" is one full pass of the hot path (mag_run: node ordering, K_e build, CSR assembly, BC elimination,
matrix-free CG to the configured tolerance, reactions, stress -- solver.rs:548-583) over a synthetic mesh
whose flat arrays are already resident in HBM (mag_upload happens before the timed region).

  N = 1 : BASELINE config 3, the configuration the metric is quoted on: ~1M-triangle plate with a hole,
          left edge fixed, right edge ux = delta, CG to relative residual 1e-8.
  N > 1 : one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), either started under
          the launcher,
              python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
                  bench.py --gpus N --steps K --warmup W
          or bare, `python bench.py --gpus N ...`: the parent then starts exactly that command as a fresh child process
          BEFORE it has imported torch or touched the GPU (a process that has initialised HIP is never re-exec'ed), lets
          rank 0's single JSON line through and exits with the children's code.
          weak scaling (default): every GPU owns ~1M triangles of one global plate N times as tall; ranks own contiguous
          Hilbert-tile ranges; per CG iteration ONE exchange of [dot partials | q on interface nodes].
          --partition strong: the workload at its BASELINE size split over the N GPUs (configs 4 and 5).

value = elements of the global mesh * steps / wall time of the timed region (max over ranks).
The line carries `roofline` (dominant kernel: bound, algorithmic bytes or flops per launch / live HIP-event time, peak,
PMC traffic), `spmv`, `hbm_resident` (the same kernels on the 16M-triangle mesh, beyond the Infinity Cache) and
`cpu_baseline` (the oracle timed on this host).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.65  # half the guide's 157.3 TFLOP/s fp32 vector peak: 256 CUs x 4 SIMDs x 16 fp64 FMA lanes
INFINITY_CACHE_BYTES = 256 << 20
PMC_SUMMARY = os.path.join(ROOT, "profiles", "pmc_summary.json")  # scripts/pmc_summarize.py (not round-specific)
CPU_RECORD = os.path.join(ROOT, "profiles", "cpu_baseline.json")  # scripts/cpu_baseline_record.py


def bytes_spmv(E, N):
    """matrix-free SpMV -- connectivity 12E, coordinates 16N, x read 16N, y write 16N, BC mask 2N."""
    return 12.0 * E + 50.0 * N


def bytes_iteration_unfused(E, N):
    """one CG iteration as separate passes -- SpMV + dot(p,q) 16n + x-axpy 24n + r-axpy 24n + dot(r,r) 8n +
    p-update 24n (n = 2N): 12E + 242N.  What a textbook implementation streams; the fused kernel moves less."""
    return 12.0 * E + 242.0 * N


def bytes_iteration_fused(E, N):
    """What ONE fused launch per iteration must move at the least: every node's r, q, p and x
    read once and written once (4 x 16 + 4 x 16 = 128N: alpha is known only after the grid-wide reduction, so q of the
    previous launch has to come back), coordinates 16N, BC mask 1N, connectivity 12E: 12E + 145N."""
    return 12.0 * E + 145.0 * N


def flops_iteration(E, N):
    """~110 flop per element for the element-loop operator (area, B, D B u_e, B^T sigma A t) + the CG
    vector work, 2 dots (2n flop each) and 3 axpys (2n each) = 10n = 20N."""
    return 110.0 * E + 20.0 * N


def pmc_is_stale():
    """The committed counters describe the kernels of ONE build: scripts/pmc_summarize.py records the digest of the kernel
    sources the profiled library was linked from, csrc/Makefile writes the digest of the library loaded here next to it.
    True when they differ (or either is missing): the counters are then of other code than the one being timed."""
    from magnetite_amd import _lib
    if not os.path.exists(PMC_SUMMARY):
        return True
    recorded = json.load(open(PMC_SUMMARY)).get("_meta", {}).get("source_hash")
    return recorded is None or recorded != _lib.built_source_hash()


def load_pmc(key):
    if os.path.exists(PMC_SUMMARY):
        e = json.load(open(PMC_SUMMARY)).get(key)
        if e is not None:
            e = dict(e, stale=pmc_is_stale())
        return e
    return None


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
    if workload == "frontal1m":
        # what gmsh's frontal mesher hands solver::run (mesher.rs:501-506): an UNSTRUCTURED mesh, a quarter of its nodes with
        # seven or more neighbours (meshgen.frontal_like: jittered equilateral lattice, Delaunay); 1 006 602 triangles
        if n_gpus != 1:
            raise SystemExit("frontal1m is a single-GPU workload")
        return meshgen.config_fixed_left_pull_right(meshgen.frontal_like(660, 0.4, 1)), "frontal-like Delaunay mesh, 660 pitch"
    if workload == "plate100k":
        return meshgen.config_fixed_left_point_load(meshgen.plate(224, 224 * n_gpus, 1.0, float(n_gpus))), "plate 224^2"
    if workload == "plate4m":
        return meshgen.config_fixed_left_pull_right(meshgen.plate(1414, 1414 * n_gpus, 1.0, float(n_gpus))), "plate 1414^2"
    if workload == "multihole16m":
        n = meshgen.grid_for_triangles(16e6, np.pi * 0.25 ** 2)
        return meshgen.config_fixed_left_pull_right(meshgen.multi_hole(n, 4, 0.25)), f"multi-hole {n}^2"
    if workload.startswith("plate:") or workload.startswith("frontal:"):  # size sweeps: plate:<cells per side>, frontal:<pitch>
        if n_gpus != 1:
            raise SystemExit(f"{workload} is a single-GPU workload")
        kind, n = workload.split(":")
        mesh = meshgen.plate(int(n), int(n)) if kind == "plate" else meshgen.frontal_like(int(n), 0.4, 1)
        return meshgen.config_fixed_left_pull_right(mesh), f"{kind} {n}"
    raise SystemExit(f"unknown workload {workload}")


def usable_cores():
    """Host cores this process may actually run on: the scheduler affinity mask, cut to the cgroup's CPU quota when
    one is set (a GPU box hands a 1-GPU job a share of the host, not all of it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(prob, gpu_iterations, stop_mode, tol, sample_iters, threads=0):
    """Oracle (C restatement of solver.rs, sparse path) timed on a bounded sample of the same workload, twice:
    1 thread (the reference is single-threaded) and OpenMP on every core this process may use.  Each leg:
    full K_e + CSR assembly + BC elimination (timed around the C calls), then a sample of CG iterations; the CG time is
    scaled to the iteration count the GPU needed (the oracle runs the same recurrence)."""
    import ctypes as C

    import numpy as np

    import oracle
    L = oracle.lib()
    xy, conn = oracle._prep(prob.xy_flat, prob.conn_flat)
    N, E = xy.size // 2, conn.size // 3
    uk = np.ascontiguousarray(prob.u_known, dtype=np.uint8)
    ui, fi = np.ascontiguousarray(prob.u_in, dtype=np.float64), np.ascontiguousarray(prob.f_in, dtype=np.float64)
    nf = int(2 * N - uk.sum())

    def assemble(omp):
        b = np.zeros(max(nf, 1))
        fa = L.orc_assemble_sparse_omp if omp else L.orc_assemble_sparse
        fr = L.orc_reduce_system_omp if omp else L.orc_reduce_system
        t0 = time.perf_counter()
        K = fa(N, E, oracle._d(xy), oracle._i(conn), float(prob.poisson_ratio), float(prob.youngs_modulus),
               float(prob.part_thickness))
        A = fr(K, oracle._b(uk), oracle._d(ui), oracle._d(fi), oracle._d(b))
        dt = time.perf_counter() - t0
        L.orc_csr_free(K)
        return oracle.Csr(A), b[:nf], dt

    A, b, t_asm = assemble(False)
    t1 = time.perf_counter()
    _, it, _, _ = oracle.cg(A, b, stop_mode=stop_mode, tol=tol, max_iter=sample_iters)
    per_iter = (time.perf_counter() - t1) / max(it, 1)
    total = t_asm + per_iter * gpu_iterations
    cores = threads or usable_cores()
    oracle.set_threads(cores)
    A2, b2, t_asm_par = assemble(True)
    same = bool(np.array_equal(A2.val, A.val) and np.array_equal(A2.col, A.col) and np.array_equal(b2, b))
    t3 = time.perf_counter()
    _, itp, _ = oracle.cg_parallel(A, b, stop_mode=stop_mode, tol=tol, max_iter=sample_iters * 4, threads=cores)
    per_iter_par = (time.perf_counter() - t3) / max(itp, 1)
    total_par = t_asm_par + per_iter_par * gpu_iterations
    return {
        "all_cores": {"value": E / total_par, "unit": "elements/s", "cores": cores, "kind": "port",
                      "cores_on_host": os.cpu_count(),
                      "cores_rule": "affinity mask cut to the cgroup CPU quota (usable_cores)" if not threads else "--cpu-threads",
                      "assembly_elements_per_s": E / t_asm_par, "cg_iters_per_s": 1.0 / per_iter_par,
                      "assembly_bit_identical_to_1_thread": same,
                      "sample": f"OpenMP K_e+CSR assembly+BC elimination over row nodes ({t_asm_par:.3f} s) + {itp} OpenMP "
                                f"CG iterations ({per_iter_par * 1e3:.2f} ms each), {cores} threads, CG scaled to the "
                                f"{gpu_iterations} iterations of the GPU solve"},
        "value": E / total, "unit": "elements/s", "cores": 1, "kind": "port",
        "sample": f"oracle/magnetite_oracle.c on the same mesh: full K_e+CSR assembly+BC elimination "
                  f"({t_asm:.2f} s) + {it} CG iterations ({per_iter * 1e3:.2f} ms each), CG scaled to the "
                  f"{gpu_iterations} iterations of the GPU solve",
        "assembly_elements_per_s": E / t_asm, "cg_iters_per_s": 1.0 / per_iter,
    }


def fixture_parity(workload, prob, u, f, stress, iterations, stop, tol):
    """This rank's returned solution against the oracle's sampled solution of the workload at BASELINE size
    (tests/golden/fullsize_<workload>.npz, generator tests/golden/make_fullsize_fixtures.py): the mesh must be the
    fixture's (sizes and checksums), the iteration count the oracle's, u within 1e-8 relative L2 at the sampled DOFs,
    reactions and stress within 1e-7 (first differences of an iterative solution: tests/test_fullsize_parity_gpu.py)."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", f"fullsize_{workload}.npz")
    fx = np.load(path, allow_pickle=False)
    N, E = prob.mesh.num_nodes, prob.mesh.num_elements
    mesh_ok = ((N, E) == (int(fx["num_nodes"]), int(fx["num_elements"]))
               and float(np.sum(prob.xy_flat * np.arange(1, 2 * N + 1) % 7.0)) == float(fx["xy_checksum"])
               and int(np.sum(prob.conn_flat.astype(np.int64) * (np.arange(3 * E) % 11 + 1))) == int(fx["conn_checksum"]))
    rule_ok = stop == "rel" and tol == float(fx["rel_tol"])
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    iu, ie = fx["dof_idx"], fx["elem_idx"]
    slack = 0 if "1 thread" in str(fx["solver"]) else max(2, int(fx["iterations"]) // 1000)
    if workload == "frontal1m":  # the edge blocks add a node's triangles in another order than the oracle's rows: the count
        slack = max(2, int(fx["iterations"]) // 100)  # wobbles at the threshold (plain CG's residual is not monotone)
    d = {"fixture": os.path.relpath(path, ROOT), "oracle_iterations": int(fx["iterations"]), "iterations": int(iterations),
         "mesh_is_the_fixtures": bool(mesh_ok), "stop_rule_is_the_fixtures": bool(rule_ok),
         "rel_l2_u_sampled": rel(u[iu], fx["u_at"]),
         "u_norm_rel_diff": abs(float(np.linalg.norm(u)) - float(fx["u_norm"])) / float(fx["u_norm"]),
         "f_sampled_max_over_reaction_norm": float(np.abs(f[iu] - fx["f_at"]).max() / float(fx["f_known_norm"])),
         "rel_l2_stress_sampled": rel(stress[ie], fx["stress_at"]), "bars": {"u": 1e-8, "f": 1e-7, "stress": 1e-7}}
    d["ok"] = bool(mesh_ok and rule_ok and abs(d["iterations"] - d["oracle_iterations"]) <= slack
                   and d["rel_l2_u_sampled"] <= 1e-8 and d["u_norm_rel_diff"] <= 1e-8
                   and d["f_sampled_max_over_reaction_norm"] <= 1e-7 and d["rel_l2_stress_sampled"] <= 1e-7)
    return d


def kernel_line(kernel, nbytes, formula, ms_per_launch, pmc):
    """HBM-roofline entry of one streaming kernel: algorithmic bytes per launch / live HIP-event time per launch."""
    gbs = nbytes / (ms_per_launch * 1e-3) / 1e9
    d = {"kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": gbs / HBM_PEAK_GBS, "bytes_per_launch": nbytes, "bytes_formula": formula,
         "us_per_launch": ms_per_launch * 1e3, "traffic": None, "traffic_source": None,
         # a working set below the 256 MiB Infinity Cache is served on-die across back-to-back launches: the figure is
         # then a cache-resident rate, not an HBM-resident one (hbm_resident carries that)
         "working_set_fits_infinity_cache": bool(nbytes < INFINITY_CACHE_BYTES)}
    if pmc and "hbm_bytes_per_launch" in pmc:
        d["traffic"] = pmc["hbm_bytes_per_launch"]
        d["traffic_source"] = "profiles/pmc_summary.json (2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes)"
        d["traffic_stale"] = bool(pmc["stale"])  # true: counters of another build of the kernels than the one timed
    return d


def roofline_streaming(kind, E, N, ms_op, pmc, tile):
    if kind == 1:
        return kernel_line("k_cg_fused_dma<%d> (whole CG iteration in one launch: r, x, p updates + matrix-free SpMV + "
                           "4 dot partials)" % tile, bytes_iteration_fused(E, N),
                           "12E+145N: what one fused launch per iteration must move (DESIGN.md section 5); the unfused "
                           "iteration of SURVEY 8d would stream 12E+242N", ms_op, pmc)
    return kernel_line("k_operator_lds<%d> (matrix-free SpMV fused with the p and x updates and p.q)" % tile,
                       bytes_spmv(E, N) + 48.0 * N, "12E+50N (SpMV, SURVEY 8d) + 48N (p read, p and x written)", ms_op, pmc)


def roofline_onchip(E, N, iters, ms_cg, pmc, tile, edge_blocks=False, tiles_per_workgroup=None):
    """k_cg_persist keeps the CG state in registers and LDS: HBM only sees the per-iteration exchange, so an HBM roof
    says nothing about it.  Its largest counted resource is fp64 vector issue (PMC: SQ_ACTIVE_INST_VALU), so it is
    priced against the fp64 vector peak with the ALGORITHMIC flops of the iterations it ran; the counted utilisations
    of every resource (separate rocprofv3 --pmc passes, profiles/pmc_summary.json) ride along."""
    flops = flops_iteration(E, N) * iters
    tf = flops / (ms_cg * 1e-3) / 1e12
    d = {"kernel": "k_cg_persist<%d> (the whole CG solve in ONE launch: state resident in registers and LDS, grid-wide "
                   "exchange by tagged granules every iteration; %s)"
                   % (tile, ("edge-block instantiation with overflow: six symmetric 2 x 2 blocks per node in registers, the blocks of "
                             "longer rows in an LDS pool (unstructured meshes)") if int(edge_blocks) == 2
                      else ("edge-block instantiation: six symmetric 2 x 2 blocks per node in registers" if edge_blocks
                            else "triangle-walk instantiation: cached triangle weights")),
         "edge_blocks": bool(edge_blocks), "edge_block_mode": int(edge_blocks),
         # 512-node tiles per workgroup; below four the instantiation with that many node slots per lane runs (DESIGN R4.6)
         "tiles_per_workgroup": tiles_per_workgroup,
         "bound": "valu-fp64", "achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
         "frac": tf / FP64_VECTOR_PEAK_TFLOPS, "flops_per_launch": flops,
         "flops_formula": "(110E + 20N) x iterations (SURVEY 8d: ~110 flop per element-loop SpMV element + 10 flop per "
                          "DOF of CG vector work); one launch = one solve",
         "us_per_launch": ms_cg * 1e3, "traffic": None, "traffic_source": None, "counted": None}
    if pmc:
        t_it = ms_cg * 1e-3 / max(iters, 1)
        d["traffic"] = pmc["hbm_bytes_setup"] + pmc["hbm_bytes_per_iteration"] * iters
        d["traffic_source"] = "profiles/pmc_summary.json (2*FETCH_SIZE + WRITE_SIZE; runs of two lengths)"
        d["traffic_stale"] = bool(pmc["stale"])
        if not pmc["stale"]:  # utilisations of another build say nothing about this one: dropped, not quoted
            d["counted"] = dict(pmc.get("utilisation", {}))
            d["counted"]["hbm_frac_live"] = pmc["hbm_bytes_per_iteration"] / t_it / 1e9 / HBM_PEAK_GBS
            d["counted"]["source"] = "profiles/pmc_summary.json, per CG iteration (difference of two run lengths)"
    return d


def hbm_resident_leg(device, reps):
    """The same two kernels on BASELINE config 5's 16M-triangle mesh, whose working set (0.6 GB SpMV, 1.4 GB iteration)
    cannot sit in the 256 MiB Infinity Cache: the figure the north star's ">= 40 % of peak HBM bandwidth in the CG
    SpMV" has to stand on.  No solve (20 000 iterations): mag_time_* only need the symbolic phase."""
    from magnetite_amd import Context
    t0 = time.perf_counter()
    prob, desc = build_problem("multihole16m", 1)
    E, N = prob.mesh.num_elements, prob.mesh.num_nodes
    with Context(device=device, cg_variant=1) as c:
        c.upload_problem(prob)
        ms_it = c.time_operator(reps)
        ms_sp = c.time_spmv(reps)
    tile = 512 if N >= 512 * 512 else 256  # the library's automatic choice (mag_options.tile_nodes = 0)
    key = f"multihole16m:tile{tile}"
    return {"workload": f"multihole16m: {desc}, {E} triangles, {N} nodes (BASELINE config 5 geometry, one GPU)",
            "elements": E, "nodes": N, "tile_nodes": tile, "launches_timed": reps,
            "spmv": kernel_line("k_operator_lds<%d,false> (plain matrix-free SpMV)" % tile, bytes_spmv(E, N),
                                "12E+50N (SpMV, SURVEY 8d)", ms_sp, load_pmc(key + ":spmv")),
            "iteration": kernel_line("k_cg_fused_dma<%d> (whole CG iteration in one launch)" % tile,
                                     bytes_iteration_fused(E, N), "12E+145N (fused iteration, DESIGN.md section 5)",
                                     ms_it, load_pmc(key + ":kernel1")),
            "seconds": time.perf_counter() - t0}


def unstructured_leg(device, tol, reps=3):
    """The drop-in case next to the headline: solver::run is handed gmsh meshes (mesher.rs:501-506), not structured plates.
    frontal1m = 1 006 602 triangles of a jittered equilateral lattice, Delaunay (meshgen.frontal_like: a quarter of the
    nodes with seven or more neighbours, like a frontal mesh): `reps` whole mag_run steps after one warm-up, the same stop
    rule as the headline, the solution verified by its true residual."""
    import numpy as np

    from magnetite_amd import Context, _lib
    t0 = time.perf_counter()
    prob, desc = build_problem("frontal1m", 1)
    E, N = prob.mesh.num_elements, prob.mesh.num_nodes
    with Context(device=device, stop_mode=_lib.MAG_STOP_REL, tol=tol) as c:
        c.upload_problem(prob)
        c.run()
        ms = []
        for _ in range(reps):
            c.run()
            ms.append(c.stats()["ms_total"])
        st = c.stats()
        u, f_out, s_out = c.download()
        r = prob.f_in - c.apply_operator(u)  # f - K u on every DOF; the unknowns are where u is not prescribed
        free = prob.u_known == 0
        verify = float(np.linalg.norm(r[free]) / max(st["rhs_norm"], 1e-300))
    # against the oracle's sampled solution at full size: one more solve at the FIXTURE's tolerance (1e-10: at a relative 1e-8
    # the solution of this mesh is only determined to ~2e-8, tests/golden/make_fullsize_fixtures.py)
    parity = None
    fpath = os.path.join(ROOT, "tests", "golden", "fullsize_frontal1m.npz")
    if os.path.exists(fpath):
        ftol = float(np.load(fpath, allow_pickle=False)["rel_tol"])
        with Context(device=device, stop_mode=_lib.MAG_STOP_REL, tol=ftol) as c:
            c.upload_problem(prob)
            c.run()
            fst = c.stats()
            fu, ff, fs = c.download()
        parity = fixture_parity("frontal1m", prob, fu, ff, fs, int(fst["iterations"]), "rel", ftol)
        parity["tol"] = ftol
    mean_ms = sum(ms) / len(ms)
    return {"workload": f"frontal1m: {desc}, {E} triangles, {N} nodes, left edge fixed, right edge ux=delta; CG stop=rel tol={tol:g}",
            "elements": E, "nodes": N, "steps": reps, "ms_per_step": mean_ms, "value": E / (mean_ms * 1e-3), "unit": "elements/s",
            "cg_kernel": int(st["cg_kernel"]), "edge_blocks": int(st["edge_blocks"]),
            "tiles_per_workgroup": int(st.get("tiles_per_workgroup", 0)), "iterations": int(st["iterations"]),
            "us_per_iteration": st["ms_cg"] * 1e3 / max(int(st["iterations"]), 1), "converged": int(st["converged"]),
            "verify_rel_residual": verify, "fixture_parity": parity, "phases_ms": {k: st[k] for k in ("ms_order", "ms_csr_symbolic", "ms_assemble",
                                                                           "ms_bc", "ms_cg", "ms_post", "ms_total")},
            "seconds": time.perf_counter() - t0}


def spawn_ranks(n):
    """`python bench.py --gpus N` started bare: run N ranks of this same command line under torch.distributed.run as a
    fresh CHILD process (this parent has made no GPU call and never will), stdout/stderr inherited so that rank 0's
    single JSON line is what the caller reads; returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:  # a free port for the rendezvous on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="hole1m")
    ap.add_argument("--tol", type=float, default=1e-8)
    ap.add_argument("--stop", default="rel", choices=["rel", "rnorm", "rnorm_sq"])
    ap.add_argument("--max-iter", type=int, default=0, help="CG iteration cap (0: the reference's MAX_CG_ITER = 1e7)")
    ap.add_argument("--tile", type=int, default=0, help="0: library default (512 for >= 262144 nodes, else 256)")
    ap.add_argument("--check-every", type=int, default=64)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cg-variant", type=int, default=2,
                    help="2: on-chip single-launch CG when the mesh fits the chip (else as 1), 1: one fused launch per CG "
                         "iteration, 0: two launches")
    ap.add_argument("--precision", default="fp64", choices=["fp64", "fp32"],
                    help="fp32: the fp32 leg of BASELINE config 5's sweep (CG state and operator in fp32, dots in fp64; "
                         "streaming kernels; cannot meet the 1e-8 parity bar) -- the line then says dtype f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-unstructured", action="store_true",
                    help="N = 1, default workload: skip the unstructured leg (whole solves of the 1M-triangle frontal-like mesh)")
    ap.add_argument("--no-hbm-resident", action="store_true",
                    help="N = 1: skip the HBM-resident leg (SpMV and iteration kernel timed on the 16M-triangle mesh)")
    ap.add_argument("--partition", default="weak", choices=["weak", "strong"],
                    help="N > 1.  weak (default, the scaling curve): every GPU gets the workload's mesh, stacked N times "
                         "along y.  strong: the workload at its BASELINE size split over the N GPUs (config 4 = "
                         "--gpus 4 --workload plate4m --partition strong; config 5 = --gpus 8 --workload multihole16m "
                         "--partition strong)")
    ap.add_argument("--cpu-sample-iters", type=int, default=1500,
                    help="CG iterations of the 1-thread CPU sample (default ~11 s at 1M triangles; x4 for the OpenMP leg)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the all-cores CPU leg (default: every core this process may use, usable_cores())")
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
    ap.add_argument("--check-fixture", action="store_true",
                    help="compare the returned solution on EVERY rank with the oracle's sampled solution of this workload "
                         "at BASELINE size (tests/golden/fullsize_<workload>.npz: iteration count, u / f / stress at 4096 "
                         "positions; needs --partition strong at N > 1, --stop rel and the fixture's tolerance); the line "
                         "carries `fixture_parity`, exit code 6 on a miss")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

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
    prob, desc = build_problem(args.workload, world if args.partition == "weak" else 1)
    E, N = prob.mesh.num_elements, prob.mesh.num_nodes

    ctx = Context(device=local_rank, stop_mode=stop_mode, tol=args.tol, tile_nodes=args.tile,
                  check_every=args.check_every, use_graph=0 if args.no_graph else 1, cg_variant=args.cg_variant,
                  precision=1 if args.precision == "fp32" else 0, **({"max_iter": args.max_iter} if args.max_iter else {}))
    if use_dist and args.share_gpu:
        def host_allreduce(arr):
            dist.all_reduce(torch.from_numpy(arr))
        ctx.init_callback(host_allreduce, rank, world)
    elif use_dist:
        ctx.init_rccl_from_torch(dist, rank, world)
    shm = None
    exchange, autotune = None, None
    if world > 1:  # what carries the per-iteration all-reduce: RCCL on the library's stream, or (--share-gpu) gloo on the host
        exchange = ("all-reduce per iteration through the host callback (gloo; --share-gpu rehearsal)" if args.share_gpu
                    else "RCCL all-reduce per iteration")
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

    if world > 1 and args.exchange in ("auto", "inboxes"):
        # Multi-GPU on-chip CG: every rank runs its share as one persistent launch and the per-iteration exchange goes
        # through per-rank inboxes in device memory (HIP IPC over xGMI) instead of a collective.  It cannot be
        # measured on the one-GPU development box, so the choice is made HERE, before anything is timed: one solve to
        # warm up and one timed solve each way, the faster exchange is kept (every rank sees the same two numbers).
        if open_inboxes():
            import numpy as np
            ctx.run()
            t_inbox, st_in = timed_solve(), ctx.stats()
            k_inbox, it_inbox, ex_inbox = int(st_in["cg_kernel"]), int(st_in["iterations"]), int(st_in["exchange"])
            u_inbox = ctx.download()[0]
            barrier()
            ctx.close_inboxes()
            ctx.run()
            t_rccl, it_rccl = timed_solve(), int(ctx.stats()["iterations"])
            u_rccl = ctx.download()[0]
            # the inbox path has never run across two devices before this node: it is kept only if it reproduces the
            # all-reduce solve (same recurrence, so the same iteration count up to round-off at the threshold, and
            # displacements within the parity bar) on EVERY rank -- being faster is not enough
            du = float(np.linalg.norm(u_inbox - u_rccl) / max(np.linalg.norm(u_rccl), 1e-300))
            # (exchange 2: on-chip kernels through the inboxes; 3: the mesh does not fit the chips, streaming kernels
            # through the inboxes; 1: the inbox path gave up and fell back)
            same = all_agree(ex_inbox in (2, 3) and abs(it_inbox - it_rccl) <= max(2, it_rccl // 1000) and du <= 1e-8)
            autotune = {"s_per_solve_inboxes": t_inbox, "kernel_with_inboxes": k_inbox, "exchange_with_inboxes": ex_inbox,
                        "s_per_solve_allreduce": t_rccl,
                        "iterations_inboxes": it_inbox, "iterations_allreduce": it_rccl, "rel_l2_u_between_them": du,
                        "solutions_agree_on_every_rank": same}
            if same and (args.exchange == "inboxes" or t_inbox < t_rccl) and open_inboxes():
                exchange = ("per-rank inboxes in device memory (HIP IPC), " +
                            ("on-chip CG on every rank" if ex_inbox == 2 else "streaming kernels (k_stream_exchange)"))

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
    # HIP events on the library's own stream around op_reps back-to-back launches (mag_time_operator / mag_time_spmv);
    # with several ranks each launch covers the tiles this rank owns (its per-GPU share)
    ms_op = ctx.time_operator(args.op_reps)   # the streaming CG iteration kernel (one launch per iteration)
    ms_spmv = ctx.time_spmv(args.op_reps)     # the plain matrix-free SpMV y = M K M v
    u, f_out, s_out = ctx.download()
    # Self-check of the returned solution, independent of how it was produced and gathered: the true residual
    # f - K u on the unknown DOFs, with K applied matrix-free to the whole mesh by this rank alone, against |b|.  A
    # converged solve sits at about the stop tolerance; an exchange or gather that went wrong would not.
    import numpy as np
    y = ctx.apply_operator(u, masked=False)
    free = prob.u_known == 0
    verify = float(np.linalg.norm((prob.f_in - y)[free]) / max(st["rhs_norm"], 1e-300))
    verify_bar = max(1e-6, 100.0 * args.tol) if args.precision == "fp64" else 1e-3
    verify_ok = bool(verify <= verify_bar) if st["converged"] else None
    if use_dist:
        verify_ok = all_agree(verify_ok is not False)
    parity = None
    if args.check_fixture:
        if world > 1 and args.partition != "strong":
            raise SystemExit("--check-fixture needs the workload at its BASELINE size: --partition strong")
        parity = fixture_parity(args.workload, prob, u, f_out, s_out, st["iterations"], args.stop, args.tol)
        parity["ok_on_every_rank"] = all_agree(parity["ok"]) if use_dist else parity["ok"]
        if not parity["ok"]:
            print(f"rank {rank}: fixture parity missed: {json.dumps(parity)}", file=sys.stderr, flush=True)
    capped = not st["converged"]  # MAG_TERM_MAX_ITERS (or a breakdown): a normal return of the library, not a bench result
    comm = ctx.comm_info()
    fallback = bool(st["persist_timeout"])  # the on-chip kernel was chosen, gave up at its grid barrier, streaming redid it

    if rank == 0:
        iters = int(st["iterations"])
        ms_step = elapsed * 1e3 / args.steps
        Eloc, Nloc = E / world, N / world  # per-GPU share: what one launch of the kernels processes
        kind = int(st["cg_kernel"])  # what ran: 2 on-chip single launch, 1 one fused launch per iteration, 0 two launches
        tile_key = f"{args.workload}:tile{args.tile}"
        if world > 1:  # the committed PMC passes are single-GPU runs of the single-GPU kernels: no counters for N > 1
            tile_key = "none"
        if kind == 2:
            roofline = roofline_onchip(Eloc, Nloc, iters, st["ms_cg"], load_pmc(f"{tile_key}:kernel2"), args.tile,
                                       int(st.get("edge_blocks", 0)), int(st.get("tiles_per_workgroup", 0)) or None)
        elif kind == 4:  # fp32 leg: value terms halved (r, q, p, x in and out 64N, coordinates 8N, mask 1N); no timing
            # hook of its own, so the launch time is the CG phase / iterations (graph gaps and early exits included)
            roofline = kernel_line("k_cg_fused32<%d> (whole CG iteration in one launch, fp32 state)" % args.tile,
                                   12.0 * Eloc + 73.0 * Nloc, "12E+73N (fused iteration, fp32 values)",
                                   st["ms_cg"] / max(iters, 1), None)
        else:
            roofline = roofline_streaming(kind, Eloc, Nloc, ms_op, load_pmc(f"{tile_key}:kernel{kind}"), args.tile)
        roofline["us_per_iteration"] = st["ms_cg"] * 1e3 / max(iters, 1)
        spmv = kernel_line("k_operator_lds<%d,false> (plain matrix-free SpMV y = M K M v)" % args.tile,
                           bytes_spmv(Eloc, Nloc), "12E+50N (SpMV, SURVEY 8d), per-GPU share", ms_spmv,
                           load_pmc(f"{tile_key}:spmv"))
        asm_ms = st["ms_element"] + st["ms_assemble"] + st["ms_bc"]
        out = {
            # BASELINE.json's metric; quoted on the 1M-triangle mesh, other workloads say which mesh they ran
            "metric": "elements/sec assembly + CG iters/sec (achieved HBM GB/s), " +
                      ("1M-tri mesh" if args.workload == "hole1m" else f"{args.workload} mesh"),
            "value": E * args.steps / elapsed, "unit": "elements/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            # N = 1 is the base point of the series the driver runs with these flags (weak: every added GPU adds one
            # copy of the workload's mesh; strong: the same mesh split over more GPUs)
            "higher_is_better": True, "scaling": args.partition, "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}, {E} triangles, {N} nodes, left edge fixed, "
                                   f"right edge ux=delta; full solver::run per step; CG stop={args.stop} tol={args.tol:g}",
                       "elements": E, "nodes": N, "tile_nodes": args.tile, "cg_variant": args.cg_variant, "cg_kernel": kind, "cg_stop": args.stop, "cg_tol": args.tol,
                       "parallelism": f"hilbert-tile-ranges{world}" if world > 1 else "single",
                       "partition": args.partition if world > 1 else None,
                       "exchange": exchange, "exchange_kind": int(st["exchange"]), "exchange_autotune": autotune,
                       "ranks": comm["ranks"], "transport": comm["transport"], "rccl_ranks": comm["rccl_ranks"]},
            "roofline": roofline,
            "spmv": spmv,
            "fallback": fallback,
            "verify": {"rel_true_residual": verify, "bar": verify_bar, "ok_on_every_rank": verify_ok,
                       "what": "|f - K u| on the unknown DOFs / |b|, K applied matrix-free to the whole mesh"},
            "cg_iterations": iters, "cg_converged": int(st["converged"]), "cg_final_cost": st["final_cost"],
            "cg_termination": {0: "none", 1: "target_cost", 2: "max_iters", 3: "breakdown"}.get(int(st["termination"])),
            "cg_best_iteration": int(st["best_iteration"]),
            # stopped at the iteration cap (solver.rs:149-176 returns Ok(best_param) there, and so does mag_run): the
            # steps then measured unconverged solves, each running the CG twice (best_param rerun) -- not a result
            "capped": bool(capped), "fixture_parity": parity,
            "cg_iters_per_sec": iters / (st["ms_cg"] * 1e-3) if st["ms_cg"] > 0 else None,
            # the SURVEY 8(d) figure of the metric's name: bytes an UNFUSED iteration streams / time per iteration.  An
            # equivalent rate, not HBM utilisation (the fused and on-chip kernels move far less): see roofline
            "cg_iteration_gbps_unfused_equivalent":
                bytes_iteration_unfused(E, N) * iters / (st["ms_cg"] * 1e-3) / 1e9 if st["ms_cg"] > 0 else None,
            "assembly_elements_per_sec": E / (asm_ms * 1e-3) if asm_ms > 0 else None,
            # numeric assembly (K_e + CSR rows per element tile + BC elimination) against the HBM roof with SURVEY 8(d)'s
            # bytes (k_assemble_fan: fp64-issue-bound; MAG_TUNE_ASSEMBLY selects the older kernels); the
            # CSR pattern is ms_csr_symbolic
            "assembly": ({"kernel": "k_assemble_fan + k_rhs_touched (b = 0.0 + f of the other rows comes from the ordering phase)", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                          "bytes": 12.0 * Eloc + 240.0 * Nloc, "bytes_formula": "12E+240N (SURVEY 8d), per-GPU share",
                          "ms": asm_ms, "achieved": (12.0 * Eloc + 240.0 * Nloc) / (asm_ms * 1e-3) / 1e9,
                          "frac": (12.0 * Eloc + 240.0 * Nloc) / (asm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                         if asm_ms > 0 else None),
            "phases_ms": {k: st[k] for k in ("ms_order", "ms_csr_symbolic", "ms_element", "ms_assemble", "ms_bc",
                                             "ms_cg", "ms_post", "ms_total")},
            "u_max": float(abs(u).max()),
        }
        if world == 1 and not args.no_hbm_resident and args.workload != "multihole16m":
            ctx.close()  # its buffers are not needed any more; the 16M leg allocates ~5 GB of its own
            out["hbm_resident"] = hbm_resident_leg(local_rank, args.op_reps)
        if world == 1 and not args.no_unstructured and args.workload == "hole1m":
            ctx.close()  # (idempotent)
            out["unstructured"] = unstructured_leg(local_rank, args.tol)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prob, iters, stop_mode, args.tol, args.cpu_sample_iters, args.cpu_threads)
            out["cpu_baseline"]["cores_available"] = os.cpu_count()
            rec = CPU_RECORD if os.path.exists(CPU_RECORD) else os.path.join(ROOT, "profiles", "r02_cpu_baseline.json")
            if os.path.exists(rec):  # scripts/cpu_baseline_record.py: BASELINE.md section 2 items 1-3, timed once
                rj = json.load(open(rec))
                out["cpu_baseline"]["config1_dense_s"] = rj.get("config1_dense_s")
                c3 = rj.get("config3_hole1m", {}).get("sparse_1_thread")
                if c3:  # the UNSCALED 1-thread solve of this very workload, timed once on rj["host"]
                    out["cpu_baseline"]["config3_full_solve_s"] = c3["assembly_and_bc_s"] + c3["cg_s"]
                    out["cpu_baseline"]["config3_full_solve_elements_per_s"] = c3["elements_per_s"]
                out["cpu_baseline"]["record_host"] = rj.get("host")
                out["cpu_baseline"]["record"] = os.path.relpath(rec, ROOT)
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
    if capped:
        print(f"bench.py: the CG stopped without reaching its target (termination {int(st['termination'])}, "
              f"{int(st['iterations'])} iterations, best iterate {int(st['best_iteration'])}): the line is marked "
              f"\"capped\": true and is not a result", file=sys.stderr, flush=True)
        sys.exit(5)
    if parity is not None and not parity.get("ok_on_every_rank"):
        print("bench.py: the returned solution misses the oracle fixture on some rank (fixture_parity)", file=sys.stderr,
              flush=True)
        sys.exit(6)
    if verify_ok is False:
        print(f"bench.py: the returned solution does not satisfy K u = f (relative true residual {verify:.3e})",
              file=sys.stderr, flush=True)
        sys.exit(4)
    if fallback:
        # the line above is marked "fallback": true; a run that silently streamed where the on-chip kernel was chosen
        # must not pass for a result of the default configuration
        print("bench.py: the on-chip CG kernel timed out at its grid barrier and the streaming kernels redid the "
              "solve (stats.persist_timeout)", file=sys.stderr, flush=True)
        sys.exit(3)


if __name__ == "__main__":
    main()
