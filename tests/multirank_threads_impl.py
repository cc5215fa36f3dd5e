"""(run by tests/test_multirank_threads_gpu.py in a process of its own) EIGHT ranks on the one GPU of the test box -- the rank count of BASELINE configs 4/5 and of the driver's
scaling run -- as eight THREADS of this process, one mag_ctx (one stream) each.

The GPU box admits at most six processes on its card, so the process-per-rank rehearsals of test_distributed_gpu.py
stop at four ranks; mag_ctx is documented as usable from distinct threads, and the library reaches an inbox created in
its own process through the pointer itself (HIP does not open an IPC handle in the exporting process), so the whole
multi-rank path runs here with real library code at R = 8: tile-range partition, interface list with an 8-bit reader
mask, the exchange buffer of the streaming protocol, the on-chip kernels of all eight ranks co-resident and exchanging
through per-rank inboxes, the collective fall-back agreement, and the sharded assembly (every rank keeps only the K
rows of its own nodes, one ghost layer and the prescribed nodes).  What it cannot show is the part that needs a node:
RCCL between devices and stores crossing xGMI.

Geometry: bench.py's weak-scaling mesh for --gpus 8 (plate-with-hole stacked 8 times along y) at reduced resolution.
The sum-all-reduce between the threads is done on the host, in rank order, by one thread: the same bits for everybody.
"""
import threading

import numpy as np
import pytest

import oracle
from magnetite_amd import Context, _lib, meshgen

pytestmark = pytest.mark.gpu
R = 8


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def stacked_problem(n=110, copies=R):
    holes = [(0.5, (k + 0.5) / copies, 0.15) for k in range(copies)]
    return meshgen.config_fixed_left_pull_right(
        meshgen.plate_with_holes(n, n * copies, 1.0, float(copies), holes=holes))


class HostAllReduce:
    """sum-all-reduce between R threads: everybody deposits its view, rank 0 adds them in rank order, everybody copies"""

    def __init__(self, ranks):
        self.slots = [None] * ranks
        self.barrier = threading.Barrier(ranks, timeout=180)
        self.total = None

    def __call__(self, rank, arr):
        self.slots[rank] = arr
        self.barrier.wait()
        if rank == 0:
            t = self.slots[0].copy()
            for other in self.slots[1:]:
                t += other
            self.total = t
        self.barrier.wait()
        arr[:] = self.total
        self.barrier.wait()


def run_ranks(prob, inboxes, solves=1, inbox_bytes=1 << 20, **opts):
    comm = HostAllReduce(R)
    sync = threading.Barrier(R, timeout=180)
    handles, results, errors = [None] * R, [None] * R, []

    def worker(rank):
        try:
            with Context(device=0, **opts) as c:
                c.init_callback(lambda a, r=rank: comm(r, a), rank, R)
                if inboxes:
                    handles[rank] = c.create_inbox(inbox_bytes)
                    sync.wait()
                    c.open_inboxes(handles)
                outs = []
                for _ in range(solves):
                    out = c.solve(prob)
                    out.update(comm_info=c.comm_info())
                    outs.append(out)
                results[rank] = outs
                sync.wait()  # nobody frees an inbox another rank's kernel may still touch
        except Exception as exc:  # a failing rank must not leave the others waiting for ever
            errors.append((rank, repr(exc)))
            comm.barrier.abort()
            sync.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(R)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert all(r is not None for r in results)
    return results


@pytest.fixture(scope="module")
def case(built):
    p = stacked_problem()
    ref = oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                     p.part_thickness, path="sparse")
    with Context(device=0) as c:
        single = c.solve(p)
    return p, ref, single


def check(case, results, kernel, exchange=None):
    p, ref, single = case
    if exchange is not None:
        assert all(o["exchange"] == exchange for outs in results for o in outs), [outs[0]["exchange"] for outs in results]
    k = p.u_known == 1
    for rank, outs in enumerate(results):
        for out in outs:
            assert out["converged"] == 1 and out["cg_kernel"] == kernel, (rank, out["cg_kernel"])
            assert out["comm_info"]["ranks"] == R and out["comm_info"]["rank"] == rank
            assert abs(out["iterations"] - ref["iterations"]) <= max(3, ref["iterations"] // 50)
            assert rel(out["u"], ref["u"]) <= 1e-8 and rel(out["u"], single["u"]) <= 1e-9
            assert np.array_equal(out["u"][k], p.u_in[k]) and np.array_equal(out["f"][~k], p.f_in[~k])
            # reactions and stress come from the rank's OWN rows of K (prescribed nodes are kept by every rank) and the
            # gathered displacements
            assert rel(out["f"][k], ref["f"][k]) <= 1e-7 and rel(out["stress"], ref["stress"]) <= 1e-7
            # sharded assembly: a rank holds about an eighth of K plus ghost and boundary rows, never all of it
            assert 0 < out["nnz"] < 0.3 * single["nnz"], (rank, out["nnz"], single["nnz"])
        assert all(np.array_equal(o["u"], outs[0]["u"]) for o in outs[1:])  # repeated solves: same bits
    us = [o["ms_cg"] * 1e3 / max(1, o["iterations"]) for outs in results for o in outs]
    print(f"[{R} ranks, cg_kernel {kernel}] {results[0][0]['iterations']} iterations, {min(us):.2f}-{max(us):.2f} us per "
          f"iteration; single rank: {single['ms_cg'] * 1e3 / single['iterations']:.2f} us (cg_kernel {single['cg_kernel']})")
    for outs in results[1:]:  # every rank returns the same full solution, bit for bit
        assert np.array_equal(outs[0]["u"], results[0][0]["u"]) and np.array_equal(outs[0]["f"], results[0][0]["f"])
    assert sum(outs[0]["nnz"] for outs in results) >= single["nnz"]


def test_eight_ranks_streaming_kernels_one_allreduce_per_iteration(case):
    check(case, run_ranks(case[0], inboxes=False, cg_variant=1, tile_nodes=512), kernel=1, exchange=1)


def test_eight_ranks_streaming_kernels_exchange_through_inboxes(case, monkeypatch):
    """meshes the chips cannot hold stream (BASELINE config 5 on 8 GPUs: 1M nodes per GPU); with the inboxes open the
    per-iteration exchange of the streaming kernels goes through them too (k_stream_exchange) instead of an all-reduce;
    then with every wait cut short: the ranks agree to fall back to the all-reduce and still return the right answer"""
    check(case, run_ranks(case[0], inboxes=True, solves=2, cg_variant=1, tile_nodes=512), kernel=1, exchange=3)
    monkeypatch.setenv("MAG_TUNE_STREAM_SPIN", "0")
    check(case, run_ranks(case[0], inboxes=True, cg_variant=1, tile_nodes=512), kernel=1, exchange=1)


def test_eight_ranks_on_chip_kernels_exchange_through_inboxes(case):
    """22-23 tiles per rank, one workgroup each: the eight persistent launches are co-resident on the 256 CUs and meet
    every iteration through their inboxes; two solves in a row (fresh tags, inboxes cleared between them)"""
    check(case, run_ranks(case[0], inboxes=True, solves=2, cg_variant=2, tile_nodes=512), kernel=2, exchange=2)


def test_eight_ranks_on_chip_edge_blocks(case, monkeypatch):
    """The multi-GPU instantiation of the on-chip kernel with edge blocks in registers (the default across ranks since round
    4; the stacked plates are structured, so every rank finds the mesh eligible): same answers, and the stats say which
    instantiation ran; MAG_TUNE_PERSIST_MG_BLOCKS=0 keeps the triangle walk with cached weights."""
    results = run_ranks(case[0], inboxes=True, solves=2, cg_variant=2, tile_nodes=512)
    check(case, results, kernel=2, exchange=2)
    assert all(o["edge_blocks"] == 1 for outs in results for o in outs)
    monkeypatch.setenv("MAG_TUNE_PERSIST_MG_BLOCKS", "0")
    results = run_ranks(case[0], inboxes=True, cg_variant=2, tile_nodes=512)
    check(case, results, kernel=2, exchange=2)
    assert all(o["edge_blocks"] == 0 and o["cg_kernel"] == 2 for outs in results for o in outs)


def test_eight_ranks_on_chip_edge_blocks_with_overflow_on_an_unstructured_mesh(built, monkeypatch):
    """What the reference's mesher hands `solver::run` (mesher.rs:501-506) is unstructured: across ranks such a mesh now runs
    the edge-block instantiation with overflow records too (edge_blocks == 2; until round 4 it fell to the triangle walk as
    soon as there was a second rank), every rank deciding from the pool limits of ALL ranks' workgroups.  Forced to two and
    three tiles per workgroup as well (sibling slots inside the pool's entries), and with MAG_TUNE_PERSIST_MG_OVERFLOW=0:
    the walk."""
    p = meshgen.config_fixed_left_pull_right(meshgen.shuffle(meshgen.frontal_like(300, 0.4, 11), 4))
    assert p.mesh.num_nodes >= 8 * 6 * 512
    ref = oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                     p.part_thickness, path="sparse", stop_mode=oracle.STOP_REL, tol=1e-9)
    with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-9) as c:
        single = c.solve(p)
        assert c.stats()["edge_blocks"] == 2
    case = (p, ref, single)
    for k, mode in ((None, 2), ("2", 2), ("3", 2), (None, 0)):
        if k:
            monkeypatch.setenv("MAG_TUNE_PERSIST_K", k)
        else:
            monkeypatch.delenv("MAG_TUNE_PERSIST_K", raising=False)
        if mode == 0:
            monkeypatch.setenv("MAG_TUNE_PERSIST_MG_OVERFLOW", "0")
        results = run_ranks(p, inboxes=True, solves=2 if k is None and mode else 1, cg_variant=2, tile_nodes=512,
                            stop_mode=_lib.MAG_STOP_REL, tol=1e-9)
        assert all(o["edge_blocks"] == mode and o["cg_kernel"] == 2 and o["exchange"] == 2 for outs in results for o in outs), \
            (k, mode, [(o["edge_blocks"], o["cg_kernel"]) for outs in results for o in outs])
        for rank, outs in enumerate(results):
            for out in outs:
                assert out["converged"] == 1
                assert abs(out["iterations"] - ref["iterations"]) <= max(3, ref["iterations"] // 50)
                # (a relative stop at 1e-9 leaves two solves that far apart, reactions and stresses at ~1e-8)
                assert rel(out["u"], ref["u"]) <= 1e-8 and rel(out["u"], single["u"]) <= 1e-8
                kn = p.u_known == 1
                assert rel(out["f"][kn], ref["f"][kn]) <= 1e-7 and rel(out["stress"], ref["stress"]) <= 1e-7
            assert all(np.array_equal(o["u"], outs[0]["u"]) for o in outs[1:])
        for outs in results[1:]:
            assert np.array_equal(outs[0]["u"], results[0][0]["u"])


def test_eight_ranks_sharded_ordering_phase_builds_the_same_solve(case, monkeypatch):
    """Across ranks the ordering phase builds incidence lists, halo lists, ring words and tile tables for the tiles a rank
    needs only -- its own, those of its ghost rows and of the prescribed rows -- and derives the interface from one pass over
    the elements (round 4; MAG_TUNE_SHARD_ORDER=0: every rank builds the whole mesh's tables, as before).  The tables of
    a rank's own tiles, the interface list and the decisions the ranks share are the same either way, so both modes return
    the same bits, on-chip and streaming; the stats show the smaller tables."""
    p = case[0]
    for variant in (2, 1):
        sharded = run_ranks(p, inboxes=True, cg_variant=variant, tile_nodes=512)
        monkeypatch.setenv("MAG_TUNE_SHARD_ORDER", "0")
        replicated = run_ranks(p, inboxes=True, cg_variant=variant, tile_nodes=512)
        monkeypatch.delenv("MAG_TUNE_SHARD_ORDER")
        check(case, sharded, kernel=variant, exchange=2 if variant == 2 else 3)
        for (a,), (b,) in zip(sharded, replicated):
            assert a["cg_kernel"] == b["cg_kernel"] == variant and a["edge_blocks"] == b["edge_blocks"]
            assert a["iterations"] == b["iterations"] and a["nnz"] == b["nnz"]
            assert np.array_equal(a["u"], b["u"]) and np.array_equal(a["f"], b["f"]) and np.array_equal(a["stress"], b["stress"])
            assert a["max_tile_halo"] == b["max_tile_halo"]  # (the largest halo of ANY tile: all-reduced in the sharded phase)
            assert a["ell_entries"] < b["ell_entries"] and a["halo_nodes"] < b["halo_nodes"]
        # (the stacked plates are prescribed along both long edges, so every rank needs the tiles along them: about half of
        # the mesh here; on BASELINE config 5's square a rank builds 15 % of the tables, scripts/order_phase_ranks.py)
        assert sum(a["ell_entries"] for (a,) in sharded) < 0.7 * sum(b["ell_entries"] for (b,) in replicated)


def test_eight_ranks_agree_to_fall_back(case, monkeypatch):
    """MAG_TUNE_PERSIST_SPIN=0: every wait of the on-chip kernels gives up at once; the ranks agree on it through one
    all-reduce of their failure flags and all redo the solve with the streaming kernels -- and say so in their stats"""
    monkeypatch.setenv("MAG_TUNE_PERSIST_SPIN", "0")
    results = run_ranks(case[0], inboxes=True, cg_variant=2, tile_nodes=512)
    check(case, results, kernel=1)
    assert all(outs[0]["persist_timeout"] == 1 for outs in results)


def test_eight_ranks_fp32_leg_of_config5(case):
    """BASELINE config 5 asks for fp64 vs fp32 CG across 8 GPUs: the fp32 leg (mag_options.precision = 1) runs the same
    streaming protocol -- one all-reduce per iteration of [dot partials | interface q], kept in doubles -- and follows the
    single-rank fp32 solve iteration for iteration; like it, it stalls at fp32 accuracy (no 1e-8 parity claim)."""
    p, ref, _ = case
    opts = dict(precision=1, stop_mode=_lib.MAG_STOP_REL, tol=1e-7, tile_nodes=512)
    with Context(device=0, **opts) as c:
        one = c.solve(p)
    results = run_ranks(p, inboxes=False, **opts)
    for rank, outs in enumerate(results):
        out = outs[0]
        assert out["converged"] == 1 and out["cg_kernel"] == 4, (rank, out["cg_kernel"])
        assert abs(out["iterations"] - one["iterations"]) <= max(3, one["iterations"] // 100)
        assert rel(out["u"], one["u"]) <= 2e-5          # two fp32 solves, different summation orders
        assert 1e-9 < rel(out["u"], ref["u"]) < 5e-4     # fp32 accuracy, as on one GPU
        assert 0 < out["nnz"] < 0.3 * one["nnz"]
        assert np.array_equal(out["u"], results[0][0]["u"])


@pytest.mark.parametrize("inboxes,variant,exchange", [(False, 1, 1), (True, 1, 3), (True, 2, 2)])
def test_eight_ranks_iteration_cap_returns_best_param(case, inboxes, variant, exchange):
    """solver.rs:149-176 across ranks: at the iteration cap every rank returns argmin's best_param -- the lowest-cost
    iterate, recovered by repeating the solve up to that iteration.  The repeat must take the path of the first pass
    (same kernel, same exchange) and land on the recorded best cost bit for bit: best_param_mismatch == 0."""
    p, _, _ = case
    cap = 60
    ref = oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                     p.part_thickness, path="sparse", max_iter=cap, hist_len=cap)
    kbest = int(np.argmin(ref["history"])) + 1
    assert ref["iterations"] == cap and kbest < cap  # the cap falls into a rising stretch of the residual history
    results = run_ranks(p, inboxes=inboxes, cg_variant=variant, tile_nodes=512, max_iter=cap)
    for rank, (out,) in enumerate(results):
        assert out["converged"] == 0 and out["termination"] == _lib.MAG_TERM_MAX_ITERS and out["iterations"] == cap
        assert out["cg_kernel"] == variant and out["exchange"] == exchange, (rank, out["cg_kernel"], out["exchange"])
        assert out["best_iteration"] == kbest and out["best_param_mismatch"] == 0, (rank, out["best_iteration"], kbest)
        assert abs(out["final_cost"] - ref["final_cost"]) <= 1e-9 * ref["final_cost"]
        assert rel(out["u"], ref["u"]) <= 1e-8
        assert np.array_equal(out["u"], results[0][0]["u"])


def test_config5_eight_ranks_at_baseline_size_against_the_oracle_fixture(built):
    """BASELINE config 5 as written: the 16M-triangle multi-hole mesh split over EIGHT ranks (1M nodes per rank: beyond the
    on-chip kernel, so the streaming kernels trade through the inboxes, k_stream_exchange), relative stop 1e-8; every
    rank's returned solution against the oracle's sampled solution (tests/golden/fullsize_multihole16m.npz: iteration
    count, u at 4096 DOFs <= 1e-8, reactions and stress <= 1e-7)."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_multihole16m.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not committed")
    fx = np.load(path, allow_pickle=False)
    p = meshgen.baseline_problem("multihole16m")
    N, E = p.mesh.num_nodes, p.mesh.num_elements
    assert (N, E) == (int(fx["num_nodes"]), int(fx["num_elements"]))
    assert float(np.sum(p.xy_flat * np.arange(1, 2 * N + 1) % 7.0)) == float(fx["xy_checksum"])
    results = run_ranks(p, inboxes=True, inbox_bytes=32 << 20, cg_variant=1, stop_mode=_lib.MAG_STOP_REL,
                        tol=float(fx["rel_tol"]))
    iu, ie = fx["dof_idx"], fx["elem_idx"]
    slack = 0 if "1 thread" in str(fx["solver"]) else max(2, int(fx["iterations"]) // 1000)
    known = p.u_known == 1
    for rank, (out,) in enumerate(results):
        assert out["converged"] == 1 and out["cg_kernel"] == 1 and out["exchange"] == 3, (rank, out["exchange"])
        assert out["comm_info"]["ranks"] == R and out["comm_info"]["rank"] == rank
        assert abs(int(out["iterations"]) - int(fx["iterations"])) <= slack, (rank, out["iterations"])
        assert rel(out["u"][iu], fx["u_at"]) <= 1e-8, rank
        assert abs(np.linalg.norm(out["u"]) - float(fx["u_norm"])) <= 1e-8 * float(fx["u_norm"])
        assert np.array_equal(out["u"][known], p.u_in[known])
        assert np.abs(out["f"][iu] - fx["f_at"]).max() <= 1e-7 * float(fx["f_known_norm"]), rank
        assert rel(out["stress"][ie], fx["stress_at"]) <= 1e-7, rank
        assert np.array_equal(out["u"], results[0][0]["u"]) and np.array_equal(out["f"], results[0][0]["f"])
        if rank:  # 8 x 0.4 GB of results: keep rank 0's only
            out["u"] = out["f"] = out["stress"] = None
    us = [o[0]["ms_cg"] * 1e3 / o[0]["iterations"] for o in results]
    print(f"[config 5, {R} ranks sharing one GPU] {results[0][0]['iterations']} iterations, {min(us):.1f}-{max(us):.1f} us "
          f"per iteration")


def test_full_matrix_is_still_available_on_a_multi_rank_context(case):
    """mag_assemble_csr on a rank of a multi-rank communicator hands out ALL of K (the run itself kept only the rank's
    rows): bit-identical to the oracle's"""
    p = case[0]
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    comm = HostAllReduce(1)
    with Context(device=0, tile_nodes=512) as c:
        c.init_callback(lambda a: comm(0, a), 3, R)  # rank 3 of 8; nothing below needs a collective
        c.upload_problem(p)
        rowptr, col, val = c.assemble_csr()
    assert np.array_equal(rowptr.astype(np.int64), K.rowptr) and np.array_equal(col, K.col) and np.array_equal(val, K.val)
