"""-m gpu: the multi-rank CG path, rehearsed on ONE GPU (see tests/dist_worker.py for what each mode covers)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def launch(nproc, mode, port, variant=1, extra=(), env_extra=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER, "--mode", mode, "--variant", str(variant),
           *extra]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("nproc,variant", [(2, 1), (3, 1), (2, 0)])
def test_ranks_sharing_one_gpu_through_gloo(built, nproc, variant):
    r = launch(nproc, "callback", 29540 + nproc + 10 * variant, variant)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def test_four_ranks_on_the_weak_scaling_geometry(built):
    """bench.py --gpus 4's mesh (plate-with-hole stacked 4x along y) at reduced resolution, default tile size,
    default CG variant: 4 ranks share the one GPU, collectives through gloo."""
    r = launch(4, "callback", 29571, 1, ("--tile", "512", "--stacked", "4", "--mesh", "150"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def test_on_chip_cg_across_ranks_falls_back_together(built):
    """MAG_TUNE_PERSIST_SPIN=0: every wait of the multi-GPU on-chip kernel gives up at once.  The ranks must agree on
    it (one all-reduce of the failure flags), redo the solve with the streaming kernels + one all-reduce per iteration
    and still return the right answer, identically, on every rank."""
    r = launch(3, "callback", 29597, 1, ("--window", "2", "--tile", "512", "--mesh", "120", "--expect-kernel", "1"),
               {"MAG_TUNE_PERSIST_MIN_K": "1", "MAG_TUNE_PERSIST_SPIN": "0"})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("nproc", [2, 3, 4])
def test_on_chip_cg_across_ranks_through_device_inboxes(built, nproc):
    """The same protocol with one inbox per rank in DEVICE memory, mapped by the other ranks through HIP IPC: a rank
    polls only its own inbox, writers store into the inboxes of the ranks that read a value.  (Here the ranks share
    one GPU, so the IPC mappings are same-device; on a node the stores cross xGMI.)"""
    r = launch(nproc, "callback", 29587 + nproc, 1, ("--window", "2", "--tile", "512", "--mesh", "120"),
               {"MAG_TUNE_PERSIST_MIN_K": "1"})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("nproc", [2, 4])
def test_streaming_kernels_exchange_through_device_inboxes(built, nproc):
    """k_stream_exchange with one PROCESS per rank: the inboxes are real HIP IPC mappings (same device here); streaming
    kernels, one exchange launch per iteration instead of an all-reduce; two solves in a row give the same bits."""
    r = launch(nproc, "callback", 29601 + nproc, 1,
               ("--window", "2", "--stream-inbox", "--tile", "512", "--mesh", "120", "--expect-kernel", "1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("nproc", [2, 3])
def test_on_chip_cg_across_ranks_through_a_host_window(built, nproc):
    """Every rank runs its share of the mesh as one persistent launch; per iteration the ranks exchange one record of
    sums and the q of the interface nodes through a window of shared host memory (tagged granules, system scope).  On
    this one-GPU box the ranks' kernels run side by side on the same GPU; on a node each rank has its own."""
    r = launch(nproc, "callback", 29577 + nproc, 1, ("--window", "1", "--tile", "512", "--mesh", "120"),
               {"MAG_TUNE_PERSIST_MIN_K": "1"})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def test_two_ranks_several_tiles_per_workgroup(built):
    """MAG_TUNE_GRID=4: every workgroup of the COMM kernel walks several tiles, ghost loop strided over few workgroups"""
    r = launch(2, "callback", 29576, 1, ("--precond", "2"), {"MAG_TUNE_GRID": "4"})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def test_three_ranks_with_the_opt_in_preconditioner(built):
    """block-Jacobi PCG across ranks: five sums in the exchange buffer, ghost nodes apply Minv locally"""
    r = launch(3, "callback", 29575, 1, ("--precond", "2"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("variant", [1, 0])
def test_rccl_single_rank_communicator(built, variant):
    r = subprocess.run([sys.executable, WORKER, "--mode", "rccl1", "--variant", str(variant)], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
