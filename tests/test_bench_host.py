"""CPU: the host-side logic of bench.py that needs no GPU -- the bare `--gpus N` launch (the parent starts its ranks as a
fresh child under torch.distributed.run before importing torch), the PMC staleness flag, the usable-core count."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_bare_multi_gpu_launch_spawns_its_ranks_and_relays_their_exit_code(built):
    """`python bench.py --gpus 2` with no RANK/WORLD_SIZE in the environment: the parent runs the two ranks under
    torch.distributed.run as a child.  There is no GPU here, so every rank stops at bench.py's own "needs a HIP device"
    exit -- reached only INSIDE a rank (RANK set by the launcher), and the parent returns the launcher's non-zero code."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--workload",
                        "plate100k", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_entry_points_gpu.py")
    assert r.returncode != 0
    assert "bench.py needs a HIP device" in r.stderr, r.stderr[-2000:]
    assert "but WORLD_SIZE=1" not in r.stderr


def test_spawn_command_line(monkeypatch):
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert bench.spawn_ranks(4) == 7
    cmd = seen["cmd"]
    assert cmd[:4] == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_pmc_counters_are_marked_stale_when_the_kernels_changed(built, tmp_path, monkeypatch):
    """profiles/pmc_summary.json records the digest of the kernel sources it was collected on; the library records its
    own next to the .so.  A mismatch flags the traffic and drops the counted utilisations."""
    from magnetite_amd import _lib
    assert _lib.built_source_hash() == _lib.kernel_source_hash()  # the library in the tree is built from the tree
    summary = json.load(open(bench.PMC_SUMMARY))
    fake = tmp_path / "pmc_summary.json"
    monkeypatch.setattr(bench, "PMC_SUMMARY", str(fake))
    E, N = 1001352, 502584
    for recorded, stale in ((_lib.built_source_hash(), False), ("0123456789abcdef", True), (None, True)):
        summary["_meta"] = {"source_hash": recorded}
        fake.write_text(json.dumps(summary))
        assert bench.pmc_is_stale() is stale
        line = bench.roofline_onchip(E, N, 5389, 57.0, bench.load_pmc("hole1m:tile512:kernel2"), 512)
        assert line["traffic"] > 0 and line["traffic_stale"] is stale
        assert (line["counted"] is None) is stale
        k = bench.kernel_line("k", 1e9, "f", 0.2, bench.load_pmc("multihole16m:tile512:spmv"))
        assert k["traffic_stale"] is stale and k["traffic"] > 0


def test_usable_cores_is_within_the_host():
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_size_sweep_workloads():
    """`--workload plate:<cells per side>` / `frontal:<pitch>`: the size-sweep meshes of profiles/r04_persist_ab.txt (one,
    two and three tiles per workgroup of the on-chip kernel); single-GPU workloads, like frontal1m."""
    p, desc = bench.build_problem("plate:24", 1)
    assert p.mesh.num_nodes == 25 * 25 and p.mesh.num_elements == 2 * 24 * 24 and desc == "plate 24"
    q, desc = bench.build_problem("frontal:12", 1)
    assert q.mesh.num_elements > 200 and desc == "frontal 12"
    assert (q.u_known == 1).any() and (p.u_known == 1).any()
    with pytest.raises(SystemExit):
        bench.build_problem("plate:24", 2)
    with pytest.raises(SystemExit):
        bench.build_problem("no-such-workload", 1)
