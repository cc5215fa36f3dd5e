"""CPU, gloo, world_size 2 and 3: the multi-GPU CG protocol (tests/dist_protocol.py) against the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["", "--fused", "--onchip"], ids=["two-collectives", "fused-exchange-buffer", "on-chip-inboxes"])
@pytest.mark.parametrize("world", [2, 3])
def test_distributed_cg_protocol_over_gloo(built, world, mode):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29560 + world + 10 * len(mode)),
           os.path.join(ROOT, "tests", "dist_protocol.py")] + ([mode] if mode else [])
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
