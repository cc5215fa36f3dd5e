"""-m gpu: the eight-rank rehearsal of tests/multirank_threads_impl.py (eight ranks as eight threads of ONE process on
the one GPU), run in a process of its own: eight persistent kernels must be co-resident, and HIP multiplexes a process's
streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) -- two persistent launches sharing a queue would wait
for each other for ever (they would time out and fall back, by design, but that is not what is to be tested).  The
variable is read when HIP initialises, hence the fresh process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eight_ranks_as_threads_of_one_process(built):
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "multirank_threads_impl.py"), "-m",
                        "gpu", "-q", "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=1100)
    assert r.returncode == 0, r.stdout[-6000:] + r.stderr[-2000:]
    assert "13 passed" in r.stdout, r.stdout[-2000:]
