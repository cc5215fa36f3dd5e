"""Soak of the default configuration: N solves of the 1M-triangle benchmark mesh in one context; every solve must take
the on-chip kernel, the same iteration count and return the same bits (bounded waits never hit on an idle GPU)."""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from magnetite_amd import Context, _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
prob, _ = bench.build_problem("hole1m", 1)
digests, its, kern, tmo, t0 = set(), set(), set(), 0, time.time()
with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8) as c:
    c.upload_problem(prob)
    for k in range(n):
        c.run()
        st = c.stats()
        its.add(int(st["iterations"]))
        kern.add(int(st["cg_kernel"]))
        tmo += int(st["persist_timeout"])
        digests.add(hashlib.sha256(c.download()[0].tobytes()).hexdigest()[:12])
        if k % 50 == 49:
            print(f"{k + 1} solves, {time.time() - t0:.0f} s", flush=True)
print(f"{n} solves: iterations {sorted(its)}, kernels {sorted(kern)}, timeouts {tmo}, distinct results {len(digests)}")
sys.exit(0 if (len(digests) == 1 and kern == {2} and tmo == 0 and len(its) == 1) else 1)
