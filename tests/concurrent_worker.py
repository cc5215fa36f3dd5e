"""Worker for test_two_processes_share_the_gpu: solves the 1M-triangle benchmark mesh a few times with the default
options (on-chip CG when every workgroup can be resident) and prints what ran and a digest of the result."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from magnetite_amd import Context, _lib, meshgen  # noqa: E402

p = meshgen.baseline_problem("hole1m")
out = []
with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8) as c:
    c.upload_problem(p)
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        c.run()
        st = c.stats()
        u = c.download()[0]
        out.append({"kernel": int(st["cg_kernel"]), "iterations": int(st["iterations"]),
                    "converged": int(st["converged"]), "digest": hashlib.sha1(u.tobytes()).hexdigest()})
print(json.dumps(out), flush=True)
