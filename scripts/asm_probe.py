"""Phase times of the non-CG part of a step (ordering, CSR pattern, numeric assembly, BC) for the assembly kernels:
default = k_assemble_ctile (fed from the CG tiles), MAG_TUNE_ASSEMBLY=tiles = round 2's k_assemble_tiles.
    python scripts/asm_probe.py [workload ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import json, sys
sys.path.insert(0, %r)
import bench
from magnetite_amd import Context, _lib
prob, desc = bench.build_problem(sys.argv[1], 1)
with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8, max_iter=64) as c:
    c.upload_problem(prob)
    best = None
    for _ in range(6):
        c.run()
        st = c.stats()
        if best is None or st["ms_assemble"] < best["ms_assemble"]:
            best = st
print(json.dumps({k: best[k] for k in ("ms_order", "ms_csr_symbolic", "ms_assemble", "ms_bc", "nnz", "num_tiles")}))
""" % ROOT
for wl in (sys.argv[1:] or ["hole1m", "plate4m", "plate100k"]):
    for how in ("ctile", "tiles"):
        env = dict(os.environ, MAG_TUNE_ASSEMBLY=how)
        r = subprocess.run([sys.executable, "-c", WORKER, wl], env=env, capture_output=True, text=True, timeout=900)
        if r.returncode:
            print(wl, how, "FAILED", r.stderr[-800:])
            continue
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        d.update(workload=wl, assembly=how)
        print(json.dumps(d), flush=True)
