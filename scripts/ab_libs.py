"""A/B of several builds of the library in ONE GPU session (boxes differ by a few per cent): every library named on the
command line (loaded through MAG_LIB_PATH, one process each, interleaved over `--rounds` passes) solves the benchmark
workload; prints us per CG iteration, iteration count and a digest of u per library.

    python scripts/ab_libs.py [--workload hole1m] [--rounds 2] [--variant 2] lib_a.so lib_b.so ...
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import hashlib, json, sys
sys.path.insert(0, %r)
import bench
from magnetite_amd import Context, _lib
prob, desc = bench.build_problem(sys.argv[1], 1)
with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8, cg_variant=int(sys.argv[2])) as c:
    c.upload_problem(prob)
    c.run()
    ts = []
    for _ in range(3):
        c.run()
        st = c.stats()
        ts.append(st["ms_cg"] * 1e3 / max(1, st["iterations"]))
    u = c.download()[0]
print(json.dumps({"us_per_iteration": min(ts), "all": ts, "iterations": st["iterations"], "cg_kernel": st["cg_kernel"],
                  "ms_total": st["ms_total"], "ms_assemble": st["ms_assemble"], "ms_order": st["ms_order"],
                  "digest": hashlib.sha256(u.tobytes()).hexdigest()[:12]}))
""" % ROOT

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="hole1m")
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--variant", type=int, default=2)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
res = {l: [] for l in a.libs}
for _ in range(a.rounds):
    for lib in a.libs:
        env = dict(os.environ, MAG_LIB_PATH=os.path.abspath(lib))
        r = subprocess.run([sys.executable, "-c", WORKER, a.workload, str(a.variant)], env=env, capture_output=True,
                           text=True, timeout=900)
        if r.returncode != 0:
            print(lib, "FAILED", r.stderr[-500:], flush=True)
            continue
        res[lib].append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
for lib, rs in res.items():
    if rs:
        print(json.dumps({"lib": os.path.basename(lib), "us_per_iteration": [round(x["us_per_iteration"], 3) for x in rs],
                          "iterations": rs[0]["iterations"], "cg_kernel": rs[0]["cg_kernel"], "digest": rs[0]["digest"],
                          "ms_assemble": round(rs[0]["ms_assemble"], 4)}), flush=True)
