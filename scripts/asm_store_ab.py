import json, os, subprocess, sys
ROOT='/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd()
WORKER = r"""
import json, sys
sys.path.insert(0, %r)
import bench
from magnetite_amd import Context, _lib
prob, desc = bench.build_problem(sys.argv[1], 1)
with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8, max_iter=64) as c:
    c.upload_problem(prob)
    rows=[]
    for _ in range(8):
        c.run()
        st = c.stats()
        rows.append((st["ms_assemble"], st["ms_bc"], st["ms_assemble"]+st["ms_bc"]))
rows.sort(key=lambda r:r[2])
print(json.dumps({"asm_bc_best3": [[round(x,4) for x in r] for r in rows[:3]]}))
""" % ROOT
for wl in sys.argv[2:]:
    for lib in sys.argv[1].split(','):
        env = dict(os.environ, MAG_LIB_PATH=os.path.abspath(lib))
        r = subprocess.run([sys.executable, "-c", WORKER, wl], env=env, capture_output=True, text=True, timeout=600)
        print(wl, os.path.basename(lib), [l for l in r.stdout.splitlines() if l.startswith("{")][-1] if r.returncode==0 else r.stderr[-300:], flush=True)
