import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W=r"""
import hashlib, json, sys, os
sys.path.insert(0, %r)
import bench
import numpy as np
from magnetite_amd import Context, _lib
prob, desc = bench.build_problem(sys.argv[1], 1)
with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8, cg_variant=2, history_len=6000) as c:
    c.upload_problem(prob); c.run(); st=c.stats(); h=c.history(st["iterations"]); u=c.download()[0]
print(json.dumps({"it": st["iterations"], "final_cost": repr(st["final_cost"]), "hist": hashlib.sha256(h.tobytes()).hexdigest()[:12], "u": hashlib.sha256(u.tobytes()).hexdigest()[:12], "umax": float(abs(u).max())}))
np.save("/tmp/u_"+os.path.basename(os.environ.get("MAG_LIB_PATH","lib"))+".npy", u)
""" % ROOT
for lib in sys.argv[2:]:
    r=subprocess.run([sys.executable,"-c",W,sys.argv[1]],env=dict(os.environ,MAG_LIB_PATH=os.path.abspath(lib)),capture_output=True,text=True)
    print(os.path.basename(lib), r.stdout.strip().splitlines()[-1] if r.returncode==0 else r.stderr[-300:])
import numpy as np
a=np.load("/tmp/u_"+os.path.basename(sys.argv[2])+".npy"); b=np.load("/tmp/u_"+os.path.basename(sys.argv[3])+".npy")
print("rel diff u", float(np.linalg.norm(a-b)/np.linalg.norm(b)), "max abs", float(abs(a-b).max()), "n diff", int((a!=b).sum()), a.size)
