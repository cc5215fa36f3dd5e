"""How does the average launch time of the CG kernel depend on how many launches run back to back?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from magnetite_amd import Context, _lib
prob, _ = bench.build_problem(os.environ.get("WL", "hole1m"), 1)
with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8, max_iter=100, tile_nodes=int(os.environ.get("TILE", "512"))) as c:
    c.upload_problem(prob); c.run(allow_not_converged=True)
    for reps in (1, 2, 4, 10, 50, 400, 2000):
        time.sleep(0.2)
        ms = [c.time_operator(reps) for _ in range(3)]
        print("reps %5d  us/launch %s" % (reps, " ".join("%.2f" % (m * 1e3) for m in ms)), flush=True)
