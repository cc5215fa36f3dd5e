#!/bin/bash
# usage: scripts/tune_op.sh "<env>" <tile>: operator-only timing after a short run
env $1 python3 - "$1" "$2" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import bench
from magnetite_amd import Context, _lib
prob,_ = bench.build_problem(os.environ.get("WL","hole1m"), 1)
with Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8, max_iter=200, tile_nodes=int(sys.argv[2])) as c:
    c.upload_problem(prob); c.run(allow_not_converged=True)
    st=c.stats()
    ms=min(c.time_operator(300) for _ in range(3))
    print(sys.argv[1].ljust(52), "tile", sys.argv[2], "us/launch %.2f  iter_us %.2f"%(ms*1e3, st['ms_cg']*1e3/st['iterations']))
PY
