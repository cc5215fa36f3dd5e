"""Wall time of the first solve in a fresh context (what a one-shot caller such as the Rust CLI pays) vs later solves."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from magnetite_amd import Context, _lib
prob, desc = bench.build_problem(sys.argv[1] if len(sys.argv) > 1 else "hole1m", 1)
t0 = time.perf_counter()
c = Context(device=0, stop_mode=_lib.MAG_STOP_REL, tol=1e-8)
t1 = time.perf_counter()
for i in range(3):
    ta = time.perf_counter(); c.upload_problem(prob); tb = time.perf_counter(); c.run(); tc = time.perf_counter(); c.download(); td = time.perf_counter()
    st = c.stats()
    print(f"solve {i}: upload {1e3*(tb-ta):.1f} ms run {1e3*(tc-tb):.1f} ms (device total {st['ms_total']:.1f}, cg {st['ms_cg']:.1f}, order {st['ms_order']:.2f}, csr {st['ms_csr_symbolic']:.2f}) download {1e3*(td-tc):.1f} ms", flush=True)
print(f"context creation {1e3*(t1-t0):.1f} ms")
