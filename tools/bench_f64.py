"""Iteration time of the float64 engine (csrc/f64.hip: plain one-kernel-per-phase code in double; the KNN through a filter from 131072 edges on) on bench
workloads -- a correctness feature, timed once per round for the record.  python tools/bench_f64.py [workload ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from graphem_rapids_amd import _native

for name in (sys.argv[1:] or ["rr100k", "rr1m"]):
    n, D, k, S, edges, pos = bench.make_workload(name)
    eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=0, dtype="float64")
    eng.set_positions(pos.astype("float64"))
    eng.run(3); eng.sync()
    iters = 10
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); eng.run(iters); eng.sync(); ts.append((time.perf_counter() - t0) / iters * 1e6)
    print(f"{name} float64 engine: us per iteration, 3 passes of {iters}:", " ".join("%.0f" % x for x in ts), flush=True)
    eng.close()
