"""Iteration time across parameter ranges (sample size, neighbours, dimension) on the 1M-vertex graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from graphem_rapids_amd import _native
n, D0, k0, S0, edges, pos0 = bench.make_workload("rr1m")
cases = [(3, 10, 256), (3, 15, 512), (3, 10, 1024), (3, 10, 4096), (3, 32, 256), (3, 63, 256), (3, 100, 256), (2, 10, 256), (4, 10, 256), (8, 10, 256), (16, 10, 256)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
for D, k, S in cases:
    pos = (np.random.default_rng(0).standard_normal((n, D)) * 0.1).astype(np.float32)
    eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S)
    eng.set_positions(pos)
    eng.run(3); eng.sync()
    t0 = time.perf_counter(); eng.run(20); eng.sync(); dt = (time.perf_counter() - t0) / 20
    eng.timing_enable(True); eng.timing_reset(); eng.run(5); eng.sync()
    tm = {a: round(1e3 * b[0] / b[1], 1) for a, b in eng.timings().items()}
    print(f"D={D} k={k} S={S}: {1e6 * dt:.0f} us/iter", tm, flush=True)
    eng.close()
