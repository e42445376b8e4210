"""Iteration rate over a long run at a given sample size (does the MFMA pre-filter keep its speed once the layout
has grown outliers beyond its f16 range?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from graphem_rapids_amd import _native
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n, D, k, _, edges, pos = bench.make_workload("rr1m")
eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=1)
eng.set_positions(pos)
for b in range(6):
    t0 = time.perf_counter(); eng.run(100); eng.sync(); dt = time.perf_counter() - t0
    p = eng.get_positions()
    print(f"S={S} block {b}: {1e6 * dt / 100:.0f} us/iter, max|pos| {np.abs(p).max():.1f}, rows beyond 128: {(np.abs(p).max(axis=1) > 128).sum()}", flush=True)
eng.close()
