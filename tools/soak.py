"""Soak run: many iterations of a bench workload; positions must stay finite and unit-std, the overflow
counters are reported, and the iteration rate is printed every block."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from graphem_rapids_amd import _native
wl = sys.argv[1] if len(sys.argv) > 1 else "rr1m"
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 10
per = int(sys.argv[3]) if len(sys.argv) > 3 else 300
n, D, k, S, edges, pos = bench.make_workload(wl)
eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=1)
eng.set_positions(pos)
for b in range(blocks):
    t0 = time.perf_counter(); eng.run(per); eng.sync(); dt = time.perf_counter() - t0
    p = eng.get_positions()
    sub, fin, ovf = eng.knn_last_counts()
    assert np.isfinite(p).all(), "non-finite positions"
    sd = p.astype(np.float64).std(axis=0, ddof=1)
    assert np.abs(sd - 1).max() < 1e-3 and np.abs(p.astype(np.float64).mean(axis=0)).max() < 1e-3, (sd, p.mean(axis=0))
    print(f"{wl} block {b}: {per / dt:.0f} it/s, max|pos| {np.abs(p).max():.2f}, candidates/query mean {fin.mean():.0f} max {fin.max()}, overflowed {int(ovf.sum())}", flush=True)
eng.close()
print("soak ok")
