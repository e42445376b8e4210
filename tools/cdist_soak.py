"""Long parity-mode runs: `iters` iterations from the reference's start, every iteration's counters checked -- rows listed,
rows left unresolved (must stay 0), candidate lists that overflowed or had to be searched exhaustively -- and the layout kept
finite.  python tools/cdist_soak.py [workload] [iters]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from graphem_rapids_amd import _native

name = sys.argv[1] if len(sys.argv) > 1 else "rr1m"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n, D, k, S, edges, pos = bench.make_workload(name)
eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=0, knn_distance="cdist")
eng.set_positions(pos)
listed = unres = ovf1 = ovf2 = 0
worst = 0
t0 = time.time()
for t in range(iters):
    eng.step(None)
    full, u = eng.knn_cdist_stats()
    _, _, ovf = eng.knn_last_counts()
    listed += full; unres += u; worst = max(worst, full)
    ovf1 += int((ovf == 1).sum()); ovf2 += int((ovf >= 2).sum())
p = eng.get_positions().astype(np.float64)   # (a float32 std of 1 M values with a few far-flung vertices is off by 2e-3)
print(f"{name}: {iters} iterations in parity mode, {time.time() - t0:.1f} s: listed rows {listed} (at most {worst} in one iteration), "
      f"unresolved {unres}, exhaustive searches {ovf1}, other overflow marks {ovf2}; positions finite: {bool(np.isfinite(p).all())}, "
      f"unbiased std per column {np.round(p.std(axis=0, ddof=1), 7).tolist()}")
sys.exit(0 if unres == 0 and np.isfinite(p).all() else 1)
