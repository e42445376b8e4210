"""How many rows of a knn_distance='cdist' run need the pass over all edges, iteration by iteration
(gh_knn_cdist_stats), and what the per-iteration kernels cost.  python tools/cdist_probe.py [workload] [iters]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from graphem_rapids_amd import _native

name = sys.argv[1] if len(sys.argv) > 1 else "rr1m"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n, D, k, S, edges, pos = bench.make_workload(name)
eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=0, knn_distance="cdist")
eng.set_positions(pos)
rows = []
for t in range(iters):
    eng.step(None)
    full, unres = eng.knn_cdist_stats()
    _, _, ovf = eng.knn_last_counts()
    rows.append((full, int((ovf == 1).sum()), int((ovf >= 2).sum())))   # (ovf: 0 once the next set-up has run ahead)
print(name, "per iteration (full-pass rows, of which ties only, of which list not provably complete):")
print(" ".join(f"{a}/{b}/{c}" for a, b, c in rows))
eng.timing_enable(True); eng.timing_reset(); eng.run(20); eng.sync()
print({kname: round(1e3 * tot / cnt, 1) for kname, (tot, cnt) in eng.timings().items()})
