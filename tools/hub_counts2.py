"""Candidate counts of the KNN phase on the 100 K-vertex graph with hubs of degree 20000 / 2000 (tools/hub_probe.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from hub_probe import hub_graph
from graphem_rapids_amd import _native
n, D, k = 100000, 3, 10
hubs = [(17, 20000), (4021, 2000)]
edges = hub_graph(n, 8, hubs)
pos = np.random.default_rng(1).standard_normal((n, D)).astype(np.float32)
eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, 256)
eng.set_positions(pos)
for it in range(12):
    eng.run(1); eng.sync()
    sub, fin, ovf = eng.knn_last_counts()
    p = eng.get_positions()
    print(f"iter {it}: candidates mean {fin.mean():.0f} median {np.median(fin):.0f} p99 {np.quantile(fin, 0.99):.0f} max {fin.max()} over-cap {(fin > 8192).sum()}; "
          f"max|pos| {np.abs(p).max():.1f}, hub at {np.round(p[17], 1)}", flush=True)
