"""Candidate counts of the KNN phase on a graph with one 100000-degree hub."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from hub_probe import hub_graph
from graphem_rapids_amd import _native
n, D, k = 1000000, 3, 10
edges = hub_graph(n, 8, [(5, 100000)])
pos = np.random.default_rng(1).standard_normal((n, D)).astype(np.float32)
eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, 256)
eng.set_positions(pos)
for it in range(4):
    eng.run(1); eng.sync()
    sub, fin, ovf = eng.knn_last_counts()
    p = eng.get_positions()
    print(f"iter {it}: candidates mean {fin.mean():.0f} median {np.median(fin):.0f} max {fin.max()} overflowed {int(ovf.sum())}; max|pos| {np.abs(p).max():.1f}, hub at {p[5]}", flush=True)
