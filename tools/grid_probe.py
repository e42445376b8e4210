"""Diagnostics of the grid KNN on a bench workload's state after a few iterations."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from graphem_rapids_amd import _native
wl = sys.argv[1] if len(sys.argv) > 1 else "rr1m"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n, D, k, _, edges, pos = bench.make_workload(wl)
a = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, knn_method="scan")
a.set_positions(pos)
a.run(6)
state = a.get_positions()
rng = np.random.default_rng(0)
sampled = rng.permutation(len(edges))[:S].astype(np.int32)
ref = a.knn_midpoints(sampled)
a.close()
print("state: std", state.std(0), "abs quantiles 50/90/99/99.9/max", np.quantile(np.abs(state), [0.5, 0.9, 0.99, 0.999, 1.0]))
mid = (state[edges[sampled, 0]] + state[edges[sampled, 1]]) / 2
print("query |q| quantiles", np.quantile(np.linalg.norm(mid, axis=1), [0.25, 0.5, 0.75, 0.9, 0.99, 1.0]))
g = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, knn_method="grid")
g.set_positions(state)
out = g.knn_midpoints(sampled)
sub, fin, ovf = g.knn_last_counts()
print("identical", np.array_equal(out, ref), "overflowed queries", int(ovf.sum()), "final list lengths: quantiles",
      np.quantile(fin, [0, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0]))
bad = np.nonzero(ovf)[0][:10]
print("overflowed: |q| =", np.linalg.norm(mid[bad], axis=1), "counts", fin[bad])
g.close()
