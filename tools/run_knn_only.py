"""Runs the KNN phase (or whole steps) a few times on a bench workload; for rocprofv3 counter passes."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from graphem_rapids_amd import _native
wl = sys.argv[1] if len(sys.argv) > 1 else "rr1m"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = sys.argv[3] if len(sys.argv) > 3 else "knn"
n, D, k, S, edges, pos = bench.make_workload(wl)
if "--dim" in sys.argv:
    D = int(sys.argv[sys.argv.index("--dim") + 1])
    pos = (np.random.default_rng(0).standard_normal((n, D)) * 0.1).astype(np.float32)
eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S)
eng.set_positions(pos)
eng.run(3)
rng = np.random.default_rng(0)
if mode == "run":   # the loop as bench.py runs it (device sampler, set-up inside the normalise launches)
    eng.run(reps)
    reps = 0
for it in range(reps):
    sampled = rng.permutation(len(edges))[:S].astype(np.int32)
    if mode == "knn":
        eng.knn_midpoints(sampled)
    else:
        eng.step(sampled)
eng.sync()
print("done")
