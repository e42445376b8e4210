"""Iteration time on small graphs (per-query KNN kernels, everything latency-bound)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import graphem_rapids_amd as gra
from graphem_rapids_amd import _native
for name, n, p, D, k in [("C1: ER n=1000 p=0.01", 1000, 0.01, 3, 10), ("ER n=5000 p=0.002", 5000, 0.002, 3, 10), ("ER n=20000 p=0.0005", 20000, 0.0005, 3, 10),
                         ("ER n=4039 p=0.0108 D=16 k=32", 4039, 0.0108, 16, 32)]:
    edges = gra.erdos_renyi_edges(n, p, seed=1).astype(np.int32)
    pos = np.random.default_rng(0).standard_normal((n, D)).astype(np.float32)
    eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, 256)
    eng.set_positions(pos)
    eng.run(10); eng.sync()
    t0 = time.perf_counter(); eng.run(200); eng.sync(); dt = (time.perf_counter() - t0) / 200
    eng.timing_enable(True); eng.timing_reset(); eng.run(20); eng.sync()
    tm = {a: round(1e3 * b[0] / b[1], 1) for a, b in eng.timings().items()}
    print(f"{name}: E={len(edges)} {1e6 * dt:.0f} us/iter = {1 / dt:.0f} it/s", tm, flush=True)
    eng.close()
