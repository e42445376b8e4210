"""Random configurations of the inverted file in its exact mode against the scan on the same state and sample: graph
family and size, components, neighbours, queries, and the state (raw start, a few iterations in, rescaled, outliers,
collapsed clusters, duplicated positions).    python tools/ivf_fuzz.py [cases] [seed]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import graphem_rapids_amd as gra
from graphem_rapids_amd import _native

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    n = int(rng.integers(30000, 200000))
    deg = int(rng.choice([4, 6, 8, 12, 16]))
    fam = rng.choice(["rr", "er", "pp"])
    if fam == "rr":
        edges = gra.random_regular_edges(n - (n * deg) % 2, deg, seed=int(rng.integers(1 << 30)))
        n = n - (n * deg) % 2
    elif fam == "er":
        edges = gra.erdos_renyi_edges(n, deg / (n - 1), seed=int(rng.integers(1 << 30)))
    else:
        edges = gra.planted_partition_edges(n, max(2, n // 500), deg * 0.8, deg * 0.2, seed=int(rng.integers(1 << 30)))
    edges = np.ascontiguousarray(edges, dtype=np.int32)
    E = len(edges)
    if E < 70000:
        continue
    D = int(rng.integers(2, 9))
    k = int(rng.choice([1, 5, 10, 15, 32, 64]))
    S = int(rng.choice([2048, 3000, 4096, 8192]))
    state = rng.choice(["start", "layout", "scaled", "outliers", "clusters", "duplicates"])
    pos = rng.standard_normal((n, D)).astype(np.float32) * np.float32(0.1)
    if state in ("layout", "scaled", "outliers"):
        e = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, 10, 256, seed=1, knn_method="scan")
        e.set_positions(pos)
        e.run(int(rng.integers(3, 15)))
        pos = e.get_positions()
        e.close()
    if state == "scaled":
        pos *= np.float32(10.0 ** rng.integers(-3, 4))
    if state == "outliers":
        far = rng.permutation(n)[: int(rng.integers(1, 200))]
        pos[far] *= np.float32(10.0 ** rng.integers(2, 6))
    if state == "clusters":
        centres = rng.standard_normal((50, D)).astype(np.float32)
        pos = centres[rng.integers(0, 50, n)] + rng.standard_normal((n, D)).astype(np.float32) * np.float32(1e-3)
    if state == "duplicates":
        pos = pos[rng.integers(0, max(2, n // 100), n)]
    sampled = rng.permutation(E)[:S].astype(np.int32)
    rows = {}
    for m, kw in (("scan", {}), ("ivf", {"ivf_probes": -1})):
        e = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=2, knn_method=m, **kw)
        e.set_positions(pos)
        rows[m] = e.knn_midpoints(sampled)
        if m == "ivf":
            _, fin, ovf = e.knn_last_counts()
            cfg = e.knn_ivf_config()
        e.close()
    differ = int((rows["ivf"] != rows["scan"]).any(axis=1).sum())
    bad += differ
    print(f"case {c}: {fam} n={n} E={E} D={D} k={k} S={S} {state}: lists {cfg[0]}, rows that differ {differ}, exhaustive fallbacks {int(ovf.sum())}, "
          f"candidates max {int(fin.max())}", flush=True)
print("ok" if bad == 0 else f"FAILED: {bad} rows differ")
sys.exit(0 if bad == 0 else 1)
