"""Long run through the inverted file in its exact mode: every `every` iterations the rows it returns on the current
layout are compared with the scan's on the same layout and sample, and the queries handed to the exhaustive search are
counted.    python tools/ivf_soak.py rr1m --dim 3 --S 4096 --iters 300 --every 25
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from graphem_rapids_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="rr1m")
ap.add_argument("--dim", type=int, default=0)
ap.add_argument("--S", type=int, default=4096)
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--every", type=int, default=25)
ap.add_argument("--probes", type=int, default=-1)
a = ap.parse_args()
n, D, k, S, edges, pos = bench.make_workload(a.workload)
if a.dim:
    D = a.dim
    pos = (np.random.default_rng(0).standard_normal((n, D)) * 0.1).astype(np.float32)
S = a.S
ivf = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=3, knn_method="ivf", ivf_probes=a.probes)
scan = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=3, knn_method="scan")
ivf.set_positions(pos)
rng = np.random.default_rng(1)
bad = 0
for it in range(0, a.iters, a.every):
    ivf.run(a.every)
    layout = ivf.get_positions()
    sampled = rng.permutation(len(edges))[:S].astype(np.int32)
    rows = ivf.knn_midpoints(sampled)
    _, fin, ovf = ivf.knn_last_counts()
    scan.set_positions(layout)
    ref = scan.knn_midpoints(sampled)
    differ = int((rows != ref).any(axis=1).sum())
    bad += differ
    sz = ivf.knn_ivf_list_sizes()
    print(f"iteration {it + a.every}: rows that differ {differ} of {S}, exhaustive fallbacks {int(ovf.sum())}, candidates max {int(fin.max())}, "
          f"lists {len(sz)} sizes {sz.min()}..{sz.max()}, |x| max {np.abs(layout).max():.1f}", flush=True)
print("ok" if bad == 0 or a.probes >= 0 else "FAILED")
sys.exit(0 if bad == 0 or a.probes >= 0 else 1)
