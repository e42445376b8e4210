"""knn_method='ivf' (--probes -1: its exact mode) against the exact scan on a bench workload: recall of the neighbour rows, distance ratio, time per
iteration (whole steps, host-timed over `iters` iterations of a device-sampled run) and the per-kernel table.

    python tools/ivf_probe.py rr1m --dim 16 --S 4096 --probes 0,32,64,128 [--lists 0] [--iters 20] [--warm 10]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from graphem_rapids_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="rr1m")
ap.add_argument("--dim", type=int, default=0)
ap.add_argument("--S", type=int, default=0)
ap.add_argument("--k", type=int, default=0)
ap.add_argument("--lists", type=int, default=0)
ap.add_argument("--probes", default="0")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--warm", type=int, default=10)
a = ap.parse_args()

n, D, k, S, edges, pos = bench.make_workload(a.workload)
if a.dim:
    D = a.dim
    pos = (np.random.default_rng(0).standard_normal((n, D)) * 0.1).astype(np.float32)
S = a.S or S
k = a.k or k
E = len(edges)
rng = np.random.default_rng(1)


def run(method, probes=0):
    eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=3, knn_method=method, ivf_lists=a.lists, ivf_probes=probes)
    eng.set_positions(pos)
    eng.run(a.warm)                      # a layout a few iterations in, the same for every method? no: IVF changes it -> see below
    return eng


# the layout the rows are compared on: `warm` exact iterations
exact = run("scan")
layout = exact.get_positions()
sampled = rng.permutation(E)[:S].astype(np.int32)
rows_exact = exact.knn_midpoints(sampled)
mid = (layout[edges[:, 0]] + layout[edges[:, 1]]) / 2.0


def d2_of(rows):
    q = mid[sampled][:, None, :]
    return ((q - mid[rows]) ** 2).sum(-1)


d_exact = d2_of(rows_exact)
exact.sync()
t0 = time.perf_counter(); exact.run(a.iters); exact.sync(); t_exact = (time.perf_counter() - t0) / a.iters * 1e6
print(f"{a.workload} n={n} E={E} D={D} k={k} S={S}: scan {t_exact:.1f} us/iter")

for p in [int(x) for x in a.probes.split(",")]:
    eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=3, knn_method="ivf", ivf_lists=a.lists, ivf_probes=p)
    eng.set_positions(layout)
    rows = eng.knn_midpoints(sampled)
    C, P = eng.knn_ivf_config()
    hit = np.array([len(np.intersect1d(rows[i], rows_exact[i])) for i in range(S)])
    d_ivf = d2_of(rows)
    ratio = np.sqrt(d_ivf[:, -1] / np.maximum(d_exact[:, -1], 1e-30))
    _, _, ovf = eng.knn_last_counts()
    sz = np.sort(eng.knn_ivf_list_sizes())
    print(f"    list sizes: min {sz[0]} median {sz[len(sz) // 2]} p99 {sz[int(len(sz) * 0.99)]} max {sz[-1]} (mean {sz.mean():.0f})")
    eng.run(3); eng.sync()
    t0 = time.perf_counter(); eng.run(a.iters); eng.sync(); t = (time.perf_counter() - t0) / a.iters * 1e6
    eng.timing_enable(True); eng.run(5); eng.sync()
    tm = {kk: round(v[0] / max(v[1], 1) * 1e3, 1) for kk, v in eng.timings().items()} if hasattr(eng, "timings") else {}
    print(f"  ivf lists={C} probes={P}: recall {hit.sum() / (S * k):.4f} (rows complete {np.mean(hit == k):.3f}), "
          f"k-th distance ratio mean {ratio.mean():.4f} max {ratio.max():.3f}, exact fallbacks {int(ovf.sum())}, {t:.1f} us/iter  {tm}")
