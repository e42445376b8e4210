"""Probe: how much faster is the layout iteration when the SAME graph is fed with BFS vertex labels
(neighbours closer in memory -> more L2 hits in the spring phase's row gathers)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from graphem_rapids_amd import _native


def pull_lists(edges, n):
    E = len(edges)
    u, v = edges[:, 0].astype(np.int64), edges[:, 1].astype(np.int64)
    src = np.concatenate([u, v]); dst = np.concatenate([v, u])
    order = np.argsort(src, kind="stable")
    src, dst = src[order], dst[order]
    rowptr = np.zeros(n + 1, np.int64); np.add.at(rowptr, src + 1, 1)
    return np.cumsum(rowptr), dst


def bfs_labels(edges, n):
    rowptr, adj = pull_lists(edges, n)
    seen = np.zeros(n, bool); order = []
    for root in range(n):
        if seen[root]:
            continue
        seen[root] = True; frontier = np.array([root])
        while len(frontier):
            order.append(frontier)
            idx = np.concatenate([np.arange(rowptr[f], rowptr[f + 1]) for f in frontier.tolist()])
            nb = adj[idx]; nb = nb[~seen[nb]]
            _, first = np.unique(nb, return_index=True)
            nb = nb[np.sort(first)]
            seen[nb] = True; frontier = nb
    order = np.concatenate(order)
    new_id = np.empty(n, np.int64); new_id[order] = np.arange(n)
    return new_id


def run(tag, n, D, k, S, edges, pos):
    eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=0)
    eng.set_positions(pos)
    eng.run(5); eng.sync()
    t0 = time.perf_counter(); eng.run(50); eng.sync(); dt = (time.perf_counter() - t0) / 50
    eng.timing_enable(True); eng.timing_reset(); eng.run(20); eng.sync()
    tm = {a: round(1e3 * b[0] / b[1], 1) for a, b in eng.timings().items()}
    print(f"{tag}: {1 / dt:.0f} it/s", tm, flush=True)
    eng.close()


wl = sys.argv[1] if len(sys.argv) > 1 else "rr1m"
n, D, k, S, edges, pos = bench.make_workload(wl)
run("natural labels", n, D, k, S, edges, pos)
nid = bfs_labels(edges, n)
e2 = nid[edges.astype(np.int64)]
e2 = np.sort(e2, axis=1)
e2 = e2[np.lexsort((e2[:, 1], e2[:, 0]))].astype(np.int32)
p2 = np.empty_like(pos); p2[nid] = pos
run("bfs labels", n, D, k, S, np.ascontiguousarray(e2), p2)
