"""Full-size reference cases (tests/golden/make_golden_large.py): inputs rebuilt from seeds, the
reference's outputs from the fixtures.  Shared by the CPU (oracle) and GPU (HIP) parity tests.

A case yields, per captured step t: the state pos_t the reference was in, its sample ids, its
cdist+topk neighbour ids, its intersection forces (dense), the sha1 of its spring forces and of its
next positions, and a row subsample of both.  For the 1 M-vertex cases the states between the steps
are not stored (12 MB each): they are regenerated with the oracle in its ATen mode (KNN phase =
cdist + topk restated from PyTorch's source, oracle/aten_cdist_topk.cpp), which reproduces the reference
bit for bit there -- verified step by step against the stored sha1, so a broken chain is detected, not
silently used.
"""
import functools
import hashlib
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SUB = 61
K, S, D = 10, 256, 3
PARAMS = (1.0, 0.2, 0.5)   # L_min, k_attr, k_inter: the reference's defaults (pt.py:57-59)


def sha1(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def dense_inter(g, t, n):
    F = np.zeros((n, D), dtype=np.float32)
    F[g[f"F_inter_rows_{t}"]] = g[f"F_inter_vals_{t}"]
    return F


@functools.lru_cache(maxsize=2)
def load(name):
    """-> dict(n, edges, g, states): states[t] = pos_t for every captured step (None where the chain broke)."""
    import graphem_rapids_amd as gra
    import oracle
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    if name == "c2_rr100k":
        edges = np.ascontiguousarray(g["edges"], dtype=np.int32)
        n = 100000
        states = [g["p0"], g["pos_next_0"], g["pos_next_1"]]
        return dict(name=name, n=n, edges=edges, g=g, states=states, steps=3)
    n = 1_000_000
    if name == "c3_er1m_ref":
        edges = gra.erdos_renyi_edges(n, 1e-5, seed=12345)
    elif name == "rr1m_ref":
        edges = gra.random_regular_edges(n, 8, seed=0)
    else:
        raise ValueError(name)
    edges = np.ascontiguousarray(edges, dtype=np.int32)
    assert sha1(edges) == str(g["edges_sha1"]), "generator drifted: the fixture belongs to a different graph"
    rng = np.random.default_rng(0)
    p0 = (rng.standard_normal((n, D)) * 0.1).astype(np.float32)
    assert sha1(p0) == str(g["p0_sha1"])
    states = [p0]
    for t in range(2):  # pos_{t+1} of the reference = oracle step from pos_t, if the sha1 says so
        prev = states[-1]
        nxt = None
        if prev is not None:
            cand = oracle.step_aten(prev, edges, g[f"sampled_{t}"], K, *PARAMS)
            if sha1(cand) == str(g[f"pos_next_sha1_{t}"]):
                nxt = cand
        states.append(nxt)
    return dict(name=name, n=n, edges=edges, g=g, states=states, steps=3)


def knn_agreement(knn, ref):
    """(rows identical in order, rows equal as sets, recall of ids)."""
    same = int((knn == ref).all(axis=1).sum())
    sets = int(sum(set(a) == set(b) for a, b in zip(knn, ref)))
    recall = float(np.mean([len(set(a) & set(b)) / len(b) for a, b in zip(knn, ref)]))
    return same, sets, recall


def explain_knn_differences(pos, edges, sampled, knn, ref):
    """Every place where `knn` (exact-difference distances, the sampled edge itself dropped) departs from the
    reference's cdist+topk ids must be something cdist's rounding explains.  pt.py:580 evaluates
    |q|^2 + |m|^2 - 2 q.m in fp32, so its squared distances are quantised in steps of ulp(|q|^2 + |m|^2) and
    carry a few such steps of noise; two ids whose exact squared distances are closer than that can swap,
    and the sampled edge's own distance is not 0, so a neighbour closer than the noise floor can take column 0
    and be dropped as "self" while the sampled edge itself stays in the list (pt.py:417-421).
    Checks, per differing column, that the fp64 squared distances of the two ids differ by at most 8 quanta,
    and that this build's row is in ascending exact order.  Returns (rows with a difference, of which rows
    where the reference kept the sampled edge itself, largest gap in quanta)."""
    worst, n_rows, n_self = 0.0, 0, 0
    for r in np.nonzero(~(knn == ref).all(axis=1))[0]:
        n_rows += 1
        e = int(sampled[r])
        n_self += int(e in ref[r])
        q = (pos[edges[e, 0]].astype(np.float64) + pos[edges[e, 1]].astype(np.float64)) / 2

        def d2(ids):
            m = (pos[edges[ids, 0]].astype(np.float64) + pos[edges[ids, 1]].astype(np.float64)) / 2
            return ((m - q) ** 2).sum(axis=1), (m ** 2).sum(axis=1)
        da, _ = d2(knn[r])
        dr, mr = d2(ref[r])
        assert np.all(np.diff(da) >= -1e-12 * da[-1]), f"row {r}: this build's neighbours are not in exact order"
        quantum = float(np.spacing(np.float32((q ** 2).sum() + mr.max())))
        for c in np.nonzero(knn[r] != ref[r])[0]:
            gap = abs(da[c] - dr[c]) / quantum
            assert gap <= 8.0, f"row {r} col {c}: ids {knn[r][c]} / {ref[r][c]}: exact d2 differ by {gap:.1f} cdist quanta"
            worst = max(worst, gap)
    return n_rows, n_self, worst
