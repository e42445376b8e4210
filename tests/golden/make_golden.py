#!/usr/bin/env python3
"""Generate golden vectors from the reference's PyTorch-CPU backend.

Runs ONLY in the build container, where /root/reference is mounted read-only.
Nothing from the reference is copied: the outputs are data (inputs and expected
outputs of the hot path) written as compressed .npz next to this
script.  The GPU box never sees the reference; tests there read the .npz files.

Usage:  PYTHONDONTWRITEBYTECODE=1 GRAPHEM_RAPIDS_QUIET=true python tests/golden/make_golden.py

Each case is produced in a FRESH subprocess because the reference's Laplacian
initialisation is reproducible only for the first eigsh call of a process
(SURVEY.md quirk Q11).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

CASES = {
    # name: (graph spec, constructor kwargs, iterations, captured steps)
    "c1_er1000": dict(graph=("er", 1000, 0.01, 0), D=3, kw=dict(seed=0),
                      iters=50, steps=[0, 1, 2, 5, 10, 25, 49]),
    "rr50_d4": dict(graph=("rr", 50, 4, 42), D=2,
                    kw=dict(L_min=10.0, k_attr=0.5, k_inter=0.1, n_neighbors=15, sample_size=256, seed=1),
                    iters=6, steps=[0, 1, 5]),
    "rr200_s64": dict(graph=("rr", 200, 4, 7), D=2,
                      kw=dict(n_neighbors=8, sample_size=64, seed=3),
                      iters=6, steps=[0, 1, 5]),
    "two_triangles": dict(graph=("tri",), D=2,
                          kw=dict(L_min=10.0, k_attr=0.5, k_inter=0.1, n_neighbors=5, sample_size=6, seed=2),
                          iters=2, steps=[0, 1]),
    "two_hexagons": dict(graph=("hex",), D=2, kw=dict(seed=4), iters=8, steps=[0, 1, 7]),
    "d16_er2000": dict(graph=("er", 2000, 0.005, 5), D=16,
                       kw=dict(n_neighbors=32, sample_size=256, seed=5),
                       iters=4, steps=[0, 1, 3]),
    "d4_rr300": dict(graph=("rr", 300, 6, 11), D=4, kw=dict(n_neighbors=10, sample_size=128, seed=6),
                     iters=4, steps=[0, 3]),
    # the reference computing in float64 (pt.py:56 dtype; tests/test_pytorch_backend.py:169-181): fixtures of the float64 engine
    "c1_er1000_f64": dict(graph=("er", 1000, 0.01, 0), D=3, kw=dict(seed=0, dtype="float64"),
                          iters=6, steps=[0, 1, 2, 5]),
    "d16_er2000_f64": dict(graph=("er", 2000, 0.005, 5), D=16,
                           kw=dict(n_neighbors=32, sample_size=256, seed=5, dtype="float64"),
                           iters=3, steps=[0, 2]),
}


def worker(name):
    import logging
    import types
    for mod in ("ndlib", "ndlib.models", "ndlib.models.ModelConfig", "ndlib.models.epidemics", "loguru"):
        sys.modules[mod] = types.ModuleType(mod)  # deps unrelated to the hot path
    sys.modules["loguru"].logger = logging.getLogger("loguru-stub")
    sys.path.insert(0, REF)
    import numpy as np
    import scipy.sparse as sp
    import torch
    import graphem_rapids as gr

    spec = CASES[name]
    g = spec["graph"]
    if g[0] == "er":
        adj = gr.erdos_renyi_graph(n=g[1], p=g[2], seed=g[3])
    elif g[0] == "rr":
        adj = gr.generate_random_regular(n=g[1], d=g[2], seed=g[3])
    elif g[0] == "tri":
        adj = np.array([[0, 1, 1, 0, 0, 0], [1, 0, 1, 0, 0, 0], [1, 1, 0, 0, 0, 0],
                        [0, 0, 0, 0, 1, 1], [0, 0, 0, 1, 0, 1], [0, 0, 0, 1, 1, 0]])
    elif g[0] == "hex":
        e = np.array([[0, 1], [1, 2], [2, 3], [3, 4], [4, 5], [5, 0],
                      [6, 7], [7, 8], [8, 9], [9, 10], [10, 11], [11, 6]])
        a = sp.csr_matrix((np.ones(len(e)), (e[:, 0], e[:, 1])), shape=(12, 12))
        adj = a + a.T
    else:
        raise ValueError(g)

    kw = dict(spec["kw"])
    if kw.get("dtype") == "float64":
        kw["dtype"] = torch.float64
    emb = gr.create_graphem(adj, n_components=spec["D"], backend="pytorch", device="cpu",
                            verbose=False, **kw)
    out = {}
    edges = emb.edges.numpy()
    out["edges"] = edges.astype(np.int32)
    out["n"] = np.int64(emb.n)
    out["D"] = np.int64(emb.n_components)
    out["params"] = np.array([emb.L_min, emb.k_attr, emb.k_inter], dtype=np.float64)
    out["k"] = np.int64(emb.n_neighbors)
    out["S"] = np.int64(emb.sample_size)
    out["p0"] = emb.get_positions().copy()
    out["steps"] = np.array(spec["steps"], dtype=np.int64)
    samples = []
    for t in range(spec["iters"]):
        X = emb._positions
        Ed = emb.edges
        st = torch.get_rng_state()
        # replay of the phases of update_positions on the same state and the same sample
        F_s = emb._compute_spring_forces(X, Ed)
        mid = (X[Ed[:, 0]] + X[Ed[:, 1]]) / 2.0
        knn, samp = emb._locate_knn_midpoints(mid, emb.n_neighbors)
        F_i = emb._compute_intersection_forces(X, Ed, knn, samp)
        samples.append(samp.numpy().astype(np.int32))
        torch.set_rng_state(st)
        emb.update_positions()
        if t in spec["steps"]:
            out[f"pos_{t}"] = X.numpy().copy()
            out[f"sampled_{t}"] = samp.numpy().astype(np.int32)
            out[f"knn_{t}"] = knn.numpy().astype(np.int32)
            out[f"F_spring_{t}"] = F_s.numpy().copy()
            out[f"F_inter_{t}"] = F_i.numpy().copy()
            out[f"pos_next_{t}"] = emb._positions.numpy().copy()
    out["sample_stream"] = np.stack(samples)
    out["pos_final"] = emb.get_positions().copy()
    r = np.linalg.norm(out["pos_final"], axis=1)
    out["seeds_10"] = np.argsort(-r)[:10].astype(np.int64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    import hashlib
    print(name, "E=%d" % len(edges), "sha1(p0)=%s" % hashlib.sha1(out["p0"].tobytes()).hexdigest()[:12],
          "sha1(final)=%s" % hashlib.sha1(out["pos_final"].tobytes()).hexdigest()[:12])


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker(sys.argv[1])
    else:
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", GRAPHEM_RAPIDS_QUIET="true")
        for case in CASES:
            subprocess.run([sys.executable, os.path.abspath(__file__), case], check=True, env=env, cwd="/tmp")
