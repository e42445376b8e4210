"""The CPU oracle (oracle/graphem_oracle.c) against golden vectors produced by the
reference's PyTorch-CPU backend (tests/golden/make_golden.py).  CPU only.

The reference's own tests hold no golden numbers, so these
fixtures are the pin: the oracle must reproduce every captured phase output, every
captured step and the whole trajectory BIT FOR BIT.
"""
import numpy as np
import pytest

import oracle


def _params(g):
    Lm, ka, ki = (float(x) for x in g["params"])
    return int(g["D"]), int(g["k"]), Lm, ka, ki


def test_spring_forces(golden):
    D, k, Lm, ka, ki = _params(golden)
    for t in golden["steps"]:
        F = oracle.spring_forces(golden[f"pos_{t}"], golden["edges"], Lm, ka)
        assert np.array_equal(F, golden[f"F_spring_{t}"]), f"step {t}"


def test_knn_rows_identical(golden):
    """Same neighbour ids in the same order as torch.cdist+topk gave (pt.py:580-583, column 0 dropped)."""
    D, k, Lm, ka, ki = _params(golden)
    for t in golden["steps"]:
        knn = oracle.knn_midpoints(golden[f"pos_{t}"], golden["edges"], golden[f"sampled_{t}"], k)
        assert np.array_equal(knn, golden[f"knn_{t}"]), f"step {t}"


def test_intersection_forces(golden):
    D, k, Lm, ka, ki = _params(golden)
    for t in golden["steps"]:
        F = oracle.intersection_forces(golden[f"pos_{t}"], golden["edges"], golden[f"sampled_{t}"],
                                       golden[f"knn_{t}"], ki)
        assert np.array_equal(F, golden[f"F_inter_{t}"]), f"step {t}"


def test_integrate_normalise(golden):
    """Column-major reduction order: the reference's tensors are column-major after its Laplacian start."""
    for t in golden["steps"]:
        out = oracle.integrate_normalise(golden[f"pos_{t}"], golden[f"F_spring_{t}"], golden[f"F_inter_{t}"],
                                         colmajor=True)
        assert np.array_equal(out, golden[f"pos_next_{t}"]), f"step {t}"
        out_c = oracle.integrate_normalise(golden[f"pos_{t}"], golden[f"F_spring_{t}"], golden[f"F_inter_{t}"],
                                           colmajor=False)
        np.testing.assert_allclose(out_c, golden[f"pos_next_{t}"], rtol=0, atol=2e-6)


def test_single_step(golden):
    D, k, Lm, ka, ki = _params(golden)
    for t in golden["steps"]:
        out = oracle.step(golden[f"pos_{t}"], golden["edges"], golden[f"sampled_{t}"], k, Lm, ka, ki, colmajor=True)
        assert np.array_equal(out, golden[f"pos_next_{t}"]), f"step {t}"


def test_full_trajectory(golden):
    D, k, Lm, ka, ki = _params(golden)
    fin = oracle.run_layout(golden["p0"], golden["edges"], golden["sample_stream"], k, Lm, ka, ki, colmajor=True)
    assert np.array_equal(fin, golden["pos_final"])
    r = np.linalg.norm(fin, axis=1)  # caller contract of influence.graphem_seed_selection (influence.py:31-35)
    assert np.array_equal(np.argsort(-r)[:10], golden["seeds_10"])


def test_sample_stream_contract(golden):
    """S >= E draws nothing and uses arange(E) (pt.py:412); otherwise S distinct ids < E."""
    E, S = len(golden["edges"]), int(golden["S"])
    ss = golden["sample_stream"]
    assert ss.shape[1] == S
    if S >= E:
        assert (ss == np.arange(E)).all()
    else:
        assert all(len(set(row)) == S for row in ss) and ss.max() < E and ss.min() >= 0


def test_topk_error_when_k_exceeds_edges():
    """torch.topk raises RuntimeError when k+1 > E."""
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN_DIR, "two_triangles.npz"))
    with pytest.raises(RuntimeError):
        oracle.knn_midpoints(g["p0"], g["edges"], np.arange(6, dtype=np.int32), 6)


@pytest.mark.parametrize("n,D", [(6, 2), (7, 3), (50, 2), (1000, 3), (300, 4), (300, 5), (5000, 8), (2000, 16),
                                 (500, 33), (64, 250), (20000, 3)])
def test_column_sums_match_torch(n, D):
    """torch.sum(dim=0) summation order, both memory layouts, bit for bit (torch CPU is on both boxes)."""
    import torch
    rng = np.random.default_rng(n * 131 + D)
    a = (rng.standard_normal((n, D)) + 0.3).astype(np.float32)
    t_row = torch.from_numpy(a)
    assert np.array_equal(oracle.column_sums(a), torch.sum(t_row, dim=0).numpy())
    t_col = torch.from_numpy(np.asfortranarray(a))  # stride (1, n) like the reference's Laplacian start
    assert t_col.stride() == (1, n) or n == 1
    assert np.array_equal(oracle.column_sums(a, colmajor=True), torch.sum(t_col, dim=0).numpy())


@pytest.mark.parametrize("D", [1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 16, 31, 33, 250])
def test_row_norm_matches_torch(D):
    """torch.norm(dim=1) order (pt.py:623) via spring forces on a path graph: |F| encodes the norm."""
    import torch
    rng = np.random.default_rng(D)
    pos = rng.standard_normal((400, D)).astype(np.float32)
    edges = np.stack([np.arange(0, 399), np.arange(1, 400)], 1).astype(np.int32)
    diff = torch.from_numpy(pos[edges[:, 1]] - pos[edges[:, 0]])
    dist = torch.norm(diff, dim=1, keepdim=True) + 1e-6
    ef = (-0.2 * (dist - 1.0)) * (diff / dist)
    F = torch.zeros(400, D)
    F.index_add_(0, torch.from_numpy(edges[:, 0]).long(), ef)
    F.index_add_(0, torch.from_numpy(edges[:, 1]).long(), -ef)
    assert np.array_equal(oracle.spring_forces(pos, edges, 1.0, 0.2), F.numpy())


def test_bench_baselines_equal_the_oracle(golden):
    """bench.py's CPU baselines are the same algorithm: the all-cores port (oracle.OmpStepper: spring forces by
    per-vertex pull, bit-identical; normalise with fp64 column sums) and the PyTorch-CPU restatement
    (oracle/torch_cpu.py: the reference's torch ops without its MemoryManager wrappers)."""
    import torch
    from oracle import torch_cpu
    g = golden
    edges, n, k = g["edges"], int(g["n"]), int(g["k"])
    Lm, ka, ki = (float(x) for x in g["params"])
    st = oracle.OmpStepper(n, edges)
    te = torch.from_numpy(edges.astype(np.int64))
    for t in g["steps"]:
        pos, sampled = g[f"pos_{t}"], g[f"sampled_{t}"]
        assert np.array_equal(st.spring_forces(pos, Lm, ka), g[f"F_spring_{t}"])
        out = st.step(pos, sampled, k, Lm, ka, ki)
        assert np.abs(out - g[f"pos_next_{t}"]).max() <= 2e-6
        assert np.array_equal(oracle.knn_midpoints(pos, edges, sampled, k, tiled=True),
                              oracle.knn_midpoints(pos, edges, sampled, k))
        tp = torch_cpu.step(torch.from_numpy(np.ascontiguousarray(pos)), te, torch.from_numpy(sampled.astype(np.int64)),
                            k, Lm, ka, ki).numpy()
        assert np.abs(tp - g[f"pos_next_{t}"]).max() <= 2e-6
        tk = torch_cpu.knn_midpoints((torch.from_numpy(np.ascontiguousarray(pos))[te[:, 0]] +
                                      torch.from_numpy(np.ascontiguousarray(pos))[te[:, 1]]) / 2.0,
                                     torch.from_numpy(sampled.astype(np.int64)), k).numpy()
        assert np.array_equal(tk, g[f"knn_{t}"])


def test_tiled_knn_of_the_baseline_equals_the_plain_search_with_ties():
    """The blocked search bench.py's all-cores baseline runs (go_knn_midpoints_tiled) returns the rows of the per-query
    search, including on a lattice where most distances tie (ties by smaller id across blocks and threads)."""
    rng = np.random.default_rng(5)
    n = 3000
    pos = rng.integers(0, 3, size=(n, 2)).astype(np.float32)     # midpoints on a 5 x 5 lattice: ties everywhere
    edges = rng.integers(0, n, size=(9000, 2)).astype(np.int32)
    edges = edges[edges[:, 0] != edges[:, 1]]
    sampled = rng.choice(len(edges), 200, replace=False).astype(np.int32)
    for k in (1, 7, 40):
        assert np.array_equal(oracle.knn_midpoints(pos, edges, sampled, k, tiled=True),
                              oracle.knn_midpoints(pos, edges, sampled, k))
    pos = rng.standard_normal((n, 5)).astype(np.float32)
    assert np.array_equal(oracle.knn_midpoints(pos, edges, sampled, 15, tiled=True),
                          oracle.knn_midpoints(pos, edges, sampled, 15))
