"""Drop-in behaviour of GraphEmbedderHIP / create_graphem on the GPU: the reference's own
property tests (tests/test_pytorch_backend.py, tests/test_embedder.py, tests/test_integration.py
in the reference tree) restated against the HIP backend."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _rr(n=50, d=4, seed=42):
    import graphem_rapids_amd as gra
    return gra.generate_random_regular(n=n, d=d, seed=seed)


def test_initialization_attributes():
    import torch
    import graphem_rapids_amd as gra
    emb = gra.GraphEmbedderHIP(_rr(), n_components=2, L_min=10.0, k_attr=0.5, k_inter=0.1, n_neighbors=15,
                               sample_size=256, verbose=False)
    assert emb.n == 50 and emb.n_components == 2 and emb.n_edges == 100
    assert emb.sample_size == 100  # min(sample_size, E)
    assert emb.positions.shape == (50, 2) and isinstance(emb.positions, np.ndarray)
    assert isinstance(emb._positions, torch.Tensor) and emb._positions.device.type == "cuda"
    assert emb._positions.dtype == torch.float32 and emb.device.type == "cuda"
    assert tuple(emb.edges.shape) == (100, 2)
    assert "GraphEmbedderHIP" in repr(emb)


def test_layout_changes_positions_and_stays_finite():
    import graphem_rapids_amd as gra
    emb = gra.create_graphem(_rr(30, 4), n_components=3, backend="hip", verbose=False, seed=1)
    p0 = emb.get_positions().copy()
    out = emb.run_layout(num_iterations=5)
    assert out.shape == (30, 3) and np.isfinite(out).all()
    assert not np.allclose(out, p0)
    for _ in range(3):  # repeated calls continue from the current state
        emb.run_layout(num_iterations=2)
        assert np.isfinite(emb.positions).all() and np.abs(emb.positions).max() < 1000


def test_positions_setter_and_device_view():
    import torch
    import graphem_rapids_amd as gra
    emb = gra.create_graphem(_rr(), n_components=2, verbose=False)
    new = np.random.default_rng(0).standard_normal((50, 2)).astype(np.float32)
    emb.positions = new
    assert np.array_equal(emb.positions, new)
    assert np.array_equal(emb._positions.cpu().numpy(), new)
    emb.positions = torch.from_numpy(new * 2)
    assert np.array_equal(emb.get_positions(), new * 2)


def test_dtype_plumbing():
    import torch
    import graphem_rapids_amd as gra
    emb = gra.GraphEmbedderHIP(_rr(), n_components=2, dtype=torch.float64, verbose=False)
    assert emb._positions.dtype == torch.float64 and emb.positions.dtype == np.float64
    emb.run_layout(2)
    assert np.isfinite(emb.positions).all()


def test_parameter_validation_and_errors():
    import graphem_rapids_amd as gra
    adj = _rr()
    with pytest.raises(ValueError):
        gra.GraphEmbedderHIP(adj, n_components=0, verbose=False)
    with pytest.raises(ValueError):
        gra.GraphEmbedderHIP(adj, n_components=2, k_attr=-1.0, verbose=False)
    with pytest.raises(RuntimeError):
        gra.GraphEmbedderHIP(adj, n_components=2, device="invalid_device", verbose=False)
    with pytest.raises(RuntimeError):
        gra.GraphEmbedderHIP(adj, n_components=2, device="cpu", verbose=False)
    with pytest.raises(ValueError):
        gra.GraphEmbedderHIP(np.zeros((3, 4)), verbose=False)
    with pytest.raises((ValueError, RuntimeError)):  # all-zero adjacency (reference: ValueError|RuntimeError)
        gra.create_graphem(sp.csr_matrix((10, 10)), n_components=2, verbose=False).run_layout(2)
    with pytest.raises(ValueError):
        gra.create_graphem(adj, backend="nonsense")


def test_reproducibility_same_seed():
    import graphem_rapids_amd as gra
    adj = _rr(60, 4, 3)
    a = gra.create_graphem(adj, n_components=2, verbose=False, seed=123, init="random").run_layout(6)
    b = gra.create_graphem(adj, n_components=2, verbose=False, seed=123, init="random").run_layout(6)
    assert np.array_equal(a, b)


def test_matches_reference_trajectory_through_the_public_api():
    """create_graphem(...).run_layout(5) with the reference's own sample stream and start."""
    import torch
    import graphem_rapids_amd as gra
    from conftest import load_golden
    g = load_golden("c1_er1000")
    n = int(g["n"])
    e = g["edges"]
    adj = sp.csr_matrix((np.ones(len(e)), (e[:, 0], e[:, 1])), shape=(n, n))
    adj = adj + adj.T
    emb = gra.create_graphem(adj, n_components=3, backend="hip", verbose=False, sampler="torch", init="random")
    assert np.array_equal(emb._edges_np, e)
    emb.positions = g["p0"]
    torch.manual_seed(0)  # the reference seeded with 0 and drew nothing before its first randperm
    emb.run_layout(5)
    assert np.abs(emb.positions - g["pos_5"]).max() <= 1e-3
    seeds = gra.graphem_seed_selection(emb, 10, num_iterations=0)
    assert len(seeds) == 10


def test_disconnected_graphs():
    import graphem_rapids_amd as gra
    tri = np.array([[0, 1, 1, 0, 0, 0], [1, 0, 1, 0, 0, 0], [1, 1, 0, 0, 0, 0],
                    [0, 0, 0, 0, 1, 1], [0, 0, 0, 1, 0, 1], [0, 0, 0, 1, 1, 0]])
    emb = gra.GraphEmbedderHIP(tri, n_components=2, L_min=10.0, k_attr=0.5, k_inter=0.1, n_neighbors=5,
                               sample_size=6, verbose=False)
    emb.run_layout(num_iterations=2)
    assert emb.positions.shape == (6, 2) and np.isfinite(emb.positions).all()


def test_high_dimension_random_init_path():
    """D >= n-1 makes eigsh fail -> the random-start fallback, as in the reference's D=250/300 tests."""
    import graphem_rapids_amd as gra
    emb = gra.GraphEmbedderHIP(_rr(100, 4, 1), n_components=250, n_neighbors=5, sample_size=32, verbose=False)
    out = emb.run_layout(2)
    assert out.shape == (100, 250) and np.isfinite(out).all()


def test_point_knn_helpers():
    """_compute_knn_chunked / _compute_knn_torch / _get_adaptive_chunk_size (reference
    tests/test_pytorch_backend.py:454-494, 525-558): shape, range, agreement, and exactness."""
    import torch
    import graphem_rapids_amd as gra
    emb = gra.GraphEmbedderHIP(_rr(), n_components=2, verbose=False)
    q = torch.randn(10, 2, device=emb.device, dtype=emb.dtype)
    ref = torch.randn(20, 2, device=emb.device, dtype=emb.dtype)
    a = emb._compute_knn_chunked(q, ref, 3)
    b = emb._compute_knn_torch(q, ref, 3, 5)
    assert tuple(a.shape) == (10, 3) and a.dtype == torch.long and a.device.type == "cuda"
    assert int(a.min()) >= 0 and int(a.max()) < 20 and torch.equal(a, b)
    d = torch.cdist(q.cpu().double(), ref.cpu().double())
    assert torch.equal(a.cpu(), torch.topk(d, 3, dim=1, largest=False).indices)
    assert emb._get_adaptive_chunk_size(100, 1000, "torch") > 0
    big_q, big_r = torch.randn(300, 7), torch.randn(5000, 7)
    got = emb._compute_knn_chunked(big_q, big_r, 80).cpu()  # k > 64: sort kernel
    want = torch.topk(torch.cdist(big_q.double(), big_r.double()), 80, dim=1, largest=False).indices
    assert (got == want).float().mean() > 0.999  # fp32 vs fp64 distances may swap near-ties
    with pytest.raises(RuntimeError):
        emb._compute_knn_chunked(q, ref, 21)


def test_pykeops_named_method_is_the_exact_distance_knn():
    """_compute_knn_pykeops (pt.py:485-541: argKmin of the exact-difference squared distance) under its reference name."""
    import torch
    import graphem_rapids_amd as gra
    emb = gra.create_graphem(_rr(), n_components=3, verbose=False)
    rng = np.random.default_rng(3)
    q = rng.standard_normal((37, 3)).astype(np.float32)
    r = rng.standard_normal((500, 3)).astype(np.float32)
    got = emb._compute_knn_pykeops(torch.from_numpy(q), torch.from_numpy(r), 7, 16).cpu().numpy()
    d2 = ((q[:, None, :].astype(np.float64) - r[None, :, :]) ** 2).sum(-1)
    assert np.array_equal(got, np.argsort(d2, axis=1, kind="stable")[:, :7])
    assert emb._has_pykeops is False


def test_factory_and_integration_properties():
    """Properties of the reference's integration tests (tests/test_integration.py:44-46, 136-138,
    169-174, 253-270): spread, per-dimension variance, parameter sensitivity, odd batch/sample sizes."""
    import graphem_rapids_amd as gra
    for adj in (gra.erdos_renyi_graph(300, 0.03, seed=1), gra.generate_random_regular(200, 4, seed=2)):
        emb = gra.create_graphem(adj, n_components=3, backend="hip", verbose=False, seed=3)
        pos = emb.run_layout(num_iterations=10)
        radii = np.linalg.norm(pos, axis=1)
        assert np.isfinite(pos).all() and radii.std() > 0.1 and radii.max() < 100
        assert (pos.var(axis=0) > 1e-6).all()
    adj = gra.generate_random_regular(150, 4, seed=5)
    a = gra.create_graphem(adj, n_components=2, verbose=False, seed=7, k_attr=0.1, k_inter=0.2).run_layout(8)
    b = gra.create_graphem(adj, n_components=2, verbose=False, seed=7, k_attr=0.8, k_inter=1.5).run_layout(8)
    assert np.abs(a - b).mean() > 1e-3
    emb = gra.create_graphem(adj, n_components=2, verbose=False, seed=7, batch_size=64, sample_size=100)
    assert emb.batch_size == 64 and emb.sample_size == 100
    assert np.isfinite(emb.run_layout(4)).all()


def test_update_positions_loop_equals_run_layout():
    """Scripts in the reference loop update_positions() themselves (benchmarks/run_benchmarks.py:292-293)."""
    import torch
    import graphem_rapids_amd as gra
    adj = gra.generate_random_regular(120, 4, seed=9)
    a = gra.create_graphem(adj, n_components=3, verbose=False, seed=11, init="random", sampler="torch")
    b = gra.create_graphem(adj, n_components=3, verbose=False, seed=11, init="random", sampler="torch")
    torch.manual_seed(5)
    for _ in range(4):
        a.update_positions()
    torch.manual_seed(5)
    b.run_layout(4)
    assert np.array_equal(a.get_positions(), b.get_positions())


def test_seed_selection_contract():
    """influence.graphem_seed_selection (influence.py:10-37): k vertices with the largest radius."""
    import graphem_rapids_amd as gra
    adj = gra.generate_random_regular(100, 4, seed=1)
    emb = gra.create_graphem(adj, n_components=3, verbose=False, seed=2)
    seeds = gra.graphem_seed_selection(emb, 7, num_iterations=3)
    r = np.linalg.norm(emb.positions, axis=1)
    assert seeds == np.argsort(-r)[:7].tolist()


def test_two_hexagons_separate():
    """tests/test_integration.py:274-311 of the reference: disconnected components drift apart."""
    import graphem_rapids_amd as gra
    e = np.array([[0, 1], [1, 2], [2, 3], [3, 4], [4, 5], [5, 0], [6, 7], [7, 8], [8, 9], [9, 10], [10, 11], [11, 6]])
    adj = sp.csr_matrix((np.ones(len(e)), (e[:, 0], e[:, 1])), shape=(12, 12))
    adj = adj + adj.T
    pos = gra.create_graphem(adj, n_components=2, backend="hip", verbose=False).run_layout(num_iterations=8)
    assert pos.shape == (12, 2) and np.isfinite(pos).all()
    assert np.linalg.norm(pos[:6].mean(0) - pos[6:].mean(0)) > 1e-2


def test_internal_vertex_order_is_invisible_to_the_caller():
    """gh_params.reorder: the breadth-first internal vertex order changes where rows sit on the device
    (gh_vertex_order, gh_positions_device) and nothing the caller sees: same forces, same KNN, same
    positions, in the caller's vertex order."""
    import torch
    from graphem_rapids_amd import _native
    from graphem_rapids_amd.embedder_hip import device_view
    import graphem_rapids_amd as gra
    n, D, k, S = 30000, 3, 10, 256
    edges = gra.random_regular_edges(n, 8, seed=5).astype(np.int32)
    rng = np.random.default_rng(5)
    pos = rng.standard_normal((n, D)).astype(np.float32)
    stream = np.stack([rng.permutation(len(edges))[:S] for _ in range(3)]).astype(np.int32)
    out = {}
    for mode in ("off", "bfs"):
        eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, reorder=mode)
        eng.set_positions(pos)
        order = eng.vertex_order()
        assert np.array_equal(np.sort(order), np.arange(n))
        if not os.environ.get("GRAPHEM_HIP_REORDER"):   # the override of the test runs beats the parameter
            assert (mode == "off") == np.array_equal(order, np.arange(n))
        dev = device_view(eng.positions_device_ptr(), (n, eng.ld), torch.float32, torch.device("cuda", 0), eng)
        assert np.array_equal(dev.cpu().numpy()[order, :D], pos)            # row of vertex v = order[v]
        flat = device_view(eng.positions_unpadded_device_ptr(), (n, D), torch.float32, torch.device("cuda", 0), eng)
        assert np.array_equal(flat.cpu().numpy(), pos)                       # caller's order, no padding
        F = eng.spring_forces()
        knn = eng.knn_midpoints(stream[0])
        eng.run(3, stream)
        out[mode] = (F, knn, eng.get_positions())
        eng.close()
    assert np.array_equal(out["off"][0], out["bfs"][0])
    assert np.array_equal(out["off"][1], out["bfs"][1])
    assert np.abs(out["off"][2] - out["bfs"][2]).max() <= 2e-6   # column sums are taken in row order


def test_gather_layout_argument_checks():
    from graphem_rapids_amd import _native
    import graphem_rapids_amd as gra
    n = 5000
    edges = gra.random_regular_edges(n, 6, seed=1).astype(np.int32)
    eng = _native.Engine(n, 3, edges, 1.0, 0.2, 0.5, 5, 64, partition=(0, 2500, 0, 0, _native.EDGES_HASHED))
    with pytest.raises(ValueError):
        eng.step_finish_gathered()            # no layout yet
    with pytest.raises(ValueError):
        eng.gather_layout(2, 1, 2500)         # this engine holds the rows of rank 0
    with pytest.raises(ValueError):
        eng.gather_layout(2, 0, 2000)         # chunks do not cover n
    eng.gather_layout(2, 0, 2500)
    assert eng.gather_slot_bytes() >= 2500 * 4 * 4 + eng.stats_rows() * 4 * 8
    with pytest.raises(ValueError):
        eng.gather_layout(2, 0, 2500)         # only once
    eng.close()


@pytest.mark.parametrize("n,D,k", [(1000, 3, 10), (200000, 3, 50), (5000, 16, 64), (300, 2, 1), (40, 5, 40)])
def test_radial_topk_on_device_equals_numpy(n, D, k):
    """gh_radial_topk == np.argsort(-np.linalg.norm(positions, axis=1))[:k] (influence.py:31-35),
    with and without the internal vertex order, ties resolved on the smaller id."""
    from graphem_rapids_amd import _native
    import graphem_rapids_amd as gra
    edges = gra.random_regular_edges(n, 4, seed=3).astype(np.int32)
    rng = np.random.default_rng(n)
    pos = rng.standard_normal((n, D)).astype(np.float32)
    pos[rng.permutation(n)[:5]] *= np.float32(3.0)
    pos[7] = pos[3]                                   # an exact tie
    radial = np.linalg.norm(pos, axis=1)
    want = np.lexsort((np.arange(n), -radial))[:k]    # farthest first, smaller id on ties
    for mode in ("off", "bfs"):
        eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, 3, 16, reorder=mode)
        eng.set_positions(pos)
        got = eng.radial_topk(k)
        eng.close()
        assert np.array_equal(got, want), mode
    with pytest.raises(ValueError):
        eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, 3, 16)
        eng.radial_topk(65)


def test_integration_md_binding_stub_runs():
    """The ctypes stub INTEGRATION.md shows a maintainer of the reference (section B) is executed as written
    (only the library path is made absolute) and must lay a graph out through the bare C ABI."""
    import re
    import graphem_rapids_amd as gra
    from graphem_rapids_amd import _native
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    section = text[text.index("## B."):]
    code = re.search(r"```python\n(.*?)```", section, re.S).group(1)
    assert 'ctypes.CDLL("libgraphem_hip.so")' in code
    code = code.replace('ctypes.CDLL("libgraphem_hip.so")', f'ctypes.CDLL({_native.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    adj = gra.generate_random_regular(n=3000, d=6, seed=1)
    emb = ns["GraphEmbedderHIP"](adj, n_components=3, seed=4)
    p0 = np.random.default_rng(0).standard_normal((3000, 3)).astype(np.float32)
    emb.positions = p0
    assert np.array_equal(emb.positions, p0)
    out = emb.run_layout(5)
    assert out.shape == (3000, 3) and np.isfinite(out).all()
    assert np.abs(out.std(axis=0, ddof=1) - 1).max() < 1e-3
    emb.update_positions()
    assert not np.array_equal(emb.get_positions(), out)


def test_graph_replay_equals_the_enqueued_loop(monkeypatch):
    """GRAPHEM_HIP_GRAPH=1: iterations 2.. of a device-sampled run replayed from a hipGraph (ten per graph, the iteration
    number on the device) must give the bits of the enqueued loop, across runs of any length and after set_positions."""
    import graphem_rapids_amd as gra
    from graphem_rapids_amd import _native
    n = 40000
    edges = gra.random_regular_edges(n, 8, seed=2).astype(np.int32)
    pos = (np.random.default_rng(1).standard_normal((n, 3)) * 0.1).astype(np.float32)

    def sequence():
        eng = _native.Engine(n, 3, edges, 1.0, 0.2, 0.5, 10, 256, seed=5)
        eng.set_positions(pos)
        eng.run(37)
        eng.step(None)
        eng.run(12)
        a = eng.get_positions()
        eng.set_positions(pos)
        eng.run(25)
        b = eng.get_positions()
        eng.close()
        return a, b
    a0, b0 = sequence()
    monkeypatch.setenv("GRAPHEM_HIP_GRAPH", "1")
    a1, b1 = sequence()
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1)


def test_lean_sqrt_and_division_are_ieee():
    """common.h gh_sqrt_ieee / gh_div_by (the spring phase's square root and its D divisions by one distance, with the
    reciprocal shared) must be bit-identical to sqrtf and '/': 2^32 pseudo-random operand sets over and beyond the fast
    domain -- exponents across the range, zeros, denormals, values within 2 ulp of exact squares, |n| up to d."""
    from graphem_rapids_amd import _native
    for seed in (1, 2):
        bad_sqrt, bad_div = _native.selftest_arith(1 << 31, seed=seed)
        assert (bad_sqrt, bad_div) == (0, 0)


@pytest.mark.parametrize("iters", [1, 31, 32, 33, 64, 65, 130, 300])
def test_run_torch_sampled_equals_run_over_torch_randperm(iters):
    """gh_run_torch_sampled (ids drawn by the library's host thread from torch's mt19937 state, uploaded up to 32 rows at a
    time into a device ring of 128 rows, the host buffer holding 256) == gh_run over torch.randperm(E)[:S] drawn by torch
    itself: positions bit for bit, generator state identical afterwards -- below, at and past every ring length
    (pt.py:409, 808-833)."""
    import torch
    import graphem_rapids_amd as gra
    from graphem_rapids_amd import _native
    n, D, k, S = 30000, 3, 10, 256
    edges = gra.random_regular_edges(n, 8, seed=2).astype(np.int32)
    E = len(edges)
    pos = (np.random.default_rng(1).standard_normal((n, D)) * 0.1).astype(np.float32)
    a = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, knn_distance="cdist")
    b = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, knn_distance="cdist")
    a.set_positions(pos)
    b.set_positions(pos)
    torch.manual_seed(77)
    torch.rand(11)
    state = torch.get_rng_state().numpy().copy()
    ids = np.stack([torch.randperm(E)[:S].numpy() for _ in range(iters)]).astype(np.int32)
    a.run(iters, ids)
    b.run_torch_sampled(iters, state)
    assert np.array_equal(state, torch.get_rng_state().numpy())
    assert np.array_equal(a.get_positions(), b.get_positions())
    # a second call continues from the state it left (and reuses the ring)
    ids2 = np.stack([torch.randperm(E)[:S].numpy() for _ in range(5)]).astype(np.int32)
    a.run(5, ids2)
    b.run_torch_sampled(5, state)
    assert np.array_equal(state, torch.get_rng_state().numpy())
    assert np.array_equal(a.get_positions(), b.get_positions())
    a.close()
    b.close()


def test_run_torch_sampled_small_graph_f64_and_no_draw_cases():
    import torch
    import graphem_rapids_amd as gra
    from graphem_rapids_amd import _native
    # (1) a graph below the fused path's size (the per-phase kernels), float32 and float64 engines
    n, D, k, S = 300, 2, 5, 64
    edges = gra.random_regular_edges(n, 4, seed=3).astype(np.int32)
    E = len(edges)
    pos = (np.random.default_rng(2).standard_normal((n, D)) * 0.1)
    for dtype in ("float32", "float64"):
        a = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, dtype=dtype)
        b = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, dtype=dtype)
        a.set_positions(pos)
        b.set_positions(pos)
        torch.manual_seed(3)
        state = torch.get_rng_state().numpy().copy()
        ids = np.stack([torch.randperm(E)[:S].numpy() for _ in range(40)]).astype(np.int32)
        iters = 40 if dtype == "float32" else 3
        a.run(iters, ids[:iters])
        b.run_torch_sampled(iters, state)
        if dtype == "float32":
            assert np.array_equal(state, torch.get_rng_state().numpy())
            assert np.array_equal(a.get_positions(), b.get_positions())
        else:   # (the float64 engine's double atomics are not run-to-run reproducible: 1e-15 after one step)
            torch.manual_seed(3)
            for _ in range(iters):
                torch.randperm(E)
            assert np.array_equal(state, torch.get_rng_state().numpy())
            assert np.abs(a.get_positions() - b.get_positions()).max() <= 1e-9
        a.close()
        b.close()
    # (2) S >= E: the reference uses arange(E) and draws nothing (pt.py:412): the state stays as it was
    c = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, E + 5)
    c.set_positions(pos)
    state = torch.get_rng_state().numpy().copy()
    before = state.copy()
    c.run_torch_sampled(3, state)
    assert np.array_equal(state, before)
    # (3) not a generator state: ValueError, nothing ran
    d = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S)
    d.set_positions(pos)
    with pytest.raises(ValueError):
        d.run_torch_sampled(3, np.zeros(5056, dtype=np.uint8))
    assert np.array_equal(d.get_positions(), pos.astype(np.float32))
    c.close()
    d.close()


def test_public_run_layout_draws_like_the_reference_loop():
    """run_layout(sampler='torch') leaves torch's global generator where `iters` torch.randperm(E) calls leave it and
    gives the positions of the same run with the ids drawn by torch.randperm (pt.py:409)."""
    import torch
    import graphem_rapids_amd as gra
    adj = gra.generate_random_regular(20000, 8, seed=4)
    a = gra.create_graphem(adj, n_components=3, verbose=False, seed=11, init="random", sampler="torch")
    b = gra.create_graphem(adj, n_components=3, verbose=False, seed=11, init="random", sampler="torch")
    E, S = a.n_edges, a.sample_size
    torch.manual_seed(9)
    out = a.run_layout(70)
    after = torch.get_rng_state().clone()
    torch.manual_seed(9)
    ids = np.stack([torch.randperm(E)[:S].numpy() for _ in range(70)]).astype(np.int32)
    assert torch.equal(after, torch.get_rng_state())
    b._engine.run(70, ids)
    assert np.array_equal(out, b.get_positions())


def test_bench_line_contract_on_a_small_workload():
    """bench.py end to end (the driver's command on the quick workload): ONE JSON line with the contract's fields, the
    roofline and cpu_baseline objects, and this round's sub-records (parity_mode with its own roofline, public_api)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                          "--workload", "rr20k"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_cold", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_mode", "public_api"):
        assert key in d, key
    assert d["steps"] == 5 and d["warmup"] == 2 and d["n_gpus"] == 1 and d["config"]["workload"] == "rr20k"
    assert d["config"]["reference_identical"] is False and d["parity_mode"]["reference_identical"] is True
    for rf in (d["roofline"], d["parity_mode"]["roofline"]):
        assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1 and abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-9
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 1e-6
    for case in d["public_api"]["cases"].values():
        assert case["torch"]["reference_identical"] is True and case["device"]["reference_identical"] is False
        assert case["torch"]["ms_per_iteration"] > 0 and case["device"]["ms_per_iteration"] > 0
