"""CPU-only checks: the C-ABI library loads and exports every declared symbol, host-side logic
(generators, backend names, validation) works without a GPU, and the product never imports the oracle."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from graphem_rapids_amd import _native
    from graphem_rapids_amd import build as gra_build
    gra_build.build()
    header = open(os.path.join(ROOT, "include", "graphem_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(gh_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/graphem_hip.h but not exported"
    assert sorted(_native.SYMBOLS) == declared


def test_version_and_device_count_without_gpu():
    from graphem_rapids_amd import _native
    lib = _native.load()
    assert b"gfx950" in lib.gh_version()
    assert lib.gh_device_count() >= 0  # 0 here, no compute call


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "graphem-rapids_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), f
                assert "libgraphem_oracle" not in src, f
                assert not re.search(r"#include\s+[\"<][^\n]*oracle", src), f


def test_backend_names():
    import graphem_rapids_amd as gra
    cfg = gra.BackendConfig(n_vertices=10, n_components=2, force_backend="hip")
    assert gra.get_optimal_backend(cfg) == "hip"
    assert gra.get_optimal_backend(gra.BackendConfig(n_vertices=10)) == "hip"
    with pytest.raises(ValueError):
        gra.BackendConfig(n_vertices=10, force_backend="bogus")
    with pytest.raises(ValueError):
        gra.get_optimal_backend(gra.BackendConfig(n_vertices=10, force_backend="cuvs"))
    info = gra.get_backend_info()
    assert info["hip_library"] is True


def test_generators_erdos_renyi():
    import graphem_rapids_amd as gra
    e = gra.erdos_renyi_edges(2000, 0.01, seed=3)
    assert (e[:, 0] < e[:, 1]).all() and e.max() < 2000
    assert len(np.unique(e[:, 0] * 2000 + e[:, 1])) == len(e)
    expect = 2000 * 1999 / 2 * 0.01
    assert abs(len(e) - expect) < 5 * np.sqrt(expect)
    assert np.array_equal(e, gra.erdos_renyi_edges(2000, 0.01, seed=3))
    adj = gra.erdos_renyi_graph(300, 0.05, seed=1)
    assert adj.shape == (300, 300) and (adj != adj.T).nnz == 0 and adj.dtype in (np.int32, np.int64)
    assert len(gra.erdos_renyi_edges(10, 1.0)) == 45 and len(gra.erdos_renyi_edges(10, 0.0)) == 0


def test_generators_random_regular():
    import graphem_rapids_amd as gra
    for n, d in [(50, 4), (1000, 8), (31, 6)]:
        e = gra.random_regular_edges(n, d, seed=n)
        deg = np.bincount(e.ravel(), minlength=n)
        assert (deg == d).all() and (e[:, 0] < e[:, 1]).all()
        assert len(np.unique(e[:, 0] * n + e[:, 1])) == len(e) == n * d // 2
    with pytest.raises(ValueError):
        gra.random_regular_edges(5, 3)
    adj = gra.generate_random_regular(n=100, d=3, seed=0)
    assert (np.asarray(adj.sum(axis=1)).ravel() == 3).all()


def test_snap_edge_list_loader(tmp_path):
    import graphem_rapids_amd as gra
    p = tmp_path / "toy.txt"
    p.write_text("# comment\n# another\n10 20\n20 10\n20\t30\n30 30\n5 10\n")
    verts, edges = gra.load_snap_edge_list(str(p))
    assert len(verts) == 4
    assert edges.tolist() == [[0, 1], [1, 2], [2, 3]]


def test_edge_extraction_matches_reference_rule():
    """Upper triangle of the nonzero pattern as given, CSR row order (pt.py:235-240)."""
    import scipy.sparse as sp
    from graphem_rapids_amd.embedder_hip import GraphEmbedderHIP
    a = np.zeros((5, 5))
    a[0, 3] = 2.5
    a[3, 0] = 1
    a[1, 2] = 1      # only upper entry
    a[4, 2] = 1      # only lower entry: ignored
    a[2, 2] = 7      # self loop: ignored
    adj = GraphEmbedderHIP._validate_adjacency(a)
    assert sp.issparse(adj)

    class Dummy:
        verbose = False
    e = GraphEmbedderHIP._extract_edges_from_adjacency(Dummy(), adj)
    assert e.tolist() == [[0, 3], [1, 2]]
    with pytest.raises(ValueError):
        GraphEmbedderHIP._validate_adjacency(np.zeros((2, 3)))
    with pytest.raises(ValueError):
        GraphEmbedderHIP._validate_adjacency(np.zeros((0, 0)))


def test_hip_backend_fails_loudly_without_gpu():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible")
    import graphem_rapids_amd as gra
    with pytest.raises(RuntimeError):
        gra.create_graphem(gra.generate_random_regular(20, 4, 0), n_components=2, verbose=False, init="random")


def test_memory_utilities_keep_the_reference_interface():
    import gc
    import graphem_rapids_amd as gra
    calls = []
    orig = gc.collect
    gc.collect = lambda *a, **k: calls.append(1) or 0
    try:
        with gra.MemoryManager(cleanup_on_exit=True) as mm:
            assert mm.initial_memory is not None
        gra.cleanup_gpu_memory()

        @gra.monitor_memory_usage
        def f(x):
            return x + 1
        assert f(1) == 2
    finally:
        gc.collect = orig
    assert not calls, "the memory utilities must never call gc.collect()"
    assert gra.get_optimal_chunk_size(1234, 3) == 1234
    assert set(gra.get_gpu_memory_info()) >= {"available", "total", "free", "allocated", "cached"}
