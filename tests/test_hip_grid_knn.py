"""Sub-quadratic KNN: the grid search of csrc/grid_core.h against the exact filtered scan and
the oracle.  It is an exact method (every midpoint within the threshold ball lies in a visited cell -- for
n_components > 3 the cells are those of the projection onto three coordinates, which can only shorten a distance), so
"recall against the exact kernel" must be 1 and the ids identical, at S in {256, 1024, 4096} and beyond, for 2 to 16
components, for states with outliers outside the gridded cube and for a partitioned engine.  Needs a real MI355X."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _graph(n, deg, seed):
    import graphem_rapids_amd as gra
    return np.ascontiguousarray(gra.random_regular_edges(n, deg, seed=seed), dtype=np.int32)


@pytest.mark.parametrize("n,D,S,state", [
    (60000, 3, 256, "unit"), (60000, 3, 1024, "unit"), (60000, 3, 4096, "unit"),
    (60000, 2, 1024, "unit"),
    (60000, 3, 1024, "start"),       # the reference's random start: everything inside a few central cells
    (60000, 3, 1024, "outliers"),    # vertices far outside the gridded cube (border cells are unbounded)
    (200000, 3, 16384, "unit"),      # the regime the method is for
])
def test_grid_knn_equals_the_exact_scan_and_the_oracle(n, D, S, state):
    from graphem_rapids_amd import _native
    k = 10
    edges = _graph(n, 8, seed=3)
    rng = np.random.default_rng(5)
    pos = rng.standard_normal((n, D)).astype(np.float32)
    if state == "start":
        pos *= np.float32(0.1)
    elif state == "outliers":
        far = rng.permutation(n)[:200]
        pos[far] *= np.float32(40.0)
    sampled = rng.permutation(len(edges))[:S].astype(np.int32)
    if state == "outliers":
        sampled[:16] = np.nonzero(np.isin(edges[:, 0], far))[0][:16]
    ref = oracle.knn_midpoints(pos, edges, sampled, k)
    out = {}
    for method in ("grid", "scan"):
        eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, knn_method=method)
        eng.set_positions(pos)
        out[method] = eng.knn_midpoints(sampled)
        eng.step(sampled)
        out[method + "_pos"] = eng.get_positions()
        eng.close()
    assert np.array_equal(out["grid"], ref)         # recall 1.0, identical order
    assert np.array_equal(out["scan"], ref)
    # one whole step through either search: the same positions up to the order of the fp64 column sums
    assert np.abs(out["grid_pos"] - out["scan_pos"]).max() <= 2e-6
    assert np.abs(out["grid_pos"] - oracle.step(pos, edges, sampled, k)).max() <= 1e-4


def test_grid_path_is_taken_and_runs_a_layout():
    """The grid kernels really run (timing names), the device-sampled loop works on them, and the public class exposes
    the choice."""
    import graphem_rapids_amd as gra
    from graphem_rapids_amd import _native
    n, D, k, S = 80000, 3, 10, 2048
    edges = _graph(n, 8, seed=4)
    pos = (np.random.default_rng(0).standard_normal((n, D)) * 0.1).astype(np.float32)
    res = {}
    for method in ("grid", "scan"):
        eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=2, knn_method=method)
        eng.set_positions(pos)
        eng.timing_enable(True)
        eng.run(6)
        eng.sync()
        res[method] = (eng.get_positions(), set(eng.timings()))
        eng.close()
    assert {"grid_build", "grid_tau_scan"} <= res["grid"][1] and "spring_scan" not in res["grid"][1]
    assert "spring_scan" in res["scan"][1] and "grid_build" not in res["scan"][1]
    assert np.abs(res["grid"][0] - res["scan"][0]).max() <= 1e-4      # 6 steps, same samples (same seed), exact KNN both ways
    emb = gra.create_graphem(gra.edges_to_adjacency(n, edges), n_components=3, backend="hip", verbose=False, seed=0,
                             init="random", sample_size=S, knn_method="grid")
    out = emb.run_layout(3)
    assert out.shape == (n, 3) and np.isfinite(out).all()
    with pytest.raises(ValueError):
        gra.create_graphem(gra.edges_to_adjacency(n, edges), n_components=3, backend="hip", verbose=False, knn_method="kdtree")


def test_grid_knn_is_for_two_and_three_components_only():
    """With more components a grid over three coordinates stays exact but was measured 50-80x slower than the scan
    (profiles/r03/knn_method_sweep.log); round 4 removed that form: the engine then searches with the scan."""
    import graphem_rapids_amd as gra
    from graphem_rapids_amd import _native
    n, D, k, S = 4039, 16, 32, 1024
    edges = np.ascontiguousarray(gra.erdos_renyi_edges(n, 0.0108, seed=12345), dtype=np.int32)
    rng = np.random.default_rng(2)
    pos = rng.standard_normal((n, D)).astype(np.float32)
    sampled = rng.permutation(len(edges))[:S].astype(np.int32)
    eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, knn_method="grid")
    eng.set_positions(pos)
    eng.timing_enable(True)
    knn = eng.knn_midpoints(sampled)
    assert "grid_build" not in eng.timings()
    assert np.array_equal(knn, oracle.knn_midpoints(pos, edges, sampled, k))
    eng.close()


def test_grid_knn_on_row_partitions_and_auto_choice():
    """The grid indexes the OWN edges of a rank, like the scan: three row-partitioned engines in the native loop
    (loopback collectives, one thread each) with knn_method='grid' must reproduce the single engine; and AUTO picks an
    exact sub-quadratic search from thousands of sampled midpoints on (whole graph: the inverted file in its exact mode;
    a partitioned engine: the grid)."""
    import threading
    from graphem_rapids_amd import _native
    from graphem_rapids_amd.distributed import partition_rows
    n, D, k, S, world = 90001, 3, 10, 1024, 3
    edges = _graph(n - 1, 8, seed=6)
    rng = np.random.default_rng(9)
    pos = np.vstack([rng.standard_normal((n - 1, D)).astype(np.float32), np.zeros((1, D), np.float32)])
    stream = np.stack([rng.permutation(len(edges))[:S] for _ in range(3)]).astype(np.int32)
    single = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=4, knn_method="scan")
    single.set_positions(pos)
    single.run(3, stream)
    ref = single.get_positions()
    single.close()
    lib = _native.load()
    group = lib.gh_loopback_group_create(world)
    engines = []
    for r in range(world):
        chunk, lo, hi = partition_rows(n, world, r)
        e = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=4, partition=(lo, hi, 0, 0, _native.EDGES_HASHED),
                           knn_method="grid")
        e.gather_layout(world, r, chunk)
        e.comm_init_loopback(group, r)
        e.set_positions(pos)
        engines.append(e)
    errors = []

    def work(e):
        try:
            e.timing_enable(True)
            e.run_partitioned(3, stream)
            e.sync()
        except Exception as exc:  # pylint: disable=broad-exception-caught
            errors.append(exc)
    threads = [threading.Thread(target=work, args=(e,)) for e in engines]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors and not any(t.is_alive() for t in threads), errors
    outs = [e.get_positions() for e in engines]
    assert all("grid_build" in e.timings() for e in engines)
    for e in engines:
        e.comm_destroy()
        e.close()
    lib.gh_loopback_group_destroy(group)
    assert np.abs(outs[0] - ref).max() <= 2e-6
    assert all(np.array_equal(o, outs[0]) for o in outs[1:])
    # AUTO
    big = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, 16384, seed=1)
    big.set_positions(pos)
    big.timing_enable(True)
    big.run(1)
    big.sync()
    assert "ivf_scan" in big.timings() and big.knn_ivf_config() == (320, 320)     # exact mode: room for every list
    big.close()
    chunk, lo, hi = partition_rows(n, 2, 0)
    part = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, 16384, seed=1, partition=(lo, hi, 0, 0, _native.EDGES_HASHED))
    part.set_positions(pos)
    part.timing_enable(True)
    part.step_begin(rng.permutation(len(edges))[:16384].astype(np.int32))
    part.sync()
    assert "grid_build" in part.timings()
    part.close()
    small = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, 256, seed=1)
    small.set_positions(pos)
    small.timing_enable(True)
    small.run(1)
    small.sync()
    assert "spring_scan" in small.timings() and "grid_build" not in small.timings()
    small.close()
