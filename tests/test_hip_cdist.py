"""knn_distance='cdist': the HIP path must return the neighbour rows of the reference's PyTorch-CPU backend --
torch.cdist's fp32 matmul-form values ranked by torch.topk, ties in std::partial_sort's order (pt.py:580-583) --
id for id, at every size.  Checked against the reference's own rows (golden fixtures) and against
oracle/aten_cdist_topk.cpp, the CPU statement of ATen's algorithm that tests/test_oracle_reference_fullsize.py pins to
the reference at 100 K and 1 M vertices.  Needs a real MI355X (`pytest -m gpu`)."""
import numpy as np
import pytest

import oracle
from conftest import GOLDEN_CASES, load_golden

pytestmark = pytest.mark.gpu


def _engine(n, D, edges, k, S, params=(1.0, 0.2, 0.5), **kw):
    from graphem_rapids_amd import _native
    return _native.Engine(n, D, edges, *params, k, S, knn_distance="cdist", **kw)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_cdist_rows_are_the_references_on_the_small_fixtures(case):
    """Seven graphs of 6 .. 2000 vertices, every captured step: the rows torch.topk returned."""
    g = load_golden(case)
    n, D, k, S = int(g["n"]), int(g["D"]), int(g["k"]), int(g["S"])
    eng = _engine(n, D, g["edges"], k, S, params=tuple(float(x) for x in g["params"]))
    for t in g["steps"]:
        pos, sampled = g[f"pos_{t}"], g[f"sampled_{t}"]
        eng.set_positions(pos)
        knn = eng.knn_midpoints(sampled)
        full, unresolved = eng.knn_cdist_stats()
        assert np.array_equal(knn, oracle.knn_midpoints_aten(pos, g["edges"], sampled, k)), f"{case} step {t}"
        assert np.array_equal(knn, g[f"knn_{t}"]), f"{case} step {t}: not the reference's rows"
        assert unresolved == 0, f"{case} step {t}: {unresolved} rows with a tie that ATen's nth_element path decides"
        eng.step(sampled)
        assert np.abs(eng.get_positions() - g[f"pos_next_{t}"]).max() <= 1e-4
    eng.close()


def _positions(kind, n, D, rng):
    if kind == "gauss":
        return rng.standard_normal((n, D)).astype(np.float32)
    if kind == "start":       # the reference's random start: 0.1 sigma
        return (rng.standard_normal((n, D)) * 0.1).astype(np.float32)
    if kind == "lattice":     # coordinates on a coarse lattice: equal distances everywhere, ties in every row
        return (rng.integers(-6, 7, size=(n, D)) / 4.0).astype(np.float32)
    if kind == "lattice_fine":
        return (rng.integers(-40, 41, size=(n, D)) / 32.0).astype(np.float32)
    if kind == "collapsed":   # a bulk of radius 1e-3 far from the origin: cdist's quantum exceeds the neighbour spacing
        p = (rng.standard_normal((n, D)) * 1e-3 + 3.0).astype(np.float32)
        p[: n // 50] = rng.standard_normal((n // 50, D)).astype(np.float32) * 50.0
        return p
    raise ValueError(kind)


CASES = [
    # n, degree, D, k, S, positions
    (5000, 8, 3, 10, 256, "gauss"),
    (5000, 8, 3, 10, 256, "lattice"),
    (5000, 8, 2, 10, 256, "lattice_fine"),
    (5000, 8, 3, 10, 256, "collapsed"),
    (30000, 8, 3, 10, 256, "start"),
    (30000, 8, 3, 10, 700, "lattice_fine"),
    (30000, 8, 2, 5, 256, "gauss"),
    (20000, 8, 4, 10, 256, "gauss"),
    (20000, 8, 6, 32, 128, "lattice_fine"),
    (20000, 8, 16, 32, 256, "gauss"),
    (20000, 8, 12, 10, 256, "collapsed"),
    (6000, 6, 3, 100, 64, "gauss"),        # K = 101
    (6000, 6, 3, 126, 64, "lattice_fine"), # K + 1 = 128: the largest selection the scan path takes
    (6000, 6, 3, 200, 64, "gauss"),        # past it: every row takes the full pass
    (3000, 4, 3, 10, 256, "lattice"),      # E = 6000: too small for the scan, every row takes the full pass
    (3000, 4, 20, 10, 256, "gauss"),       # generic dimension
    (400, 4, 3, 10, 64, "lattice"),        # E = 800, K * 64 = 704 <= E: still partial_sort
]


@pytest.mark.parametrize("n,deg,D,k,S,kind", CASES)
def test_cdist_rows_equal_atens_on_random_graphs(n, deg, D, k, S, kind):
    import graphem_rapids_amd as gra
    rng = np.random.default_rng(n * 7 + D * 131 + k)
    edges = np.ascontiguousarray(gra.random_regular_edges(n, deg, seed=D + k), dtype=np.int32)
    E = len(edges)
    S = min(S, E)
    pos = _positions(kind, n, D, rng)
    eng = _engine(n, D, edges, k, S)
    eng.set_positions(pos)
    for rep in range(2):
        sampled = rng.permutation(E)[:S].astype(np.int32) if S < E else np.arange(E, dtype=np.int32)
        knn = eng.knn_midpoints(sampled)
        full, unresolved = eng.knn_cdist_stats()
        want = oracle.knn_midpoints_aten(pos, edges, sampled, k)
        bad = np.nonzero(~(knn == want).all(axis=1))[0]
        assert len(bad) == 0, (f"{len(bad)} rows differ (full-pass rows {full}); first: row {bad[0]}\n"
                               f"  hip  {knn[bad[0]]}\n  aten {want[bad[0]]}")
        assert unresolved == 0
        if kind.startswith("lattice"):
            assert full > 0          # the tie path was really taken
    # and one whole step with these rows: the oracle in its ATen mode
    eng.step(sampled)
    ref = oracle.step_aten(pos, edges, sampled, k)
    assert np.abs(eng.get_positions() - ref).max() <= 1e-4
    eng.close()


@pytest.mark.parametrize("k", list(range(1, 16)))
def test_scalar_register_heap_at_every_length(k):
    """K = k + 1 = 2 .. 16 keys: the replay's heap lives in pinned scalar registers and its step is the generated block of
    csrc/cdist_heap_asm.h (tools/gen_heap_asm.py), whose shape depends on the heap length (which nodes have two children,
    one, none).  Lattice positions put ties in every row; E = 6000 sends every row through the full pass (256 replays per
    search), a second graph goes through the filtered scan + prefix / tail replay."""
    import graphem_rapids_amd as gra
    for n, deg, D, kind, S in ((3000, 4, 3, "lattice", 256), (30000, 8, 2, "lattice_fine", 256)):
        rng = np.random.default_rng(1000 * k + D)
        edges = np.ascontiguousarray(gra.random_regular_edges(n, deg, seed=k), dtype=np.int32)
        E = len(edges)
        pos = _positions(kind, n, D, rng)
        eng = _engine(n, D, edges, k, S)
        eng.set_positions(pos)
        sampled = rng.permutation(E)[:S].astype(np.int32)
        knn = eng.knn_midpoints(sampled)
        full, unresolved = eng.knn_cdist_stats()
        want = oracle.knn_midpoints_aten(pos, edges, sampled, k)
        bad = np.nonzero(~(knn == want).all(axis=1))[0]
        assert len(bad) == 0, f"K = {k + 1}, {kind}: {len(bad)} rows differ; first: row {bad[0]}\n  hip  {knn[bad[0]]}\n  aten {want[bad[0]]}"
        assert unresolved == 0 and full > 0
        eng.close()


@pytest.mark.parametrize("n,deg,D,k,kind", [
    (120, 4, 3, 10, "gauss"),          # E = 240 < 64 * 11
    (120, 4, 3, 10, "lattice"),        # ties in every row: introselect's and introsort's order of equal values
    (300, 4, 2, 10, "lattice"),        # E = 600
    (200, 6, 3, 10, "lattice_fine"),
    (1500, 4, 3, 60, "lattice"),       # K = 61: E = 3000 < 3904; K - 1 > 16: the introsort loop partitions
    (3000, 4, 3, 100, "lattice_fine"), # K = 101: E = 6000 < 6464
    (50, 4, 16, 5, "lattice"),
])
def test_tiny_graphs_where_aten_ranks_with_nth_element(n, deg, D, k, kind):
    """K * 64 > E: ATen's topk takes std::nth_element + std::sort.  The engine replays libstdc++'s introselect and
    introsort on the row's (value, index) pairs (csrc/cdist.hip cdist_nth_kernel): rows with ties come out in ATen's
    order too (round 3 put them in id order and counted them)."""
    import graphem_rapids_amd as gra
    edges = np.ascontiguousarray(gra.random_regular_edges(n, deg, seed=1), dtype=np.int32)
    E = len(edges)
    assert (k + 1) * 64 > E
    rng = np.random.default_rng(n + k)
    pos = _positions(kind, n, D, rng)
    eng = _engine(n, D, edges, k, E)
    eng.set_positions(pos)
    ids = np.arange(E, dtype=np.int32)
    knn = eng.knn_midpoints(None)
    full, unresolved = eng.knn_cdist_stats()
    want = oracle.knn_midpoints_aten(pos, edges, ids, k)
    bad = np.nonzero(~(knn == want).all(axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} of {E} rows differ; first: row {bad[0]}\n  hip  {knn[bad[0]]}\n  aten {want[bad[0]]}"
    assert full == E and unresolved == 0
    eng.step(None)
    assert np.abs(eng.get_positions() - oracle.step_aten(pos, edges, ids, k)).max() <= 1e-4
    eng.close()


PART_CASES = [
    # world, n, D, k, S, positions
    (2, 30000, 3, 10, 256, "lattice_fine"),   # own edges >= 16384: the ranks scan, re-value and prove their lists
    (3, 30000, 3, 10, 256, "start"),
    (8, 30000, 3, 10, 128, "lattice_fine"),   # 15 K own edges: too few for the scan -- nothing proven, every row replayed
    (3, 50000, 2, 5, 256, "lattice_fine"),
    (2, 40000, 6, 20, 128, "lattice_fine"),   # K = 21: the lane heap
    (3, 5000, 3, 10, 64, "lattice"),
]


@pytest.mark.parametrize("world,n,D,k,S,kind", PART_CASES)
def test_cdist_on_row_partitions_gives_the_references_rows(world, n, D, k, S, kind):
    """knn_distance='cdist' with gh_partition: `world` engines on this GPU, the all-gather of the
    ranks' (S, k + 3) records emulated by a device copy.  The merged rows must be the single cdist engine's = ATen's
    (oracle.knn_midpoints_aten), row for row on every rank, ties included; one whole step must land on the oracle's
    ATen-mode step."""
    import torch
    import graphem_rapids_amd as gra
    from graphem_rapids_amd.distributed import HipShardEngine, partition_rows
    from graphem_rapids_amd import _native
    rng = np.random.default_rng(n + 17 * world + k)
    edges = np.ascontiguousarray(gra.random_regular_edges(n, 8, seed=D + k), dtype=np.int32)
    E = len(edges)
    pos = _positions(kind, n, D, rng)
    sampled = rng.permutation(E)[:S].astype(np.int32)
    want = oracle.knn_midpoints_aten(pos, edges, sampled, k)
    shards = []
    for r in range(world):
        chunk, lo, hi = partition_rows(n, world, r)
        sh = HipShardEngine(n, D, edges, 1.0, 0.2, 0.5, k, S, 0, (lo, hi, 0, 0, _native.EDGES_HASHED), 0, knn_distance="cdist")
        assert sh.key_cols == k + 3
        sh.rank_layout(world, r, chunk)
        sh.set_positions(pos)
        shards.append(sh)
    for sh in shards:
        sh.step_begin(sampled)
    gathered = torch.stack([sh.partial.clone() for sh in shards]).contiguous()
    for sh in shards:
        sh.step_merge(gathered, world)
    listed = []
    for r, sh in enumerate(shards):
        knn = sh.merged_knn()
        bad = np.nonzero(~(knn == want).all(axis=1))[0]
        assert len(bad) == 0, f"rank {r}: {len(bad)} rows differ; first: row {bad[0]}\n  hip  {knn[bad[0]]}\n  aten {want[bad[0]]}"
        full, unresolved = sh.eng.knn_cdist_stats()
        assert unresolved == 0
        listed.append(full)
    assert len(set(listed)) == 1          # every rank lists the same rows
    if kind.startswith("lattice"):
        assert listed[0] > 0              # the tie path was really taken
    stats_all = torch.stack([sh.stats.clone() for sh in shards]).contiguous()
    for sh in shards:
        sh.step_finish_own(stats_all)
    blocks = torch.stack([sh.pos_blocks[r].clone() for r, sh in enumerate(shards)])
    for sh in shards:
        sh.pos_blocks.copy_(blocks)
    torch.cuda.synchronize()
    ref = oracle.step_aten(pos, edges, sampled, k)
    outs = [sh.get_positions() for sh in shards]
    assert np.abs(outs[0] - ref).max() <= 1e-4
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])
    for sh in shards:
        sh.eng.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_native_partitioned_loop_in_parity_mode(world):
    """gh_run_partitioned (csrc/comm.hip) with knn_distance='cdist': `world` engines, one host thread each, collectives by
    the in-process loopback backend: three iterations from a lattice (ties in every row of the first one) must follow the
    single cdist engine, and every rank must end with identical bits."""
    import threading
    import graphem_rapids_amd as gra
    from graphem_rapids_amd import _native
    from graphem_rapids_amd.distributed import partition_rows
    n, D, k, S = 40000, 3, 10, 256
    edges = np.ascontiguousarray(gra.random_regular_edges(n, 8, seed=2), dtype=np.int32)
    rng = np.random.default_rng(world)
    pos = _positions("lattice_fine", n, D, rng)
    stream = np.stack([rng.permutation(len(edges))[:S] for _ in range(3)]).astype(np.int32)
    single = _engine(n, D, edges, k, S)
    single.set_positions(pos)
    single.run(3, stream)
    ref = single.get_positions()
    single.close()
    lib = _native.load()
    group = lib.gh_loopback_group_create(world)
    engines = []
    for r in range(world):
        chunk, lo, hi = partition_rows(n, world, r)
        e = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, knn_distance="cdist", partition=(lo, hi, 0, 0, _native.EDGES_HASHED))
        e.rank_layout(world, r, chunk)
        e.comm_init_loopback(group, r)
        e.set_positions(pos)
        engines.append(e)
    errors = []

    def work(e):
        try:
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
    for e in engines:
        e.comm_destroy()
        e.close()
    lib.gh_loopback_group_destroy(group)
    assert np.abs(outs[0] - ref).max() <= 2e-6
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


def test_public_api_defaults_to_cdist_with_the_torch_sampler():
    import torch
    import graphem_rapids_amd as gra
    n = 4000
    edges = gra.random_regular_edges(n, 8, seed=3).astype(np.int32)
    adj = gra.edges_to_adjacency(n, edges)
    emb = gra.create_graphem(adj, n_components=3, backend="hip", verbose=False, seed=0, init="random", sampler="torch")
    assert emb.knn_distance == "cdist"
    fast = gra.create_graphem(adj, n_components=3, backend="hip", verbose=False, seed=0, init="random", sampler="device")
    assert fast.knn_distance == "exact"
    pos = emb.positions
    torch.manual_seed(5)
    ids = torch.randperm(len(edges))[:256].numpy().astype(np.int32)
    torch.manual_seed(5)
    knn, used = emb._locate_knn_midpoints()
    assert np.array_equal(used, ids)
    assert np.array_equal(knn, oracle.knn_midpoints_aten(pos, emb._edges_np, ids, 10))
    torch.manual_seed(5)
    emb.update_positions()
    assert np.abs(emb.positions - oracle.step_aten(pos, emb._edges_np, ids, 10)).max() <= 1e-4


@pytest.mark.parametrize("n,deg,D,k,S,kind", [(30000, 8, 3, 10, 256, "lattice_fine"), (30000, 8, 2, 10, 256, "lattice"),
                                              (20000, 8, 6, 32, 128, "lattice_fine"), (30000, 8, 3, 10, 256, "start")])
def test_loop_replays_only_ties_that_can_change_a_force(n, deg, D, k, S, kind):
    """Inside the loop a knn_distance='cdist' engine lists a row only when values 0/1 tie (which id is dropped as column 0,
    pt.py:417-421) or values k/k+1 tie (which id is a member); a tie strictly inside permutes columns of the same pair set
    (pt.py:668-699).  Against the same loop with EVERY tie replayed (gh_set_cdist_replay(1): rows column for column):
    positions bit for bit, on 25 steps each from a fresh lattice state (ties in every row) and over a 50-iteration run from
    a lattice start -- while the number of replayed rows falls."""
    import graphem_rapids_amd as gra
    rng = np.random.default_rng(n + D * 17 + k)
    edges = np.ascontiguousarray(gra.random_regular_edges(n, deg, seed=D + k + 1), dtype=np.int32)
    E = len(edges)
    few = _engine(n, D, edges, k, S)
    every = _engine(n, D, edges, k, S)
    every.set_cdist_replay(True)
    listed_few = listed_every = 0
    for t in range(25):
        pos = _positions(kind, n, D, rng)
        sampled = rng.permutation(E)[:S].astype(np.int32)
        few.set_positions(pos)
        every.set_positions(pos)
        few.step(sampled)
        every.step(sampled)
        listed_few += few.knn_cdist_stats()[0]
        listed_every += every.knn_cdist_stats()[0]
        assert np.array_equal(few.get_positions(), every.get_positions()), f"step {t}"
    assert listed_few <= listed_every
    if kind.startswith("lattice"):
        assert listed_every >= 25                 # the tie path was taken
    if kind == "lattice_fine":
        assert listed_few < listed_every          # ... and the rule is in effect (on the coarse lattice every row has a tie at both ends)
    pos = _positions(kind, n, D, rng)
    stream = np.stack([rng.permutation(E)[:S] for _ in range(50)]).astype(np.int32)
    few.set_positions(pos)
    every.set_positions(pos)
    few.run(50, stream)
    every.run(50, stream)
    assert np.array_equal(few.get_positions(), every.get_positions())
    # the per-phase call is unchanged: every tie, the reference's rows column for column
    few.set_positions(pos)
    assert np.array_equal(few.knn_midpoints(stream[0]), oracle.knn_midpoints_aten(pos, edges, stream[0], k))
    few.close()
    every.close()
