"""Parity of the HIP path (through the C ABI) against the CPU oracle and the golden vectors
from the reference.  All tests here need a real MI355X: run with `pytest -m gpu`.

Tolerances:
  spring forces, KNN ids ........ bit-exact / identical (same arithmetic order as the oracle)
  intersection forces ........... rtol 1e-6 of max|F| (fp64 atomic sum vs fp32 sequential sum)
  integrate + normalise ......... atol 2e-6 (fp64 column statistics vs ATen's fp32 cascade)
  single step (P2) .............. max abs 1e-4 on unit-std positions
  5 iterations (P3) ............. max abs 1e-3
"""
import numpy as np
import pytest

import oracle
from conftest import GOLDEN_CASES, load_golden

pytestmark = pytest.mark.gpu


def _engine(g, seed=0):
    from graphem_rapids_amd import _native
    Lm, ka, ki = (float(x) for x in g["params"])
    return _native.Engine(int(g["n"]), int(g["D"]), g["edges"], Lm, ka, ki, int(g["k"]), int(g["S"]), seed=seed)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_phases_against_golden(case):
    g = load_golden(case)
    eng = _engine(g)
    Lm, ka, ki = (float(x) for x in g["params"])
    for t in g["steps"]:
        pos = g[f"pos_{t}"]
        eng.set_positions(pos)
        assert np.array_equal(eng.get_positions(), pos)
        # spring: same summation order and arithmetic as the reference -> bit-exact
        F = eng.spring_forces()
        assert np.array_equal(F, g[f"F_spring_{t}"]), f"{case} step {t}: max diff {np.abs(F - g[f'F_spring_{t}']).max()}"
        # knn: identical ids in identical order
        knn = eng.knn_midpoints(g[f"sampled_{t}"])
        assert np.array_equal(knn, g[f"knn_{t}"]), f"{case} step {t}"
        # intersection
        Fi = eng.intersection_forces(g[f"sampled_{t}"], g[f"knn_{t}"])
        ref = g[f"F_inter_{t}"]
        np.testing.assert_allclose(Fi, ref, rtol=1e-6, atol=1e-6 * max(1.0, float(np.abs(ref).max())))
        # integrate + normalise
        out = eng.integrate_normalise(g[f"F_spring_{t}"], g[f"F_inter_{t}"])
        np.testing.assert_allclose(out, g[f"pos_next_{t}"], rtol=0, atol=2e-6)
        assert np.array_equal(eng.get_positions(), pos)  # per-phase calls leave the state alone
    eng.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_single_step_P2(case):
    g = load_golden(case)
    eng = _engine(g)
    for t in g["steps"]:
        eng.set_positions(g[f"pos_{t}"])
        eng.step(g[f"sampled_{t}"])
        out = eng.get_positions()
        err = np.abs(out - g[f"pos_next_{t}"]).max()
        assert err <= 1e-4, f"{case} step {t}: {err}"
    eng.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_short_horizon_P3(case):
    g = load_golden(case)
    Lm, ka, ki = (float(x) for x in g["params"])
    iters = min(5, g["sample_stream"].shape[0])
    eng = _engine(g)
    eng.set_positions(g["p0"])
    eng.run(iters, g["sample_stream"][:iters])
    out = eng.get_positions()
    ref = oracle.run_layout(g["p0"], g["edges"], g["sample_stream"][:iters], int(g["k"]), Lm, ka, ki, colmajor=True)
    assert np.abs(out - ref).max() <= 1e-3
    eng.close()


def test_long_horizon_invariants_P4():
    """50 iterations of C1: chaotic beyond ~10 iterations, so invariants and distributions only."""
    from scipy import stats
    g = load_golden("c1_er1000")
    eng = _engine(g)
    eng.set_positions(g["p0"])
    eng.run(50, g["sample_stream"])
    out = eng.get_positions()
    assert np.isfinite(out).all()
    assert np.abs(out.astype(np.float64).mean(0)).max() < 1e-5
    np.testing.assert_allclose(out.astype(np.float64).std(0, ddof=1), 1.0, atol=1e-4)
    assert np.abs(out).max() < 1000
    ref = g["pos_final"]
    e = g["edges"]
    len_hip = np.linalg.norm(out[e[:, 0]] - out[e[:, 1]], axis=1)
    len_ref = np.linalg.norm(ref[e[:, 0]] - ref[e[:, 1]], axis=1)
    assert stats.ks_2samp(len_hip, len_ref).statistic < 0.1
    rho = stats.spearmanr(np.linalg.norm(out, axis=1), np.linalg.norm(ref, axis=1)).statistic
    assert rho > 0.5
    eng.close()


def test_run_equals_steps_and_is_deterministic():
    g = load_golden("c1_er1000")
    a, b = _engine(g), _engine(g)
    a.set_positions(g["p0"])
    b.set_positions(g["p0"])
    a.run(8, g["sample_stream"][:8])
    for t in range(8):
        b.step(g["sample_stream"][t])
    assert np.array_equal(a.get_positions(), b.get_positions())
    a.set_positions(g["p0"])
    a.run(8, g["sample_stream"][:8])
    assert np.array_equal(a.get_positions(), b.get_positions())
    a.close(), b.close()


def test_topk_error_and_no_sampling_cases():
    g = load_golden("two_triangles")  # E = 6
    from graphem_rapids_amd import _native
    eng = _native.Engine(6, 2, g["edges"], 1.0, 0.2, 0.5, 6, 6)  # k+1 = 7 > E
    eng.set_positions(g["p0"])
    with pytest.raises(RuntimeError):
        eng.step()
    eng.close()
    eng = _engine(g)  # S >= E: arange, no sampler
    eng.set_positions(g["p0"])
    eng.step()  # no ids needed
    np.testing.assert_allclose(eng.get_positions(), g["pos_next_0"], atol=1e-5)
    eng.close()


def test_invalid_arguments():
    from graphem_rapids_amd import _native
    e = np.array([[0, 1], [1, 2]], dtype=np.int32)
    with pytest.raises(ValueError):
        _native.Engine(3, 0, e, 1.0, 0.2, 0.5, 1, 2)
    with pytest.raises(ValueError):
        _native.Engine(3, 2, e, 1.0, -0.2, 0.5, 1, 2)
    with pytest.raises(ValueError):
        _native.Engine(3, 2, np.array([[0, 7]], dtype=np.int32), 1.0, 0.2, 0.5, 1, 2)
    with pytest.raises(RuntimeError):
        _native.Engine(3, 2, e, 1.0, 0.2, 0.5, 1, 2, device_id=99)
    eng = _native.Engine(3, 2, e, 1.0, 0.2, 0.5, 1, 1)
    with pytest.raises(ValueError):
        eng.set_positions(np.zeros((4, 2), np.float32))
    with pytest.raises(ValueError):
        eng.step(np.array([5], dtype=np.int32))
    eng.close()


def _random_case(n, D, deg, k, S, seed):
    import graphem_rapids_amd as gra
    rng = np.random.default_rng(seed)
    edges = gra.random_regular_edges(n, deg, seed).astype(np.int32)
    pos = rng.standard_normal((n, D)).astype(np.float32)
    sampled = rng.permutation(len(edges))[:S].astype(np.int32)
    return edges, pos, sampled


@pytest.mark.parametrize("n,D,deg,k,S", [
    (20000, 3, 8, 10, 256),     # scan path, one level
    (100000, 3, 8, 10, 256),    # BASELINE config 2: RR n=100K d=8
    (30000, 2, 6, 10, 256),
    (30000, 4, 6, 7, 100),
    (12000, 8, 8, 12, 64),      # LD = 8 template
    (12000, 6, 8, 12, 64),      # D=6 padded into the LD=8 template
    (10000, 16, 8, 32, 256),    # BASELINE config 5 shape: D=16, k=32
    (6000, 12, 8, 20, 300),     # D=12 padded into LD=16, S > 256
    (3000, 5, 4, 10, 256),      # below the scan threshold: per-query kernel, generic spring
    (2000, 20, 4, 10, 64),      # generic D
    (900, 40, 6, 40, 128),      # generic D, large k
    (20000, 3, 8, 100, 256),    # K = 101: still the scan path (extraction up to 128 keys)
    (20000, 3, 8, 127, 64),     # K = 128: its upper end
    (20000, 3, 8, 130, 64),     # K = 131: the per-query sort kernel
])
def test_random_graphs_against_oracle(n, D, deg, k, S):
    from graphem_rapids_amd import _native
    edges, pos, sampled = _random_case(n, D, deg, k, S, seed=n + D)
    eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S)
    eng.set_positions(pos)
    assert np.array_equal(eng.spring_forces(), oracle.spring_forces(pos, edges, 1.0, 0.2))
    knn = eng.knn_midpoints(sampled)
    ref_knn = oracle.knn_midpoints(pos, edges, sampled, k)
    assert np.array_equal(knn, ref_knn)
    Fi = eng.intersection_forces(sampled, ref_knn)
    ref = oracle.intersection_forces(pos, edges, sampled, ref_knn, 0.5)
    np.testing.assert_allclose(Fi, ref, rtol=1e-6, atol=1e-6 * max(1.0, float(np.abs(ref).max())))
    eng.step(sampled)
    out = eng.get_positions()
    ref_next = oracle.step(pos, edges, sampled, k, 1.0, 0.2, 0.5)
    assert np.abs(out - ref_next).max() <= 1e-4
    eng.close()


def test_full_size_er_1m_knn_and_step():
    """BASELINE config 3: ER n=1M p=1e-5 (E ~ 5M), D=3, k=10, S=256 -- exact KNN ids and one step."""
    import graphem_rapids_amd as gra
    from graphem_rapids_amd import _native
    n = 1_000_000
    edges = gra.erdos_renyi_edges(n, 1e-5, 12345).astype(np.int32)
    rng = np.random.default_rng(7)
    pos = (rng.standard_normal((n, 3)) * 0.1).astype(np.float32)
    sampled = rng.permutation(len(edges))[:256].astype(np.int32)
    eng = _native.Engine(n, 3, edges, 1.0, 0.2, 0.5, 10, 256)
    eng.set_positions(pos)
    knn = eng.knn_midpoints(sampled)
    assert np.array_equal(knn, oracle.knn_midpoints(pos, edges, sampled, 10))
    assert np.array_equal(eng.spring_forces(), oracle.spring_forces(pos, edges, 1.0, 0.2))
    eng.step(sampled)
    out = eng.get_positions()
    assert np.abs(out - oracle.step(pos, edges, sampled, 10, 1.0, 0.2, 0.5)).max() <= 1e-4
    # size-independent invariants after a few more device-sampled iterations
    eng.run(5)
    out = eng.get_positions()
    o64 = out.astype(np.float64)  # numpy's float32 reductions are themselves off by 1e-3 at this size
    assert np.isfinite(out).all() and np.abs(o64.mean(0)).max() < 1e-5
    np.testing.assert_allclose(o64.std(0, ddof=1), 1.0, atol=1e-4)
    eng.close()


def test_degenerate_positions_force_the_overflow_fallback():
    """All vertices at one point: every distance ties at 0, every candidate list overflows, and the
    exact per-query fallback must return the smallest ids (tie-break) like the oracle."""
    from graphem_rapids_amd import _native
    edges, pos, sampled = _random_case(20000, 3, 8, 10, 64, seed=5)
    pos[:] = 0.25
    eng = _native.Engine(20000, 3, edges, 1.0, 0.2, 0.5, 10, 64)
    eng.set_positions(pos)
    assert np.array_equal(eng.knn_midpoints(sampled), oracle.knn_midpoints(pos, edges, sampled, 10))
    # clustered: half of the vertices coincide
    pos = np.random.default_rng(1).standard_normal((20000, 3)).astype(np.float32)
    pos[::2] = pos[0]
    eng.set_positions(pos)
    assert np.array_equal(eng.knn_midpoints(sampled), oracle.knn_midpoints(pos, edges, sampled, 10))
    eng.step(sampled)
    assert np.isfinite(eng.get_positions()).all()
    eng.close()


def test_device_sampler():
    from graphem_rapids_amd import _native
    import ctypes
    edges, pos, _ = _random_case(5000, 3, 8, 10, 256, seed=9)
    E = len(edges)

    def draw(seed, iters):
        eng = _native.Engine(5000, 3, edges, 1.0, 0.2, 0.5, 10, 256, seed=seed)
        eng.set_positions(pos)
        eng.run(iters)  # device sampler
        out = eng.get_positions()
        eng.close()
        return out
    a, b, c = draw(1, 3), draw(1, 3), draw(2, 3)
    assert np.array_equal(a, b)          # same seed -> same trajectory
    assert not np.array_equal(a, c)      # different seed -> different samples
    assert np.isfinite(a).all()


@pytest.mark.parametrize("world,rule,n,D", [(2, "range", 30011, 3), (3, "range", 30011, 3),
                                            (2, "hashed", 30011, 3), (3, "hashed", 30011, 3),
                                            (3, "hashed", 2001, 3),     # few edges: per-query search of d_mid
                                            (2, "hashed", 9001, 5),     # D without a templated spring kernel
                                            (3, "hashed", 20001, 16),
                                            (3, "hashed-hubs", 30011, 3)])  # rows with thousands of neighbours
@pytest.mark.parametrize("finish", ["own", "gathered", "overlap"])
def test_partitioned_engines_equal_single_engine(world, rule, n, D, finish):
    """The split step (gh_step_begin / gh_step_merge / gh_step_finish) with row partitions: `world`
    engines on ONE GPU, collectives emulated with device copies, must reproduce the unpartitioned
    engine (the oracle of the multi-GPU mode is the 1-GPU result), under both edge
    ownership rules of gh_partition."""
    import torch
    from graphem_rapids_amd import _native
    from graphem_rapids_amd.distributed import HipShardEngine, owned_edge_ids, partition_edges, partition_rows
    k, S = 10, 256
    edges, pos, _ = _random_case(n - 1, D, 8, k, S, seed=21)  # last vertex isolated
    if rule == "hashed-hubs":   # two hubs (in different row blocks) on top, and an edge between them
        rule = "hashed"
        rngh = np.random.default_rng(3)
        extra = [np.array([[100, n - 50]])]
        for hub, deg in ((100, 5000), (n - 50, 700)):
            nb = rngh.choice(n - 1, size=deg, replace=False)
            nb = nb[nb != hub]
            extra.append(np.stack([np.minimum(hub, nb), np.maximum(hub, nb)], axis=1))
        edges = np.ascontiguousarray(np.unique(np.concatenate([np.sort(edges.astype(np.int64), axis=1)] + extra), axis=0),
                                     dtype=np.int32)
    pos = np.vstack([pos, np.zeros((1, D), np.float32)])
    rng = np.random.default_rng(4)
    stream = np.stack([rng.permutation(len(edges))[:S] for _ in range(3)]).astype(np.int32)
    single = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S)
    single.set_positions(pos)
    single.run(3, stream)
    ref = single.get_positions()
    single.close()

    shards, parts, owned = [], [], 0
    for r in range(world):
        chunk, lo, hi = partition_rows(n, world, r)
        if rule == "hashed":
            part = (lo, hi, 0, 0, _native.EDGES_HASHED)
            owned += len(owned_edge_ids(edges, lo, hi, n))
        else:
            elo, ehi = partition_edges(edges, lo, hi)
            part = (lo, hi, elo, ehi, _native.EDGES_RANGE)
            owned += ehi - elo
        parts.append((lo, hi))
        shards.append(HipShardEngine(n, D, edges, 1.0, 0.2, 0.5, k, S, 0, part, 0))
        if finish == "own":
            shards[-1].rank_layout(world, r, chunk, packed=True)   # (the unpadded block exchange, on by default from 2 M vertices)
        elif finish == "overlap":
            shards[-1].overlap_layout(world, r, chunk)
        else:
            shards[-1].gather_layout(world, r, chunk)
        shards[-1].set_positions(pos)
    assert owned == len(edges)  # every edge searched by exactly one rank
    for t in range(3):
        for sh in shards:
            sh.step_begin(stream[t])
        if finish == "overlap":
            # form D: the new0 blocks travel FIRST, then keys, merge, statistics + patch lists (the owners' finished touched
            # rows); every rank writes the patch lists over the gathered rows and normalises all n rows
            def exchange_rows():
                for sh in shards:
                    sh.step_pack_rows()
                rows = torch.stack([sh.rows_all[r].clone() for r, sh in enumerate(shards)])
                for sh in shards:
                    sh.rows_all.copy_(rows)
            early = shards[0].step_rows_early()
            assert all(sh.step_rows_early() == early for sh in shards)
            assert early      # (a rank without the fused kernel -- the 2001-vertex graph -- makes new0 with a launch of its own)
            if early:
                exchange_rows()
            gathered = torch.stack([sh.partial.clone() for sh in shards]).contiguous()
            for sh in shards:
                sh.step_merge(gathered, world)
            if not early:
                exchange_rows()
            stats = torch.stack([sh.stats_all[r].clone() for r, sh in enumerate(shards)])
            for sh in shards:
                sh.stats_all.copy_(stats)
                sh.step_finish_overlap()
            continue
        gathered = torch.stack([sh.partial.clone() for sh in shards]).contiguous()
        for sh in shards:
            sh.step_merge(gathered, world)
        if finish == "own":   # all-gather of the statistics, own rows normalised, in-place all-gather of the blocks
            stats_all = torch.stack([sh.stats.clone() for sh in shards]).contiguous()
            for sh in shards:
                sh.step_finish_own(stats_all)
            if shards[0].packed_blocks is not None:   # D < ld: the blocks travel without their pad columns, then are expanded
                assert D < shards[0].ld
                packed = torch.stack([sh.packed_blocks[r].clone() for r, sh in enumerate(shards)])
                for sh in shards:
                    sh.packed_blocks.copy_(packed)
                    sh.step_unpack_rows()
                continue
            assert D == shards[0].ld
            blocks = torch.stack([sh.pos_blocks[r].clone() for r, sh in enumerate(shards)])
            for sh in shards:
                sh.pos_blocks.copy_(blocks)
            continue
        slots = torch.stack([sh.gbuf[r].clone() for r, sh in enumerate(shards)])   # the all-gather of the slots
        for sh in shards:
            sh.gbuf.copy_(slots)
            sh.step_finish_gathered()
    torch.cuda.synchronize()
    tol = 2e-6
    for sh in shards:
        got = sh.get_positions()
        assert np.abs(got - ref).max() <= tol
        assert np.array_equal(got, shards[0].get_positions())


def test_unsorted_edge_list_takes_the_gather_path():
    """An edge list that is not sorted by first endpoint cannot use the fused spring+midpoint
    kernel; results must not change (only the edge ids are permuted)."""
    from graphem_rapids_amd import _native
    edges, pos, sampled = _random_case(20000, 3, 8, 10, 256, seed=33)
    perm = np.random.default_rng(0).permutation(len(edges))
    shuffled = np.ascontiguousarray(edges[perm])
    eng = _native.Engine(20000, 3, shuffled, 1.0, 0.2, 0.5, 10, 256)
    eng.set_positions(pos)
    assert np.array_equal(eng.knn_midpoints(sampled), oracle.knn_midpoints(pos, shuffled, sampled, 10))
    assert np.array_equal(eng.spring_forces(), oracle.spring_forces(pos, shuffled, 1.0, 0.2))
    eng.step(sampled)
    assert np.abs(eng.get_positions() - oracle.step(pos, shuffled, sampled, 10)).max() <= 1e-4
    eng.close()


def test_rccl_driver_single_rank():
    """The real multi-GPU driver (PartitionedLayout + HipShardEngine) with world size 1 on this GPU, both ways:
    the loop in the C library with ncclAllGather on the library's own RCCL communicator (gh_run_partitioned), and
    the Python-driven step with torch.distributed 'nccl' collectives on torch's stream.  Must equal the plain engine."""
    import os
    import torch
    import torch.distributed as dist
    from graphem_rapids_amd import _native
    from graphem_rapids_amd.distributed import PartitionedLayout
    n, D, k, S = 25000, 3, 10, 256
    edges, pos, _ = _random_case(n, D, 8, k, S, seed=77)
    rng = np.random.default_rng(5)
    stream = np.stack([rng.permutation(len(edges))[:S] for _ in range(3)]).astype(np.int32)
    ref_eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=9)
    ref_eng.set_positions(pos)
    ref_eng.run(3, stream)
    ref_eng.run(2)           # device sampler, iterations 3 and 4
    ref = ref_eng.get_positions()
    ref_eng.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29577"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        got = {}
        for native, finish in ((True, "own"), (False, "own"), (True, "gathered"), (False, "gathered"), (True, "overlap"), (False, "overlap")):
            # the loop in the C library over the library's own RCCL communicator / driven from Python; both finishes
            lay = PartitionedLayout(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=9, rank=0, world=1, device_id=0, native=native,
                                    finish=finish)
            assert lay.native == native
            lay.set_positions(pos)
            lay.run(3, stream)
            lay.run(2)
            lay.sync()
            torch.cuda.synchronize()
            got[native, finish] = lay.get_positions()
            lay.engine.eng.close()
        assert PartitionedLayout(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=9, rank=0, world=1, device_id=0).native is False  # opt-in
    finally:
        dist.destroy_process_group()
    for key, val in got.items():
        assert np.abs(val - ref).max() <= 2e-6, key
    assert np.array_equal(got[True, "own"], got[False, "own"])
    assert np.array_equal(got[True, "gathered"], got[False, "gathered"])
    assert np.array_equal(got[True, "overlap"], got[False, "overlap"])


@pytest.mark.parametrize("world,n,D,finish", [(2, 30011, 3, "own"), (3, 30011, 3, "own"), (4, 100003, 3, "own"), (3, 9001, 5, "own"),
                                              (2, 20001, 16, "own"), (3, 30011, 3, "gathered"), (8, 100003, 3, "own"),
                                              (2, 30011, 3, "overlap"), (3, 30011, 3, "overlap"), (8, 100003, 3, "overlap"),
                                              (3, 9001, 5, "overlap"), (2, 20001, 16, "overlap"), (3, 2001, 3, "overlap")])
def test_native_partitioned_loop_on_the_loopback_backend(world, n, D, finish):
    """gh_run_partitioned (csrc/comm.hip): the whole multi-rank run inside the C library.  `world` engines on this one
    GPU, one host thread each, collectives by the in-process loopback backend (RCCL refuses two ranks on one device):
    the same loop, the same buffers and counts as over RCCL.  Must reproduce the single engine, with a host id stream
    and with the device sampler, and every rank must end with identical bits."""
    import threading
    from graphem_rapids_amd import _native
    from graphem_rapids_amd.distributed import partition_rows
    k, S = 10, 256
    edges, pos, _ = _random_case(n - 1, D, 8, k, S, seed=31)
    pos = np.vstack([pos, np.zeros((1, D), np.float32)])
    rng = np.random.default_rng(8)
    stream = np.stack([rng.permutation(len(edges))[:S] for _ in range(3)]).astype(np.int32)
    single = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=4)
    single.set_positions(pos)
    single.run(3, stream)
    single.run(2)
    ref = single.get_positions()
    single.close()

    lib = _native.load()
    group = lib.gh_loopback_group_create(world)
    engines = []
    for r in range(world):
        chunk, lo, hi = partition_rows(n, world, r)
        e = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=4, partition=(lo, hi, 0, 0, _native.EDGES_HASHED))
        if finish == "own":
            e.rank_layout(world, r, chunk)
            if D < e.ld:   # the unpadded block exchange (default from 2 M vertices on): forced on for the even world sizes
                e.set_packed_rows(world % 2 == 0)
        elif finish == "overlap":   # form D: the new0 blocks on the side stream, first
            e.overlap_layout(world, r, chunk)
        else:
            e.gather_layout(world, r, chunk)
        e.comm_init_loopback(group, r)
        e.set_positions(pos)
        engines.append(e)
    errors = []

    def work(e):
        try:
            e.run_partitioned(3, stream)
            e.run_partitioned(2)
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


def test_large_k_takes_the_sort_kernel():
    """n_neighbors + 1 > 64 selects with the LDS sort kernel instead of block-min extraction."""
    from graphem_rapids_amd import _native
    edges, pos, sampled = _random_case(3000, 3, 8, 80, 50, seed=12)
    eng = _native.Engine(3000, 3, edges, 1.0, 0.2, 0.5, 80, 50)
    eng.set_positions(pos)
    assert np.array_equal(eng.knn_midpoints(sampled), oracle.knn_midpoints(pos, edges, sampled, 80))
    eng.step(sampled)
    assert np.abs(eng.get_positions() - oracle.step(pos, edges, sampled, 80)).max() <= 1e-4
    eng.close()
    # and on a graph large enough for the scan path, where K > 64 must fall back to the per-query kernel
    edges, pos, sampled = _random_case(20000, 3, 8, 70, 32, seed=13)
    eng = _native.Engine(20000, 3, edges, 1.0, 0.2, 0.5, 70, 32)
    eng.set_positions(pos)
    assert np.array_equal(eng.knn_midpoints(sampled), oracle.knn_midpoints(pos, edges, sampled, 70))
    eng.close()


def test_many_queries_and_no_sampling_on_the_scan_path():
    """S > 256 (several query groups inside the fused kernel) and S >= E (arange, pt.py:412)."""
    from graphem_rapids_amd import _native
    edges, pos, _ = _random_case(20000, 3, 8, 10, 700, seed=14)
    sampled = np.random.default_rng(2).permutation(len(edges))[:700].astype(np.int32)
    eng = _native.Engine(20000, 3, edges, 1.0, 0.2, 0.5, 10, 700)
    eng.set_positions(pos)
    assert np.array_equal(eng.knn_midpoints(sampled), oracle.knn_midpoints(pos, edges, sampled, 10))
    eng.step(sampled)
    assert np.abs(eng.get_positions() - oracle.step(pos, edges, sampled, 10)).max() <= 1e-4
    eng.close()
    edges, pos, _ = _random_case(5000, 2, 8, 6, 10 ** 6, seed=15)   # sample_size >= E = 20000
    eng = _native.Engine(5000, 2, edges, 1.0, 0.2, 0.5, 6, 10 ** 6)
    eng.set_positions(pos)
    eng.step()
    allq = np.arange(len(edges), dtype=np.int32)
    assert np.abs(eng.get_positions() - oracle.step(pos, edges, allq, 6)).max() <= 1e-4
    eng.close()


def test_zero_neighbours_or_zero_samples_run_spring_only():
    """n_neighbors = 0 or sample_size = 0: no KNN / intersection phase, the step is spring + normalise."""
    from graphem_rapids_amd import _native
    edges, pos, _ = _random_case(4000, 3, 6, 10, 64, seed=3)
    zero = np.zeros_like(pos)
    want = oracle.integrate_normalise(pos, oracle.spring_forces(pos, edges, 1.0, 0.2), zero)
    for k, S in [(0, 64), (10, 0)]:
        eng = _native.Engine(4000, 3, edges, 1.0, 0.2, 0.5, k, S)
        eng.set_positions(pos)
        eng.step()
        eng.run(1)
        eng.set_positions(pos)
        eng.step()
        assert np.abs(eng.get_positions() - want).max() <= 2e-6
        eng.close()


def test_full_size_rr_1m_bench_workload():
    """The benchmark workload itself (random-regular n = 1 M, d = 8, E = 4 M): exact KNN ids, exact spring
    forces and one step against the oracle; then the size-independent invariants after device-sampled steps."""
    import graphem_rapids_amd as gra
    from graphem_rapids_amd import _native
    n = 1_000_000
    edges = gra.random_regular_edges(n, 8, seed=0).astype(np.int32)
    rng = np.random.default_rng(0)
    pos = (rng.standard_normal((n, 3)) * 0.1).astype(np.float32)
    sampled = rng.permutation(len(edges))[:256].astype(np.int32)
    eng = _native.Engine(n, 3, edges, 1.0, 0.2, 0.5, 10, 256)
    eng.set_positions(pos)
    assert np.array_equal(eng.knn_midpoints(sampled), oracle.knn_midpoints(pos, edges, sampled, 10))
    sub, fin, ovf = eng.knn_last_counts()
    assert ovf.sum() == 0 and fin.max() < 8192          # the filtered scan, not the fallback, produced this
    assert np.array_equal(eng.spring_forces(), oracle.spring_forces(pos, edges, 1.0, 0.2))
    eng.step(sampled)
    assert np.abs(eng.get_positions() - oracle.step(pos, edges, sampled, 10)).max() <= 1e-4
    eng.run(10)
    o64 = eng.get_positions().astype(np.float64)
    assert np.isfinite(o64).all() and np.abs(o64.mean(0)).max() < 1e-5
    np.testing.assert_allclose(o64.std(0, ddof=1), 1.0, atol=1e-4)
    eng.close()


def _fused_step_keys(n, D, edges, pos, sampled, k):
    """Keys (dist2 bits << 32 | edge id) of the KNN phase as the fused spring+scan step computes them."""
    from graphem_rapids_amd.distributed import HipShardEngine
    sh = HipShardEngine(n, D, edges, 1.0, 0.2, 0.5, k, len(sampled), 0, (0, n, 0, len(edges), 0), 0)
    sh.set_positions(pos)
    sh.step_begin(sampled)
    sh.sync()
    keys = sh.partial.cpu().numpy().copy()
    sh.eng.close()
    return keys


@pytest.mark.parametrize("n,D,deg,outliers", [
    (50000, 3, 8, "none"),       # tiles of 512 edges
    (400000, 3, 8, "none"),
    (50000, 2, 8, "none"),
    (50000, 3, 8, "some"),       # 1% of the vertices far outside the f16 range of the MFMA filter
    (5000, 3, 8, "all"),         # every coordinate outside it: exact scans only
    (50000, 3, 8, "tiny"),       # coordinates ~1e-4: distances near the absolute slack of the filter
    (60000, 3, 8, "S1100"),      # 1100 queries: several query groups, the last one ragged (MFMA is the default form here)
    # wide rows: one f16 piece per coordinate, contraction 16 deep (D <= 10) or 32 deep
    (30000, 4, 8, "none"),       # 4-float rows
    (30000, 4, 8, "some"),
    (30000, 5, 8, "none"),
    (30000, 8, 8, "none"),
    (30000, 8, 8, "some"),
    (30000, 10, 8, "none"),      # 16-deep contraction on 16-float rows
    (30000, 11, 8, "none"),      # 32-deep
    (30000, 16, 8, "none"),
    (30000, 16, 8, "some"),
    (5000, 16, 8, "all"),
    (30000, 6, 8, "tiny"),       # coordinates ~1e-4: f16 subnormals, the absolute slack of the wide form
    (30000, 12, 8, "S1100"),
    (4000, 16, 44, "none"),      # C5 shape: few long workgroups, query slices over blockIdx.y, every row on the long path
])
def test_fused_scan_knn_is_exact(n, D, deg, outliers):
    """KNN of the fused spring+scan kernel (read back after gh_step_begin) against the oracle: the pre-filter on the
    f16 matrix pipe (split operands for D <= 3, single-piece operands for D >= 4) is conservative and the decision
    exact, so ids AND distance bits must be identical.  (The packed-fp32 VALU form of the fused kernel, until round 3
    reachable through GRAPHEM_HIP_MFMA=0, was removed in round 4; the stand-alone scan kernel still uses that filter.)"""
    k, S = 10, (1100 if outliers == "S1100" else 256)
    edges, pos, sampled = _random_case(n, D, deg, k, S, seed=101)
    rng = np.random.default_rng(7)
    if outliers == "some":
        far = rng.permutation(n)[: n // 100]
        pos[far] *= np.float32(300.0)
        sampled[:8] = np.nonzero(np.isin(edges[:, 0], far))[0][:8]   # out-of-range queries too
    elif outliers == "all":
        pos *= np.float32(1000.0)
    elif outliers == "tiny":
        pos *= np.float32(1e-4)
    keys = _fused_step_keys(n, D, edges, pos, sampled, k)
    ids = (keys & 0xFFFFFFFF).astype(np.int32)
    ref = oracle.knn_midpoints(pos, edges, sampled, k)
    assert np.array_equal(ids[:, 1:], ref)
    mid = oracle.midpoints(pos, edges)
    d2 = (keys >> 32).astype(np.uint32).view(np.float32)
    diff = mid[sampled][:, None, :] - mid[ids]
    want = np.zeros(ids.shape, np.float32)
    for d in range(D):   # fma chain in coordinate order; products of fp32 are exact in fp64
        want = (diff[..., d].astype(np.float64) ** 2 + want.astype(np.float64)).astype(np.float32)
    assert np.array_equal(d2, want)


@pytest.mark.parametrize("n,D,k,S,outliers", [
    (50000, 3, 10, 256, False),    # MFMA form, 391 tiles
    (50000, 3, 10, 1100, True),    # several query groups, queries outside the f16 range (the exact-query list)
    (30000, 2, 15, 300, False),
    (20000, 4, 10, 256, False),    # packed-VALU form, stride 4
    (12000, 8, 12, 64, False),
    (10000, 16, 32, 256, False),   # C5 shape: query slices over blockIdx.y
    (700000, 3, 10, 256, False),   # forced inside a launch of many rounds of workgroups (the default there is the separate kernel)
])
def test_thresholds_inside_the_fused_launch_equal_the_separate_kernel(n, D, k, S, outliers, monkeypatch):
    """The thresholds of the filtered scan are computed either by knn_tau_kernel or by the first workgroups of the fused
    spring+scan launch itself, handed to the other workgroups through a counter (csrc/tau_core.h).  Both forms must
    give the same keys, bit for bit, and the same positions after a run that uses the device sampler."""
    from graphem_rapids_amd import _native
    edges, pos, sampled = _random_case(n, D, 8, k, S, seed=77)
    if outliers:
        far = np.random.default_rng(3).permutation(n)[: n // 100]
        pos[far] *= np.float32(300.0)
        sampled[:8] = np.nonzero(np.isin(edges[:, 0], far))[0][:8]
    keys, out = {}, {}
    for separate in ("1", "0"):
        monkeypatch.setenv("GRAPHEM_HIP_TAU_SEPARATE", separate)
        keys[separate] = _fused_step_keys(n, D, edges, pos, sampled, k)
        eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=5)
        eng.set_positions(pos)
        eng.run(6)
        out[separate] = eng.get_positions()
        eng.close()
    assert np.array_equal(keys["0"], keys["1"])
    ids = (keys["0"] & 0xFFFFFFFF).astype(np.int32)
    assert np.array_equal(ids[:, 1:], oracle.knn_midpoints(pos, edges, sampled, k))
    assert np.abs(out["0"] - out["1"]).max() <= 1e-6


@pytest.mark.parametrize("reorder,D", [("off", 3), ("bfs", 3), ("bfs", 4), ("off", 8), ("bfs", 12)])
def test_skewed_degrees_hubs(reorder, D):
    """A graph with hubs (degrees 20000, 2000, 600 on top of a sparse random graph): one row owns more
    edges than a fused workgroup holds, so the engine must take its unfused kernels; one thread
    walks a 20000-long pull list in the reference's order.  Every phase against the oracle."""
    from graphem_rapids_amd import _native
    import graphem_rapids_amd as gra
    n, k, S = 50000, 10, 256      # (D > 3: the long-row instantiation of the wide MFMA kernel)
    rng = np.random.default_rng(11)
    base = gra.random_regular_edges(n, 4, seed=9).astype(np.int64)
    extra = []
    for hub, deg in ((17, 20000), (4021, 2000), (49999, 600)):
        nb = rng.choice(n, size=deg, replace=False)
        nb = nb[nb != hub]
        extra.append(np.stack([np.minimum(hub, nb), np.maximum(hub, nb)], axis=1))
    e = np.unique(np.concatenate([np.sort(base, axis=1)] + extra), axis=0)   # u < v, sorted, no duplicates
    edges = np.ascontiguousarray(e, dtype=np.int32)
    pos = rng.standard_normal((n, D)).astype(np.float32)
    sampled = rng.permutation(len(edges))[:S].astype(np.int32)
    eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, reorder=reorder)
    eng.set_positions(pos)
    assert np.array_equal(eng.spring_forces(), oracle.spring_forces(pos, edges, 1.0, 0.2))
    assert np.array_equal(eng.knn_midpoints(sampled), oracle.knn_midpoints(pos, edges, sampled, k))
    eng.step(sampled)
    ref = oracle.step(pos, edges, sampled, k, 1.0, 0.2, 0.5)
    assert np.abs(eng.get_positions() - ref).max() <= 1e-4
    eng.run(3)                      # device sampler on the same paths
    assert np.isfinite(eng.get_positions()).all()
    eng.close()


def _random_simple_graph(rng, n, m):
    """m distinct undirected edges u < v on n vertices (fewer if the graph is too small), CSR order."""
    u = rng.integers(0, n, size=3 * m + 8)
    v = rng.integers(0, n, size=3 * m + 8)
    keep = u != v
    e = np.unique(np.stack([np.minimum(u, v)[keep], np.maximum(u, v)[keep]], axis=1), axis=0)
    if len(e) > m:
        e = e[np.sort(rng.permutation(len(e))[:m])]
    return np.ascontiguousarray(e, dtype=np.int32)


@pytest.mark.parametrize("seed", range(24))
def test_random_small_configurations(seed):
    """Fuzz of the per-query / generic kernels: random sparse graphs (isolated vertices, uneven degrees), random
    dimension, neighbour count and sample size, including S >= E (no sampling) and k + 1 == E."""
    from graphem_rapids_amd import _native
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(6, 400))
    edges = _random_simple_graph(rng, n, int(rng.integers(3, 4 * n)))
    E = len(edges)
    D = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 13, 16, 20]))
    k = int(rng.integers(1, min(E - 1, 40) + 1)) if E > 1 else 0
    if seed % 6 == 0 and E > 1:
        k = E - 1                                   # k + 1 == E: every edge is a neighbour
    S = int(rng.integers(1, 2 * E + 2))              # S >= E: the engine uses every edge (pt.py:412)
    pos = (rng.standard_normal((n, D)) * rng.choice([0.05, 1.0, 30.0])).astype(np.float32)
    eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, min(S, E))
    eng.set_positions(pos)
    sampled = np.arange(E, dtype=np.int32) if S >= E else rng.permutation(E)[:S].astype(np.int32)
    assert np.array_equal(eng.spring_forces(), oracle.spring_forces(pos, edges, 1.0, 0.2))
    if k > 0:
        assert np.array_equal(eng.knn_midpoints(sampled), oracle.knn_midpoints(pos, edges, sampled, k))
    eng.step(None if S >= E else sampled)
    ref = oracle.step(pos, edges, sampled, k, 1.0, 0.2, 0.5)
    got = eng.get_positions()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-4, (n, E, D, k, S)
    eng.close()


@pytest.mark.parametrize("seed", range(16))
def test_random_fused_configurations(seed):
    """Fuzz of the fused spring+scan path (>= 16K own edges): random dimension in 2..16 (each its own instantiation),
    neighbour count up to 63, sample size up to 700, uneven degrees, optional internal reordering."""
    from graphem_rapids_amd import _native
    rng = np.random.default_rng(2000 + seed)
    n = int(rng.integers(4000, 30000))
    edges = _random_simple_graph(rng, n, int(rng.integers(17000, 90000)))
    E = len(edges)
    D = int(rng.choice([2, 3, 3, 4, 5, 6, 7, 8, 9, 11, 13, 14, 15, 16]))
    k = int(rng.choice([1, 5, 10, 15, 31, 32, 63]))
    S = int(rng.choice([1, 7, 64, 256, 257, 512, 700]))
    pos = (rng.standard_normal((n, D)) * rng.choice([0.1, 1.0, 5.0])).astype(np.float32)
    sampled = rng.permutation(E)[:S].astype(np.int32)
    eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, reorder=("bfs" if seed % 2 else "off"))
    eng.set_positions(pos)
    assert np.array_equal(eng.spring_forces(), oracle.spring_forces(pos, edges, 1.0, 0.2))
    assert np.array_equal(eng.knn_midpoints(sampled), oracle.knn_midpoints(pos, edges, sampled, k))
    eng.step(sampled)
    ref = oracle.step(pos, edges, sampled, k, 1.0, 0.2, 0.5)
    assert np.abs(eng.get_positions() - ref).max() <= 2e-4, (n, E, D, k, S)
    eng.run(3)
    assert np.isfinite(eng.get_positions()).all()
    eng.close()


@pytest.mark.parametrize("finish", ["gathered", "overlap"])
@pytest.mark.parametrize("seed", range(6))
def test_random_partitioned_configurations(seed, finish):
    """Fuzz of the multi-rank split step on one GPU: random world size, graph (uneven degrees, sometimes a hub),
    dimension, neighbour count; device sampler on every rank (ids must agree) for two of the three iterations; form B
    (one collective of slots) and form D (rows early, statistics + patch lists late)."""
    import torch
    from graphem_rapids_amd import _native
    from graphem_rapids_amd.distributed import HipShardEngine, partition_rows
    rng = np.random.default_rng(3000 + seed)
    world = int(rng.choice([2, 3, 5]))
    n = int(rng.integers(6000, 25000))
    edges = _random_simple_graph(rng, n, int(rng.integers(20000, 70000)))
    if seed % 2:
        hub = int(rng.integers(0, n))
        nb = rng.choice(n, size=1500, replace=False)
        nb = nb[nb != hub]
        extra = np.stack([np.minimum(hub, nb), np.maximum(hub, nb)], axis=1)
        edges = np.ascontiguousarray(np.unique(np.concatenate([edges.astype(np.int64), extra]), axis=0), dtype=np.int32)
    D = int(rng.choice([2, 3, 3, 4, 8]))
    k, S = int(rng.choice([5, 10, 20])), int(rng.choice([64, 256, 300]))
    pos = rng.standard_normal((n, D)).astype(np.float32)
    first = rng.permutation(len(edges))[:S].astype(np.int32)
    single = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=7)
    single.set_positions(pos)
    single.step(first)
    single.run(2)                      # device sampler, iterations 1 and 2
    ref = single.get_positions()
    single.close()
    shards = []
    for r in range(world):
        chunk, lo, hi = partition_rows(n, world, r)
        sh = HipShardEngine(n, D, edges, 1.0, 0.2, 0.5, k, S, 7, (lo, hi, 0, 0, _native.EDGES_HASHED), 0)
        (sh.gather_layout if finish == "gathered" else sh.overlap_layout)(world, r, chunk)
        sh.set_positions(pos)
        shards.append(sh)
    for t in range(3):
        for sh in shards:
            sh.step_begin(first if t == 0 else None)
        if finish == "overlap":
            for sh in shards:
                sh.step_pack_rows()
            rows = torch.stack([sh.rows_all[r].clone() for r, sh in enumerate(shards)])
            for sh in shards:
                sh.rows_all.copy_(rows)
        gathered = torch.stack([sh.partial.clone() for sh in shards]).contiguous()
        for sh in shards:
            sh.step_merge(gathered, world)
        if finish == "overlap":
            stats = torch.stack([sh.stats_all[r].clone() for r, sh in enumerate(shards)])
            for sh in shards:
                sh.stats_all.copy_(stats)
                sh.step_finish_overlap()
            continue
        slots = torch.stack([sh.gbuf[r].clone() for r, sh in enumerate(shards)])
        for sh in shards:
            sh.gbuf.copy_(slots)
            sh.step_finish_gathered()
    torch.cuda.synchronize()
    outs = [sh.get_positions() for sh in shards]
    for sh in shards:
        sh.eng.close()
    assert np.abs(outs[0] - ref).max() <= 5e-6, (world, n, len(edges), D, k, S)
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


def test_candidate_lists_between_one_and_two_extraction_halves():
    """A candidate list longer than the 8192 keys the register extraction takes (but within the list's 16384) is selected
    in two halves and merged: a dense cluster of midpoints around the queries makes such lists; ids must stay the oracle's."""
    from graphem_rapids_amd import _native
    import graphem_rapids_amd as gra
    n, D, k, S = 60000, 3, 10, 64
    edges = np.ascontiguousarray(gra.random_regular_edges(n, 8, seed=9), dtype=np.int32)
    rng = np.random.default_rng(4)
    pos = rng.standard_normal((n, D)).astype(np.float32)
    dense = rng.permutation(n)[:13400]                      # 22 % of the vertices in ONE point: ~12 K coincident midpoints
    pos[dense] = np.float32(3.0)
    inside = np.nonzero(np.isin(edges[:, 0], dense) & np.isin(edges[:, 1], dense))[0]
    sampled = np.concatenate([inside[:32], rng.permutation(len(edges))[:S - 32]]).astype(np.int32)
    eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S)
    eng.set_positions(pos)
    knn = eng.knn_midpoints(sampled)
    _, fin, ovf = eng.knn_last_counts()
    assert np.array_equal(knn, oracle.knn_midpoints(pos, edges, sampled, k))
    print("\nlongest candidate lists:", np.sort(fin)[-5:], "overflowed:", int(ovf.sum()))
    assert fin.max() > 8192 and fin.max() <= 16384 and ovf.sum() == 0     # the two-halves path was taken, nothing fell back
    eng.close()
