"""The HIP path against the REFERENCE at full size, and the two BASELINE configs round 1 left
unexercised.  All through the C ABI; needs a real MI355X (`pytest -m gpu`).

  * c2_rr100k / rr1m_ref / c3_er1m_ref: fixtures produced by the reference's PyTorch-CPU backend
    (tests/golden/make_golden_large.py).  Per captured step, reference state injected:
      spring forces ........ bit-identical to the reference (sha1 of the whole array)
      intersection forces .. given the reference's neighbour ids: rtol 1e-6 of max|F|
      KNN ids .............. knn_distance='cdist' (parity mode): the reference's ids in all 256 rows of every step;
                             knn_distance='exact' (speed mode): identical to the oracle's exact-difference KNN,
                             agreement with the reference measured, every difference explained by cdist's
                             fp32 quantum (refcase.explain_knn_differences)
      one step (P2) ........ cdist mode: <= 1e-4 on EVERY vertex, no exemption; exact mode: <= 1e-4 wherever no
                             flipped neighbour pair reaches the vertex (<= 2e-5 (1+|x|) at 1 M)
  * C4: random-regular n = 4 M, d = 8 as EIGHT row-partitioned engines on this one GPU with the
    collectives emulated by device copies == one engine; spring forces and KNN ids exact vs the oracle.
  * C5: a SNAP-format text with facebook_combined's size and hubs (max degree ~1000) through
    load_snap_edge_list -> create_graphem(n_components=16, n_neighbors=32): every phase and one step
    against the oracle, vertex reordering off and on (first test of hubs at LD = 16).
"""
import os

import numpy as np
import pytest

import oracle
import refcase

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["c2_rr100k", "rr1m_ref", "c3_er1m_ref"])
def test_hip_exact_mode_against_the_reference_at_full_size(name):
    from graphem_rapids_amd import _native
    c = refcase.load(name)
    g, edges, n = c["g"], c["edges"], c["n"]
    Lm, ka, ki = refcase.PARAMS
    eng = _native.Engine(n, refcase.D, edges, Lm, ka, ki, refcase.K, refcase.S)
    report = []
    for t in range(c["steps"]):
        pos = c["states"][t]
        assert pos is not None, f"{name}: reference state before step {t} could not be regenerated"
        sampled, ref_knn = g[f"sampled_{t}"], g[f"knn_{t}"]
        eng.set_positions(pos)
        F = eng.spring_forces()
        assert refcase.sha1(F) == str(g[f"F_spring_sha1_{t}"]), f"{name} step {t}: spring forces are not the reference's bits"
        knn = eng.knn_midpoints(sampled)
        assert np.array_equal(knn, oracle.knn_midpoints(pos, edges, sampled, refcase.K)), f"{name} step {t}: not the exact KNN"
        same, sets, recall = refcase.knn_agreement(knn, ref_knn)
        n_rows, n_self, worst = refcase.explain_knn_differences(pos, edges, sampled, knn, ref_knn)
        Fr = refcase.dense_inter(g, t, n)
        Fi = eng.intersection_forces(sampled, ref_knn)
        np.testing.assert_allclose(Fi, Fr, rtol=1e-6, atol=1e-6 * max(1.0, float(np.abs(Fr).max())))
        # vertices whose repulsion differs because a neighbour pair flipped under cdist's rounding
        flipped = np.nonzero(np.any(oracle.intersection_forces(pos, edges, sampled, knn, ki) != Fr, axis=1))[0]
        eng.step(sampled)
        out = eng.get_positions()
        if f"pos_next_{t}" in g:
            ref_next, rows = g[f"pos_next_{t}"], np.arange(n)
        else:
            ref_next, rows = g[f"pos_next_sub_{t}"], np.arange(0, n, refcase.SUB)
        diff = np.abs(out[rows] - ref_next).max(axis=1)
        rel = diff / (1.0 + np.abs(ref_next).max(axis=1))
        rest = ~np.isin(rows, flipped)
        rec = dict(step=t, rows_identical=same, rows_set_equal=sets, recall=round(recall, 6), self_kept=n_self,
                   worst_gap_quanta=round(float(worst), 2), vertices_with_flipped_pairs=len(flipped),
                   p2_all=float(diff.max()), p2_elsewhere=float(diff[rest].max()), p2_elsewhere_rel=float(rel[rest].max()))
        report.append(rec)
        assert recall >= 0.999 and same >= 248, rec
        assert rec["p2_elsewhere_rel"] <= 2e-5 and rec["p2_elsewhere"] <= 1e-3, rec
        if name == "c2_rr100k":   # 100 K vertices: the two distance formulas agree in every row
            assert same == 256 and rec["p2_all"] <= 1e-4, rec
    eng.close()
    print(f"\n{name}: HIP vs the reference\n  " + "\n  ".join(map(str, report)))


@pytest.mark.parametrize("name", ["c2_rr100k", "rr1m_ref", "c3_er1m_ref"])
def test_hip_cdist_mode_is_the_reference_at_full_size(name):
    """knn_distance='cdist' (the parity mode, default with sampler='torch'): on every captured step of the 100 K and
    1 M-vertex fixtures the HIP path returns the reference's neighbour ids in ALL 256 rows, in its order -- ties and
    the rows where cdist's rounding swaps near neighbours or drops a neighbour instead of the sampled edge included --
    and the next positions agree on EVERY vertex: against the reference's full state where the fixture holds it
    (100 K), else against the oracle's ATen-mode step whose sha1 over all n rows equals the reference's."""
    from graphem_rapids_amd import _native
    c = refcase.load(name)
    g, edges, n = c["g"], c["edges"], c["n"]
    Lm, ka, ki = refcase.PARAMS
    eng = _native.Engine(n, refcase.D, edges, Lm, ka, ki, refcase.K, refcase.S, knn_distance="cdist")
    report = []
    for t in range(c["steps"]):
        pos = c["states"][t]
        assert pos is not None, f"{name}: reference state before step {t} could not be regenerated"
        sampled, ref_knn = g[f"sampled_{t}"], g[f"knn_{t}"]
        eng.set_positions(pos)
        knn = eng.knn_midpoints(sampled)
        full_pass, unresolved = eng.knn_cdist_stats()
        same, sets, recall = refcase.knn_agreement(knn, ref_knn)
        eng.step(sampled)
        out = eng.get_positions()
        if f"pos_next_{t}" in g:
            ref_next = g[f"pos_next_{t}"]
        else:   # all n rows of the reference's next state: the oracle's ATen-mode step, identified by the reference's sha1
            ref_next = c["states"][t + 1] if t + 1 < len(c["states"]) else None
            if ref_next is None:
                ref_next = oracle.step_aten(pos, edges, sampled, refcase.K, *refcase.PARAMS)
            assert refcase.sha1(ref_next) == str(g[f"pos_next_sha1_{t}"])
        p2_all = float(np.abs(out - ref_next).max())
        rec = dict(step=t, rows_identical=same, rows_set_equal=sets, recall=recall, full_pass_rows=full_pass,
                   unresolved_tie_rows=unresolved, p2_all_vertices=p2_all)
        report.append(rec)
        assert same == refcase.S and unresolved == 0, rec
        assert p2_all <= 1e-4, rec
    eng.close()
    print(f"\n{name}: HIP (knn_distance='cdist') vs the reference\n  " + "\n  ".join(map(str, report)))


def test_presetup_is_invalidated_when_host_ids_overwrite_the_sample(monkeypatch):
    """run(device sampler) leaves the NEXT iteration's KNN set-up done inside its last
    normalise launch; a per-phase call with caller ids then overwrites d_sampled.  The following run must
    redo the set-up instead of pairing stale query records with the new ids."""
    import graphem_rapids_amd as gra
    from graphem_rapids_amd import _native
    n, D, k, S = 30000, 3, 10, 256
    edges = gra.random_regular_edges(n, 8, seed=5).astype(np.int32)
    rng = np.random.default_rng(5)
    pos = (rng.standard_normal((n, D))).astype(np.float32)
    ids = rng.permutation(len(edges))[:S].astype(np.int32)

    def sequence():
        eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=3)
        eng.set_positions(pos)
        eng.run(2)
        cur = eng.get_positions()
        knn = oracle.knn_midpoints(cur, edges, ids, k)
        Fi = eng.intersection_forces(ids, knn)
        eng.run(2)
        out = eng.get_positions()
        eng.close()
        return Fi, out
    Fi_a, out_a = sequence()
    monkeypatch.setenv("GRAPHEM_HIP_NO_PRESETUP", "1")
    Fi_b, out_b = sequence()
    assert np.array_equal(Fi_a, Fi_b)
    assert np.array_equal(out_a, out_b)


def test_partitioned_engine_refuses_whole_graph_calls():
    """gh_step / gh_run / per-phase calls on a row partition would normalise with the own rows'
    statistics; they must fail loudly instead."""
    import graphem_rapids_amd as gra
    from graphem_rapids_amd import _native
    n = 20000
    edges = gra.random_regular_edges(n, 8, seed=1).astype(np.int32)
    eng = _native.Engine(n, 3, edges, 1.0, 0.2, 0.5, 10, 256, partition=(0, n // 2, 0, 0, _native.EDGES_HASHED))
    eng.set_positions(np.zeros((n, 3), np.float32))
    ids = np.arange(256, dtype=np.int32)
    for call in (lambda: eng.step(ids), lambda: eng.run(1), lambda: eng.knn_midpoints(ids),
                 lambda: eng.intersection_forces(ids, np.zeros((256, 10), np.int32))):
        with pytest.raises(ValueError, match="whole graph"):
            call()
    eng.close()


def test_c4_rr4m_eight_row_partitions_equal_one_engine():
    """BASELINE configs[3]: random-regular n = 4 M, d = 8 (E = 16 M), D = 3, row-partitioned over 8 ranks.
    Eight partitioned engines on ONE GPU with the two collectives of a step emulated by device copies must
    reproduce the single engine (the oracle of the multi-GPU mode), and the single engine's spring
    forces and KNN ids must equal the CPU oracle's exactly at this size."""
    import torch
    import graphem_rapids_amd as gra
    from graphem_rapids_amd import _native
    from graphem_rapids_amd.distributed import HipShardEngine, owned_edge_ids, partition_rows
    n, D, k, S, world = 4_000_000, 3, 10, 256, 8
    edges = np.ascontiguousarray(gra.random_regular_edges(n, 8, seed=0), dtype=np.int32)
    assert len(edges) == 16_000_000
    rng = np.random.default_rng(0)
    pos = (rng.standard_normal((n, D)) * 0.1).astype(np.float32)
    stream = np.stack([rng.permutation(len(edges))[:S] for _ in range(2)]).astype(np.int32)

    single = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S)
    single.set_positions(pos)
    assert np.array_equal(single.spring_forces(), oracle.spring_forces(pos, edges, 1.0, 0.2))
    assert np.array_equal(single.knn_midpoints(stream[0]), oracle.knn_midpoints(pos, edges, stream[0], k))
    single.run(2, stream)
    ref = single.get_positions()
    single.close()
    want = oracle.step(pos, edges, stream[0], k)          # and the first of the two steps against the oracle
    one = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S)
    one.set_positions(pos)
    one.step(stream[0])
    assert np.abs(one.get_positions() - want).max() <= 1e-4
    one.close()

    shards, owned = [], 0
    for r in range(world):
        chunk, lo, hi = partition_rows(n, world, r)
        shards.append(HipShardEngine(n, D, edges, 1.0, 0.2, 0.5, k, S, 0, (lo, hi, 0, 0, _native.EDGES_HASHED), 0))
        shards[-1].rank_layout(world, r, chunk)
        shards[-1].set_positions(pos)
    for t in range(2):
        for sh in shards:
            sh.step_begin(stream[t])
        gathered = torch.stack([sh.partial.clone() for sh in shards]).contiguous()   # all-gather of the keys
        for sh in shards:
            sh.step_merge(gathered, world)
        stats_all = torch.stack([sh.stats.clone() for sh in shards]).contiguous()    # all-gather of the statistics
        for sh in shards:
            sh.step_finish_own(stats_all)
        blocks = torch.stack([sh.pos_blocks[r].clone() for r, sh in enumerate(shards)])   # in-place all-gather of the blocks
        for sh in shards:
            sh.pos_blocks.copy_(blocks)
    torch.cuda.synchronize()
    first = shards[0].get_positions()
    assert np.abs(first - ref).max() <= 2e-6
    for sh in shards[1:]:
        assert np.array_equal(sh.get_positions(), first)   # every rank holds the same bits
    for sh in shards:
        sh.eng.close()


@pytest.mark.parametrize("reorder", ["1", "2"])   # GRAPHEM_HIP_REORDER: 1 = caller's vertex order, 2 = breadth-first
def test_c5_snap_text_with_hubs_at_16_components(reorder, monkeypatch, tmp_path):
    """BASELINE configs[4]: a SNAP edge-list text (comments, mixed whitespace, arbitrary labels, repeats, both
    directions) with facebook_combined's size and a hub of degree ~1000, through load_snap_edge_list and the
    public factory at n_components = 16, n_neighbors = 32: hubs on the LD = 16 fused kernel and long_sum_kernel."""
    import graphem_rapids_amd as gra
    import snap_synth
    monkeypatch.setenv("GRAPHEM_HIP_REORDER", reorder)
    text, _ = snap_synth.synth_text()
    path = tmp_path / "facebook_combined.txt"
    path.write_text(text, encoding="utf-8")
    vertices, e = gra.load_snap_edge_list(str(path))
    n = len(vertices)
    assert n == snap_synth.N_VERTICES and len(e) == snap_synth.N_EDGES
    deg = np.bincount(e.ravel(), minlength=n)
    assert deg.max() > 900 and (deg > 128).sum() > 50       # the hub paths are really taken
    emb = gra.create_graphem(gra.edges_to_adjacency(n, e), n_components=16, backend="hip", n_neighbors=32,
                             sample_size=256, verbose=False, seed=0, init="random", sampler="torch")
    edges = emb._edges_np
    assert np.array_equal(edges, e.astype(np.int32))         # the factory extracts the same edge list (pt.py:220-245)
    k, S = 32, 256
    rng = np.random.default_rng(3)
    for scale in (0.1, 1.0):                                  # the reference's random start, and a unit-std state
        pos = (rng.standard_normal((n, 16)) * scale).astype(np.float32)
        sampled = rng.permutation(len(edges))[:S].astype(np.int32)
        emb.positions = pos
        assert np.array_equal(emb._compute_spring_forces(), oracle.spring_forces(pos, edges, 1.0, 0.2))
        knn, _ = emb._locate_knn_midpoints(sampled)
        ref_knn = oracle.knn_midpoints(pos, edges, sampled, k)
        assert np.array_equal(knn, ref_knn)
        Fi = emb._compute_intersection_forces(ref_knn, sampled)
        Fr = oracle.intersection_forces(pos, edges, sampled, ref_knn, 0.5)
        np.testing.assert_allclose(Fi, Fr, rtol=1e-6, atol=1e-6 * max(1.0, float(np.abs(Fr).max())))
        emb._engine.step(sampled)
        want = oracle.step(pos, edges, sampled, k, 1.0, 0.2, 0.5)
        assert np.abs(emb.positions - want).max() <= 1e-4
    out = emb.run_layout(5)                                   # and the loop itself, host sampler as pt.py:409
    assert out.shape == (n, 16) and np.isfinite(out).all()
    np.testing.assert_allclose(out.astype(np.float64).std(0, ddof=1), 1.0, atol=1e-4)


def test_two_runs_of_the_bench_workload_give_identical_bits():
    """run-to-run reproducibility at 1 M vertices (fp64 atomics in the intersection phase are
    the only unordered sums; they are exact for the handful of terms a vertex receives)."""
    import graphem_rapids_amd as gra
    from graphem_rapids_amd import _native
    n = 1_000_000
    edges = refcase.load("rr1m_ref")["edges"]
    rng = np.random.default_rng(0)
    pos = (rng.standard_normal((n, 3)) * 0.1).astype(np.float32)
    outs = []
    for _ in range(2):
        eng = _native.Engine(n, 3, edges, 1.0, 0.2, 0.5, 10, 256, seed=11)
        eng.set_positions(pos)
        eng.run(20)
        outs.append(eng.get_positions())
        eng.close()
    assert np.array_equal(outs[0], outs[1])
