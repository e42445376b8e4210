"""The CPU oracle against the REFERENCE at full size (fixtures from
tests/golden/make_golden_large.py, which ran graphem_rapids' PyTorch-CPU backend in the build
container): BASELINE configs[1] (random-regular n=100 K), configs[2] (Erdos-Renyi n=1 M) and the
bench workload (random-regular n=1 M).  No GPU.

What is pinned here, per captured step with the reference's state injected:
  spring forces ............ bit-identical (sha1 of the whole (n, 3) array)
  KNN ids, ATen mode ....... oracle/aten_cdist_topk.cpp (cdist's matmul form + topk's partial_sort, restated
                             from PyTorch's source): the reference's ids, ties included, in every row
  intersection forces ...... bit-identical
  one step, ATen mode ...... bit-identical next positions (sha1 of the whole array), chained over 3 steps
  KNN ids, exact mode ...... exact-difference distances (the product's formula, = the reference's KeOps path
                             pt.py:531): identical rows at 100 K; at 1 M a row in ~100 differs, each difference
                             within cdist's fp32 quantum (explain_knn_differences); the step then agrees with
                             the reference to <= 2e-5 (1 + |x|) everywhere except on the endpoints of a flipped pair
Also: the SNAP-format ingest against what the reference's own parser returned for the same text.
"""
import os
import tempfile

import numpy as np
import pytest

import oracle
import refcase


def _check_case(name):
    c = refcase.load(name)
    g, edges, n = c["g"], c["edges"], c["n"]
    report = []
    for t in range(c["steps"]):
        pos = c["states"][t]
        assert pos is not None, f"{name}: the oracle (ATen mode) no longer reproduces the reference's state before step {t}"
        sampled, ref_knn = g[f"sampled_{t}"], g[f"knn_{t}"]
        # ---- the restatement of the reference, KNN as ATen computes it: everything bit for bit
        F = oracle.spring_forces(pos, edges, refcase.PARAMS[0], refcase.PARAMS[1])
        assert refcase.sha1(F) == str(g[f"F_spring_sha1_{t}"]), f"{name} step {t}: spring forces differ from the reference"
        assert np.array_equal(oracle.knn_midpoints_aten(pos, edges, sampled, refcase.K), ref_knn), \
            f"{name} step {t}: the ATen restatement (cdist + topk) no longer gives the reference's ids"
        Fi = oracle.intersection_forces(pos, edges, sampled, ref_knn, refcase.PARAMS[2])
        assert np.array_equal(Fi, refcase.dense_inter(g, t, n)), f"{name} step {t}: intersection forces differ"
        out = oracle.step_aten(pos, edges, sampled, refcase.K, *refcase.PARAMS)
        assert refcase.sha1(out) == str(g[f"pos_next_sha1_{t}"]), f"{name} step {t}: next positions differ"
        # ---- exact-difference KNN (what the product computes): how far from the reference's, and why
        knn = oracle.knn_midpoints(pos, edges, sampled, refcase.K)
        same, sets, recall = refcase.knn_agreement(knn, ref_knn)
        n_rows, n_self, worst = refcase.explain_knn_differences(pos, edges, sampled, knn, ref_knn)
        Fi2 = oracle.intersection_forces(pos, edges, sampled, knn, refcase.PARAMS[2])
        flipped = np.nonzero(np.any(Fi2 != Fi, axis=1))[0]           # vertices whose repulsion the flips changed
        out2 = oracle.step(pos, edges, sampled, refcase.K, *refcase.PARAMS)
        diff = np.abs(out2 - out).max(axis=1)
        rel = diff / (1.0 + np.abs(out).max(axis=1))   # a flipped O(1) force moves the column std: every vertex shifts in proportion to its coordinate
        rest = np.ones(n, dtype=bool)
        rest[flipped] = False
        report.append(dict(step=t, rows_identical=same, rows_set_equal=sets, recall=round(recall, 6), self_kept=n_self,
                           worst_gap_quanta=round(worst, 2), vertices_with_flipped_pairs=len(flipped),
                           p2_all=float(diff.max()), p2_elsewhere=float(diff[rest].max()),
                           p2_elsewhere_rel=float(rel[rest].max())))
        assert recall >= 0.999 and same >= 248, report[-1]
        assert report[-1]["p2_elsewhere_rel"] <= 2e-5 and report[-1]["p2_elsewhere"] <= 1e-3, report[-1]
    print(f"\n{name}: exact-difference KNN against the reference's cdist+topk\n  " + "\n  ".join(map(str, report)))
    return report


def test_c2_rr100k_oracle_reproduces_the_reference_bit_for_bit():
    rep = _check_case("c2_rr100k")
    # at 100 K vertices the two distance formulas give identical rows on every captured step
    assert all(r["rows_identical"] == 256 and r["p2_all"] == 0.0 for r in rep)


@pytest.mark.slow
def test_rr1m_oracle_reproduces_the_reference():
    _check_case("rr1m_ref")


@pytest.mark.slow
def test_c3_er1m_oracle_reproduces_the_reference():
    _check_case("c3_er1m_ref")


def test_snap_ingest_equals_the_reference_parser():
    """F2: load_snap_edge_list against SNAPDataset.load() (datasets.py:306-357) on the same text."""
    import graphem_rapids_amd as gra
    import snap_synth
    g = np.load(os.path.join(refcase.GOLDEN_DIR, "snap_fb_synth.npz"))
    text, labels = snap_synth.synth_text()
    assert snap_synth.text_sha1(text) == str(g["text_sha1"]), "the synthetic text drifted from the fixture's"
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "facebook_combined.txt")
        with open(path, "w", encoding="utf-8") as fh:
            fh.write(text)
        v, e = gra.load_snap_edge_list(path, relabel=False)           # exactly the reference's return value
        assert np.array_equal(v, g["vertices"]) and np.array_equal(e, g["edges"])
        vd, ed = gra.load_snap_edge_list(path, directed=True, relabel=False)
        assert np.array_equal(vd, g["vertices_directed"]) and len(ed) == int(g["n_edges_directed"])
        assert refcase.sha1(ed.astype(np.int64)) == str(g["edges_directed_sha1"])
        v2, e2 = gra.load_snap_edge_list(path)                        # compacted labels: indexes an adjacency
        assert np.array_equal(v2, np.arange(snap_synth.N_VERTICES))
        assert np.array_equal(g["vertices"][e2], g["edges"])
        assert len(e2) == snap_synth.N_EDGES and np.all(e2[:, 0] < e2[:, 1])
        # and the graph is the one the text was written from (labels are a permutation of the vertex ids)
        inv = np.argsort(labels)
        back = np.sort(np.sort(inv[e2], axis=1).view([("a", np.int64), ("b", np.int64)]).ravel())
        want = np.sort(snap_synth.synth_edges().view([("a", np.int64), ("b", np.int64)]).ravel())
        assert np.array_equal(back, want)
