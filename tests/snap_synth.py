"""Synthetic SNAP-format edge-list text with facebook_combined's size and a heavy-tailed degree
sequence (BASELINE configs[4] without network access; SURVEY.md 8d "C5").

The text exercises everything the reference's SNAPDataset parser handles (datasets.py:306-357):
'#' comment lines (also in the middle), tabs and runs of spaces, trailing columns, arbitrary
non-contiguous integer labels, duplicate rows, both directions of some edges, self-loops and a
blank line.  Deterministic for a given seed (numpy PCG64); tests/golden/snap_fb_synth.npz holds the
sha1 of the text and what the REFERENCE's parser returned for it.
"""
import hashlib

import numpy as np

N_VERTICES = 4039      # facebook_combined (datasets.py:206-212)
N_EDGES = 88234
MAX_DEGREE = 1045      # its largest hub


def synth_edges(seed=2024, n=N_VERTICES, m=N_EDGES, max_deg=MAX_DEGREE):
    """(m, 2) int64 undirected simple edges u < v on 0..n-1 with power-law expected degrees
    (Chung-Lu style), largest expected degree ~ max_deg, every vertex of degree >= 1."""
    rng = np.random.default_rng(seed)
    # expected degree of vertex i = w_i = max(top * (i+1)^-alpha, 4) with alpha solved so that sum(w) = 2m
    ranks = np.arange(1, n + 1, dtype=np.float64)
    top = max_deg * 1.30               # rejected repeats thin the hubs out a little
    lo_a, hi_a = 0.0, 3.0
    for _ in range(60):
        alpha = 0.5 * (lo_a + hi_a)
        if np.maximum(top * ranks ** -alpha, 4.0).sum() > 2.0 * m:
            lo_a = alpha
        else:
            hi_a = alpha
    w = np.maximum(top * ranks ** -alpha, 4.0)
    p = w / w.sum()
    have = set()
    # a spanning path first so that no vertex is isolated (labels would otherwise go missing)
    perm = rng.permutation(n)
    for a, b in zip(perm[:-1], perm[1:]):
        have.add((min(a, b), max(a, b)))
    while len(have) < m:
        k = int((m - len(have)) * 1.3) + 16
        u = rng.choice(n, size=k, p=p)
        v = rng.choice(n, size=k, p=p)
        for a, b in zip(u, v):
            if a != b and len(have) < m:
                have.add((min(a, b), max(a, b)))
    e = np.array(sorted(have), dtype=np.int64)
    return e


def synth_text(seed=2024):
    """The file content (str) and the label of every vertex 0..n-1."""
    e = synth_edges(seed)
    rng = np.random.default_rng(seed + 1)
    n = N_VERTICES
    labels = np.sort(rng.choice(10 * n, size=n, replace=False)).astype(np.int64) + 17   # arbitrary, non-contiguous
    labels = labels[rng.permutation(n)]                                                 # and not monotone in the vertex id
    rows = []
    order = rng.permutation(len(e))
    flip = rng.random(len(e)) < 0.5
    for idx in order:
        a, b = e[idx]
        if flip[idx]:
            a, b = b, a
        rows.append((labels[a], labels[b]))
    # duplicates, reversed duplicates and self-loops
    for idx in rng.choice(len(e), size=500, replace=False):
        a, b = e[idx]
        rows.insert(int(rng.integers(0, len(rows))), (labels[b], labels[a]))
    for idx in rng.choice(len(e), size=300, replace=False):
        a, b = e[idx]
        rows.insert(int(rng.integers(0, len(rows))), (labels[a], labels[b]))
    for v in rng.choice(n, size=20, replace=False):
        rows.insert(int(rng.integers(0, len(rows))), (labels[v], labels[v]))
    seps = ["\t", " ", "  ", " \t "]
    out = ["# Undirected graph: synthetic stand-in for facebook_combined.txt",
           "# Nodes: %d Edges: %d" % (n, len(e)), "# FromNodeId\tToNodeId"]
    sep_pick = rng.integers(0, len(seps), size=len(rows))
    for i, (a, b) in enumerate(rows):
        line = "%d%s%d" % (a, seps[sep_pick[i]], b)
        if i % 997 == 0:
            line += " 1.0"            # a trailing column is ignored (len(values) >= 2)
        if i % 5001 == 0:
            out.append("# a comment in the middle")
        if i == 1234:
            out.append("")            # blank line
        out.append(line)
    return "\n".join(out) + "\n", labels


def text_sha1(text):
    return hashlib.sha1(text.encode("utf-8")).hexdigest()
