"""Scale-capable synthetic graphs.

The reference builds its inputs with NetworkX (graphem_rapids/generators.py:32-49,
235-252), which is O(n^2) for G(n, p) and ~8 s per 100 K vertices for random-regular
graphs.  These generators are O(E), vectorised numpy, and return the same thing the
reference's do: a symmetric scipy CSR adjacency matrix with integer ones.
"""
import numpy as np
import scipy.sparse as sp


def edges_to_adjacency(n, edges):
    """(E, 2) undirected edge list -> symmetric CSR of ones (what _nx_to_sparse_adjacency returns)."""
    edges = np.asarray(edges).reshape(-1, 2)
    if len(edges) == 0:
        return sp.csr_matrix((n, n), dtype=np.int64)
    rows = np.concatenate([edges[:, 0], edges[:, 1]])
    cols = np.concatenate([edges[:, 1], edges[:, 0]])
    adj = sp.csr_matrix((np.ones(len(rows), dtype=np.int64), (rows, cols)), shape=(n, n))
    adj.sum_duplicates()
    adj.data[:] = 1
    return adj


def erdos_renyi_edges(n, p, seed=0):
    """G(n, p) by geometric skipping over the n(n-1)/2 vertex pairs: O(E) work.
    Returns (E, 2) int64 with u < v, sorted by (u, v)."""
    rng = np.random.default_rng(seed)
    total = n * (n - 1) // 2
    if p <= 0 or total == 0:
        return np.zeros((0, 2), dtype=np.int64)
    if p >= 1:
        iu = np.triu_indices(n, 1)
        return np.column_stack(iu).astype(np.int64)
    expected = total * p
    chunks, pos = [], -1
    while True:
        m = int(expected * 1.1 + 1000)
        gaps = rng.geometric(p, size=m).astype(np.int64)
        idx = pos + np.cumsum(gaps)
        over = np.searchsorted(idx, total)
        chunks.append(idx[:over])
        if over < m:
            break
        pos = int(idx[-1])
        expected = (total - pos) * p
    lin = np.concatenate(chunks)
    # linear index -> (u, v) in the strictly upper triangle, row-major
    nn = float(n)
    u = np.floor(((2 * nn - 1) - np.sqrt((2 * nn - 1) ** 2 - 8.0 * lin)) / 2).astype(np.int64)
    start = u * (2 * n - u - 1) // 2
    bad = start > lin          # fix the rare off-by-one of the float sqrt
    u[bad] -= 1
    start = u * (2 * n - u - 1) // 2
    nxt = (u + 1) * (2 * n - u - 2) // 2
    bad = lin >= nxt
    u[bad] += 1
    start = u * (2 * n - u - 1) // 2
    v = lin - start + u + 1
    return np.column_stack([u, v])


def erdos_renyi_graph(n, p, seed=0):
    """Same signature as the reference's erdos_renyi_graph (generators.py:32-49); a different
    random stream (numpy instead of NetworkX), the same distribution."""
    return edges_to_adjacency(n, erdos_renyi_edges(n, p, seed))


def random_regular_edges(n, d, seed=0, max_rounds=200):
    """d-regular simple graph by the pairing model with local repair: stubs are paired at
    random; pairs that are loops or repeats are dissolved together with an equal number of
    random good pairs and re-paired.  O(n d) per round, a handful of rounds in practice."""
    if (n * d) % 2 != 0:
        raise ValueError("n * d must be even")
    if d >= n:
        raise ValueError("d must be smaller than n")
    rng = np.random.default_rng(seed)
    good = np.zeros((0, 2), dtype=np.int64)
    stubs = np.repeat(np.arange(n, dtype=np.int64), d)
    for _ in range(max_rounds):
        stubs = rng.permutation(stubs)
        u, v = stubs[0::2], stubs[1::2]
        lo, hi = np.minimum(u, v), np.maximum(u, v)
        cand = np.column_stack([lo, hi])
        allp = np.vstack([good, cand])
        key = allp[:, 0] * n + allp[:, 1]
        order = np.argsort(key, kind="stable")
        sk = key[order]
        dup_sorted = np.zeros(len(sk), dtype=bool)
        dup_sorted[1:] = sk[1:] == sk[:-1]      # later copies of a repeated pair
        dup = np.zeros(len(sk), dtype=bool)
        dup[order] = dup_sorted
        bad = dup | (allp[:, 0] == allp[:, 1])
        bad[:len(good)] = False                  # established pairs stay
        good = allp[~bad]
        rest = allp[bad]
        if len(rest) == 0:
            order = np.lexsort((good[:, 1], good[:, 0]))
            return good[order]
        # dissolve as many random good pairs as there are bad ones so repair can succeed
        take = min(len(good), max(len(rest), 8))
        pick = rng.choice(len(good), size=take, replace=False)
        mask = np.ones(len(good), dtype=bool)
        mask[pick] = False
        stubs = np.concatenate([rest.ravel(), good[pick].ravel()])
        good = good[mask]
    raise RuntimeError("random_regular_edges did not converge")


def planted_partition_edges(n, communities, deg_in, deg_out, seed=0, shuffle=True):
    """A graph WITH structure for locality experiments (SNAP-like: dense communities, sparse links between them):
    `communities` equal blocks; about n * deg_in / 2 random pairs inside blocks and n * deg_out / 2 random pairs anywhere;
    loops and repeats dropped.  shuffle: vertex numbers permuted at random, so that the numbering says nothing about
    the blocks (an engine has to find the locality itself).  Returns (E, 2) int64 with u < v, sorted by (u, v)."""
    rng = np.random.default_rng(seed)
    size = n // communities
    m_in, m_out = int(n * deg_in / 2), int(n * deg_out / 2)
    u = rng.integers(0, size * communities, size=m_in)
    v = (u // size) * size + rng.integers(0, size, size=m_in)
    a = np.concatenate([u, rng.integers(0, n, size=m_out)])
    b = np.concatenate([v, rng.integers(0, n, size=m_out)])
    if shuffle:
        perm = rng.permutation(n)
        a, b = perm[a], perm[b]
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    keep = lo != hi
    key = np.unique(lo[keep] * n + hi[keep])
    return np.column_stack([key // n, key % n])


def generate_random_regular(n=100, d=3, seed=0):
    """Same signature as the reference's generate_random_regular (generators.py:235-252)."""
    return edges_to_adjacency(n, random_regular_edges(n, d, seed))


def load_snap_edge_list(path, directed=False, relabel=True):
    """SNAP text edge list -> (vertices, edges), the format and the rules of the reference's
    SNAPDataset.load() (datasets.py:306-357): lines starting with '#' are comments, a row is
    'src<ws>dst' (further columns ignored, rows with fewer than two fields skipped); an undirected
    dataset (directed=False) becomes the sorted unique pairs with u < v (self-loops and repeats in
    either direction drop out), a directed one keeps its rows as they come; vertices = the sorted
    labels that occur in the edges.

    relabel=False returns exactly what the reference's loader returns (original labels).
    relabel=True (default) compacts the labels to 0..n-1 in sorted-label order -- what the
    reference's load_dataset_as_networkx does next with convert_node_labels_to_integers
    (datasets.py:761-782) -- so that `edges` indexes an n x n adjacency directly; `vertices` is then
    arange(n).  No download, no NetworkX, no Python-level pair handling (vectorised numpy)."""
    src, dst = [], []
    with open(path, "r", encoding="utf-8") as fh:
        for line in fh:
            if line.startswith("#"):
                continue
            parts = line.strip().split()
            if len(parts) >= 2:
                src.append(int(parts[0]))
                dst.append(int(parts[1]))
    edges = np.column_stack([np.asarray(src, dtype=np.int64), np.asarray(dst, dtype=np.int64)]).reshape(-1, 2)
    if not directed and len(edges):
        lo, hi = np.minimum(edges[:, 0], edges[:, 1]), np.maximum(edges[:, 0], edges[:, 1])
        keep = lo != hi
        edges = np.unique(np.column_stack([lo[keep], hi[keep]]), axis=0)
    vertices = np.unique(edges.ravel())
    if relabel:
        edges = np.searchsorted(vertices, edges)
        vertices = np.arange(len(vertices), dtype=np.int64)
    return vertices, edges
