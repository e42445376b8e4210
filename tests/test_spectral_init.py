"""GPU spectral initialisation against the reference's method: scipy eigsh on the
normalised Laplacian (pt.py:337-379).  Eigenvectors are defined up to sign / rotation inside an
eigenspace, so eigenvalues and subspaces are compared."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.sparse.csgraph import laplacian

pytestmark = pytest.mark.gpu


def _reference_eig(adj, want):
    sym = sp.csr_matrix(adj + adj.transpose())
    sym.data = np.ones_like(sym.data)
    L = laplacian(sym, normed=True)
    vals, vecs = np.linalg.eigh(L.toarray())  # dense fp64: the exact answer eigsh approximates
    return L, vals[:want], vecs[:, :want]


@pytest.mark.parametrize("kind,n,D", [("rr", 600, 3), ("er", 800, 2), ("rr", 400, 16), ("grid", 400, 3), ("two", 300, 2)])
def test_eigenpairs_match_dense_reference(kind, n, D):
    import graphem_rapids_amd as gra
    from graphem_rapids_amd.spectral import laplacian_embedding_hip
    if kind == "rr":
        adj = gra.generate_random_regular(n, 6, seed=1)
    elif kind == "er":
        adj = gra.erdos_renyi_graph(n, 0.02, seed=2)       # has isolated vertices and small components
    elif kind == "grid":
        side = int(np.sqrt(n))
        idx = np.arange(side * side).reshape(side, side)
        e = np.vstack([np.column_stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()]),
                       np.column_stack([idx[:-1, :].ravel(), idx[1:, :].ravel()])])
        adj = gra.edges_to_adjacency(side * side, e)       # degenerate eigenvalues (symmetry)
        n = side * side
    else:
        a = gra.generate_random_regular(n // 2, 4, seed=3)
        adj = sp.block_diag([a, a]).tocsr()                # two components: a 2-dimensional null space
    want = D + 1
    emb, info = laplacian_embedding_hip(adj, D, return_info=True, tol=1e-10)
    assert emb.shape == (n, D) and emb.dtype == np.float32 and np.isfinite(emb).all()
    assert info["converged"]
    L, vals, vecs = _reference_eig(adj, want + 4)
    np.testing.assert_allclose(info["eigenvalues"], vals[:want], atol=1e-8)
    X = info["vectors_all"]
    assert np.abs(X.T @ X - np.eye(want)).max() < 1e-8     # orthonormal Ritz vectors
    assert np.abs(L @ X - X * info["eigenvalues"]).max() < 1e-7
    # subspace agreement wherever the wanted set ends at a spectral gap
    if vals[want] - vals[want - 1] > 1e-6:
        P = vecs[:, :want]
        assert np.linalg.norm(X - P @ (P.T @ X)) < 1e-6


def test_embedder_init_option_and_layout_runs():
    import graphem_rapids_amd as gra
    adj = gra.generate_random_regular(2000, 6, seed=5)
    emb = gra.create_graphem(adj, n_components=3, backend="hip", verbose=False, seed=0, init="laplacian_hip")
    p0 = emb.get_positions()
    assert p0.shape == (2000, 3)
    # same subspace as the reference's own start (scipy eigsh, pt.py:364-365)
    ref = gra.create_graphem(adj, n_components=3, backend="hip", verbose=False, seed=0, init="laplacian").get_positions()
    q, _ = np.linalg.qr(ref.astype(np.float64))
    assert np.linalg.norm(p0 - q @ (q.T @ p0)) < 1e-4
    out = emb.run_layout(5)
    assert np.isfinite(out).all()
