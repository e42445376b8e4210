#!/usr/bin/env python3
"""F1 fixture: the REFERENCE's own spectral start at scale.

Runs ONLY in the build container (/root/reference mounted read-only).  Builds BASELINE configs[1]'s graph with the
reference's generate_random_regular(100000, 8, seed=0), constructs the FIRST embedder of the process
(create_graphem(..., n_components=3, backend='pytorch', device='cpu', seed=0): scipy eigsh(which='SM'), pt.py:337-379,
~40 s) and stores its start positions.  The output is data: p0 (n, 3) float32 = eigenvectors 1..3 of the normalised
Laplacian as ARPACK returned them, their Rayleigh quotients in fp64, the edge list's sha1.

Usage:  PYTHONDONTWRITEBYTECODE=1 GRAPHEM_RAPIDS_QUIET=true python tests/golden/make_golden_spectral.py
"""
import hashlib
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_large import _import_reference  # noqa: E402  (the same stubs for ndlib / loguru)


def main():
    import numpy as np
    import scipy.sparse as sp
    from scipy.sparse.csgraph import laplacian
    gr = _import_reference()
    n, d, D = 100000, 8, 3
    adj = gr.generate_random_regular(n=n, d=d, seed=0)
    t0 = time.time()
    emb = gr.create_graphem(adj, n_components=D, backend="pytorch", device="cpu", verbose=False, seed=0)
    p0 = np.ascontiguousarray(emb.get_positions(), dtype=np.float32)
    took = time.time() - t0
    rows, cols = adj.nonzero()
    keep = rows < cols
    edges = np.ascontiguousarray(np.column_stack([rows[keep], cols[keep]]), dtype=np.int32)
    sym = sp.csr_matrix(adj + adj.T)
    sym.data[:] = 1
    L = laplacian(sym, normed=True).astype(np.float64)
    x = p0.astype(np.float64)
    lam = np.array([(x[:, j] @ (L @ x[:, j])) / (x[:, j] @ x[:, j]) for j in range(D)])
    res = np.array([np.linalg.norm(L @ x[:, j] - lam[j] * x[:, j]) / np.linalg.norm(x[:, j]) for j in range(D)])
    print("eigsh start: %.1f s, Rayleigh quotients %s, relative residuals %s" % (took, lam, res))
    np.savez_compressed(os.path.join(HERE, "spectral_rr100k.npz"), p0=p0, rayleigh=lam, residual=res, n=n, d=d, D=D,
                        edges_sha1=hashlib.sha1(edges.tobytes()).hexdigest(),
                        generator="graphem_rapids.generate_random_regular(n=100000, d=8, seed=0); create_graphem(n_components=3, backend='pytorch', device='cpu', seed=0)")


if __name__ == "__main__":
    main()
