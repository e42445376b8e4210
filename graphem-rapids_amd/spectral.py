"""Spectral initialisation on the GPU.

The reference starts from eigenvectors 1..D of the normalised Laplacian of the symmetrised,
unweighted graph, computed with scipy's eigsh(L, D+1, which='SM') (pt.py:337-379) -- ARPACK in
regular mode on the small end of the spectrum, which is what makes it the dominant end-to-end cost
(37.6 s at n = 100 K) and impractical at a million vertices.

Here: Lanczos with full reorthogonalisation on B = 2I - L, whose LARGEST eigenpairs are the wanted
ones (theta = 2 - lambda).  Thick-restart Lanczos in a basis of ~80 vectors; the steps between two
restarts run inside the library (gh_trlan_sweep, csrc/spectral.hip): sparse operator and classical Gram-Schmidt twice as
hand-written fp64 kernels, no host synchronisation; the small projected eigenproblem and the restart (one GEMM) are
solved on the host / with torch at every restart.

Like the reference's result, the embedding is defined up to the sign of each eigenvector and up to a
rotation inside a degenerate eigenspace; tests compare eigenvalues and subspaces, not entries.
"""
import ctypes
import logging

import numpy as np
import scipy.sparse as sp
import torch

from . import _native

logger = logging.getLogger(__name__)


def symmetrised_csr(adjacency):
    """A + A^T with all weights 1 (pt.py:351-352), CSR with sorted indices, diagonal removed:
    csgraph.laplacian(normed=True) (pt.py:355) leaves self-loops out of the degree and overwrites the
    diagonal of L, so they must not reach the operator either."""
    a = sp.csr_matrix(adjacency + adjacency.transpose())
    a.sort_indices()
    rows = np.repeat(np.arange(a.shape[0], dtype=a.indices.dtype), np.diff(a.indptr))
    keep = rows != a.indices                       # (setdiag(0) + eliminate_zeros costs 0.3 s at 100 K vertices)
    counts = np.bincount(rows[keep], minlength=a.shape[0])
    indptr = np.concatenate([[0], np.cumsum(counts)]).astype(a.indptr.dtype)
    idx = a.indices[keep]
    return sp.csr_matrix((np.ones(len(idx), dtype=np.float64), idx, indptr), shape=a.shape)


def _lanczos(apply_b, n, want, dev, locked, tol, max_steps, check_every, gen, stop_below=None):
    """Lanczos with full reorthogonalisation on the complement of the `locked` vectors.
    Returns (theta desc, Ritz vectors (n, m) for the `want` largest, residuals, steps, converged).
    stop_below: give up early once the largest Ritz value has CONVERGED below this number."""
    nl = 0 if locked is None else locked.shape[0]
    V = torch.empty((max_steps + 1 + nl, n), dtype=torch.float64, device=dev)
    if nl:
        V[:nl] = locked
    v0 = torch.randn(n, dtype=torch.float64, generator=gen).to(dev)
    if nl:
        v0 -= locked.t() @ (locked @ v0)
        v0 -= locked.t() @ (locked @ v0)
    V[nl] = v0 / torch.linalg.vector_norm(v0)
    w = torch.empty(n, dtype=torch.float64, device=dev)
    alpha = torch.zeros(max_steps, dtype=torch.float64, device=dev)
    beta = torch.zeros(max_steps, dtype=torch.float64, device=dev)
    steps, theta, S, resid, converged = 0, None, None, None, False
    for j in range(max_steps):
        apply_b(V[nl + j], w)
        basis = V[: nl + j + 1]
        norm0 = torch.linalg.vector_norm(w)
        c = basis @ w                        # c[nl + j] is the Lanczos alpha_j
        alpha[j] = c[nl + j]
        w -= basis.t() @ c
        b = torch.linalg.vector_norm(w)
        if b < 0.7071 * norm0:               # cancellation: a second Gram-Schmidt pass ("twice is enough")
            w -= basis.t() @ (basis @ w)
            b = torch.linalg.vector_norm(w)
        beta[j] = b
        steps = j + 1
        V[nl + j + 1] = w / b
        if steps >= min(want, max_steps) and (steps % check_every == 0 or steps == max_steps):
            al = alpha[:steps].cpu().numpy()
            be = beta[:steps].cpu().numpy().copy()
            exhausted = be[-1] < 1e-13 * max(1.0, float(np.abs(al).max()))  # invariant subspace
            if exhausted:
                be[-1] = 0.0
            T = np.diag(al) + np.diag(be[:-1], 1) + np.diag(be[:-1], -1)
            th, Sm = np.linalg.eigh(T)
            th, Sm = th[::-1], Sm[:, ::-1]
            m = min(want, steps)
            resid = np.abs(be[-1] * Sm[-1, :m]) / np.maximum(np.abs(th[:m]), 1e-300)
            theta, S = th[:m], Sm[:, :m]
            if np.all(resid <= tol) or exhausted:
                converged = True
                break
            if stop_below is not None and resid[0] <= tol and th[0] < stop_below:
                converged = True
                break
    X = V[nl: nl + steps].t() @ torch.from_numpy(np.ascontiguousarray(S)).to(dev)
    return theta, X, resid, steps, converged


def _trlan(apply_b, n, want, dev, locked, tol, max_steps, check_every, gen, stop_below=None, basis=None, sweep=None):
    """Thick-restart Lanczos (Wu & Simon) with full reorthogonalisation inside a SMALL basis.

    Same contract as _lanczos.  The basis holds at most m vectors; when it is full the Rayleigh-Ritz
    problem of the projected matrix T = V^T B V (m x m, built column by column from the
    reorthogonalisation coefficients) is solved on the host, the `keep` best Ritz vectors plus the
    residual direction become the new basis, and the iteration continues.  Per step the dense work
    is two passes over a (<= m) x n basis instead of a (steps) x n one, which is what made the unrestarted
    version spend its time in Gram-Schmidt (n = 1 M: 12 s for 800 steps).

    sweep: the library's gh_trlan_sweep bound to the graph (laplacian_embedding_hip): all steps between two restarts in
    ONE call -- matvec and classical Gram-Schmidt twice as hand-written kernels, seven launches per step, no host
    synchronisation (before: ~ten torch launches and a device-to-host copy per step).  None: the same steps with torch
    GEMVs (kept as the cross-check, method='trlan_torch')."""
    nl = 0 if locked is None else locked.shape[0]
    m = int(basis) if basis else max(4 * want + 40, 80)
    m = max(min(m, n - nl), 1)
    if nl + m + 1 > 256:
        sweep = None                          # (the kernels stage <= 256 coefficients)
    keep = min(want + max(8, want), max(m - 2, 1))
    Vall = torch.empty((nl + m + 1, n), dtype=torch.float64, device=dev)   # the locked vectors, then the basis
    if nl:
        Vall[:nl] = locked
    V = Vall[nl:]
    v0 = torch.randn(n, dtype=torch.float64, generator=gen).to(dev)
    if nl:
        v0 -= locked.t() @ (locked @ v0)
        v0 -= locked.t() @ (locked @ v0)
    V[0] = v0 / torch.linalg.vector_norm(v0)
    w = torch.empty(n, dtype=torch.float64, device=dev)
    Td = torch.zeros((m, m), dtype=torch.float64, device=dev)   # upper triangle of T, column by column
    if sweep is not None:
        work = torch.empty(n + 2 * (nl + m + 1) + ((n + 511) // 512) * (nl + m + 1), dtype=torch.float64, device=dev)
        bh = torch.zeros((2, m), dtype=torch.float64, device=dev)   # beta, max |h| of every step
    k, steps, converged = 0, 0, False
    theta = S = resid = None
    while True:
        j, beta, exhausted = k, 0.0, False
        if sweep is not None:
            sweep(Vall, nl, m, k, Td, work, bh[0], bh[1])
            host = bh.cpu().numpy()
            j = m
            for jj in range(k, m):
                steps += 1
                if not np.isfinite(host[0, jj]) or host[0, jj] < 1e-13 * max(1.0, host[1, jj]):   # invariant subspace at step jj
                    exhausted, j = True, jj + 1
                    break
            beta = float(host[0, j - 1])
        while sweep is None and j < m:
            apply_b(V[j], w)
            steps += 1
            if nl:
                w -= locked.t() @ (locked @ w)
            bas = V[: j + 1]
            norm0 = torch.linalg.vector_norm(w)
            h = bas @ w
            w -= bas.t() @ h
            b = torch.linalg.vector_norm(w)
            if b < 0.7071 * norm0:           # cancellation: a second pass ("twice is enough")
                h2 = bas @ w
                w -= bas.t() @ h2
                if nl:
                    w -= locked.t() @ (locked @ w)
                h = h + h2
                b = torch.linalg.vector_norm(w)
            Td[: j + 1, j] = h
            beta, hmax = (float(x) for x in torch.stack([b, h.abs().max()]).cpu())   # one small sync per step
            if beta < 1e-13 * max(1.0, hmax):   # invariant subspace: nothing left to add
                exhausted = True
                j += 1
                break
            V[j + 1] = w / b
            j += 1
        mm = j                                # current basis size
        T = np.triu(Td[:mm, :mm].cpu().numpy())
        T = T + np.triu(T, 1).T
        th, Sm = np.linalg.eigh(T)
        th, Sm = th[::-1], Sm[:, ::-1]
        nw = min(want, mm)
        res = np.abs(beta * Sm[mm - 1, :]) / np.maximum(np.abs(th), 1e-300)
        if exhausted:
            res = np.zeros_like(res)
        theta, S, resid = th[:nw], Sm[:, :nw], res[:nw]
        if exhausted or np.all(resid <= tol):
            converged = True
            break
        if stop_below is not None and resid[0] <= tol and th[0] < stop_below:
            converged = True
            break
        if steps >= max_steps:
            break
        # thick restart: the best `keep` Ritz vectors, then the residual direction
        kk = min(keep, mm - 1)
        Y = (torch.from_numpy(np.ascontiguousarray(Sm[:, :kk].T)).to(dev) @ V[:mm])   # (kk, n)
        r_dir = V[mm].clone()
        V[:kk] = Y
        V[kk] = r_dir
        Td.zero_()
        Td[torch.arange(kk, device=dev), torch.arange(kk, device=dev)] = torch.from_numpy(np.ascontiguousarray(th[:kk])).to(dev)
        k = kk                                # the next column of T (coupling beta * S[mm-1, :kk]) comes out of h
    X = V[: S.shape[0]].t() @ torch.from_numpy(np.ascontiguousarray(S)).to(dev)
    return theta, X, resid, steps, converged


def laplacian_embedding_hip(adjacency, n_components, device="cuda:0", tol=1e-6, max_steps=None, check_every=10,
                            seed=0, return_info=False, check_multiplicity=True, method="trlan", dtype=np.float32):
    """(n, n_components) of `dtype` (float32; float64 for a float64 engine, pt.py:372-376 casts the eigenvectors to the
    embedder's dtype): eigenvectors 1..D (ascending eigenvalue) of the normalised Laplacian.

    tol bounds the relative Ritz residual |B y - theta y| / |theta| of every wanted pair (the start of
    a layout does not need ARPACK's machine-precision default).  A single Krylov sequence sees only one
    direction of a degenerate eigenspace (grids, repeated components), so with check_multiplicity the
    search is repeated on the complement of what was found until nothing above the cut is left."""
    lib = _native.load()
    dev = torch.device(device)
    a = symmetrised_csr(adjacency)
    n = a.shape[0]
    want = n_components + 1
    if want >= n:
        raise ValueError("n_components + 1 must be smaller than the number of vertices")
    max_steps_given = max_steps
    if max_steps is None:
        max_steps = max(20 * want, 800)
    max_steps = int(min(max_steps, n - 1))
    deg = np.diff(a.indptr).astype(np.float64)
    s_host = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1.0)), 0.0)
    indptr = torch.from_numpy(a.indptr.astype(np.int64)).to(dev)
    indices = torch.from_numpy(a.indices.astype(np.int32)).to(dev)
    s = torch.from_numpy(s_host).to(dev)
    stream_ptr = torch.cuda.current_stream(dev).cuda_stream

    def apply_b(x, y):
        st = lib.gh_spmv_symnorm(ctypes.c_void_p(stream_ptr), n, ctypes.c_void_p(indptr.data_ptr()),
                                 ctypes.c_void_p(indices.data_ptr()), ctypes.c_void_p(s.data_ptr()),
                                 ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()))
        if st != 0:
            raise RuntimeError(lib.gh_spectral_last_error().decode())

    def sweep(Vall, nl, m, k, Td, work, beta, hmax):
        st = lib.gh_trlan_sweep(ctypes.c_void_p(stream_ptr), n, ctypes.c_void_p(indptr.data_ptr()), ctypes.c_void_p(indices.data_ptr()),
                                ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(Vall.data_ptr()), int(nl), int(m), int(k),
                                ctypes.c_void_p(Td.data_ptr()), ctypes.c_void_p(work.data_ptr()), ctypes.c_void_p(beta.data_ptr()),
                                ctypes.c_void_p(hmax.data_ptr()))
        if st != 0:
            raise RuntimeError(lib.gh_spectral_last_error().decode())

    if method not in ("trlan", "trlan_torch", "lanczos"):
        raise ValueError("method must be 'trlan' (thick restart, the library's sweep kernels), 'trlan_torch' (the same with "
                         "torch GEMVs) or 'lanczos' (no restart)")
    if method == "trlan":
        def solve(*a, **kw):
            return _trlan(*a, sweep=sweep, **kw)
    else:
        solve = _trlan if method == "trlan_torch" else _lanczos
    if method in ("trlan", "trlan_torch") and max_steps_given is None:
        max_steps = int(min(max(10 * max_steps, 8000), 50 * n))   # matvecs are cheap here, the basis is what costs
    gen = torch.Generator(device="cpu").manual_seed(int(seed))
    theta, X, resid, steps, converged = solve(apply_b, n, want, dev, None, tol, max_steps, check_every, gen)
    total_steps, runs = steps, 1
    while check_multiplicity and converged and runs <= want and X.shape[1] + 1 < n:
        cut = float(theta[min(want, len(theta)) - 1])
        locked = X.t().contiguous()
        th2, X2, res2, st2, conv2 = solve(apply_b, n, want, dev, locked, tol,
                                             int(min(max_steps, n - 1 - locked.shape[0])), check_every, gen,
                                             stop_below=cut - 1e-9 * max(1.0, abs(cut)))
        total_steps += st2
        runs += 1
        new = [i for i in range(len(th2)) if res2[i] <= max(tol, 1e-8) and th2[i] >= cut - 1e-9 * max(1.0, abs(cut))]
        if not new:
            break
        theta = np.concatenate([theta, th2[new]])
        X = torch.cat([X, X2[:, new]], dim=1)
        order = np.argsort(-theta, kind="stable")
        theta = theta[order]
        X = X[:, torch.from_numpy(order.copy()).to(dev)]
        if len(theta) > want + len(new):      # keep a few spare copies so the next complement is right
            theta, X = theta[: want + len(new)], X[:, : want + len(new)]
    theta, X = theta[:want], X[:, :want]
    if not converged:  # the reference falls back to its random start on ArpackNoConvergence; say so at least
        logger.warning("laplacian_embedding_hip: not converged after %d matvecs (largest relative residual %.2e); "
                       "using the current Ritz vectors", total_steps, float(np.max(resid)))
    emb = X[:, 1:want].to(torch.float64 if dtype == np.float64 else torch.float32).cpu().numpy()          # drop the first (pt.py:365)
    if return_info:
        return emb, {"steps": total_steps, "runs": runs, "converged": converged, "eigenvalues": 2.0 - theta,
                     "residuals": resid, "vectors_all": X.cpu().numpy()}
    return emb
