"""Times the GPU spectral start against scipy eigsh (the reference's method) on a random-regular graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from scipy.sparse.csgraph import laplacian
import torch
import graphem_rapids_amd as gra
from graphem_rapids_amd.spectral import laplacian_embedding_hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
D = 3
adj = gra.edges_to_adjacency(n, gra.random_regular_edges(n, 8, seed=0))
laplacian_embedding_hip(gra.generate_random_regular(1000, 4, 0), 2)  # warm-up (library load)
torch.cuda.synchronize()
t0 = time.perf_counter()
method = "lanczos" if "--lanczos" in sys.argv else "trlan"
emb, info = laplacian_embedding_hip(adj, D, return_info=True, method=method)
torch.cuda.synchronize()
t_gpu = time.perf_counter() - t0
print(f"n={n}: GPU {method} {t_gpu:.3f} s, {info['steps']} steps, {info['runs']} runs, converged={info['converged']}, "
      f"eigenvalues={info['eigenvalues']}, residuals={info['residuals']}", flush=True)
if "--scipy" in sys.argv:
    sym = sp.csr_matrix(adj + adj.T); sym.data = np.ones_like(sym.data)
    L = laplacian(sym, normed=True)
    t0 = time.perf_counter()
    vals, vecs = spla.eigsh(L, D + 1, which="SM")
    t_cpu = time.perf_counter() - t0
    print(f"scipy eigsh(which='SM') {t_cpu:.1f} s, eigenvalues={vals}; speed-up {t_cpu / t_gpu:.0f}x")
    q, _ = np.linalg.qr(vecs[:, 1:])
    print("subspace distance to scipy:", np.linalg.norm(emb - q @ (q.T @ emb)))
