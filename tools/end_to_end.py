"""What a caller sees: create_graphem(...) + run_layout(...) on a large random-regular graph, by stage."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import graphem_rapids_amd as gra
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.perf_counter(); adj = gra.generate_random_regular(n, 8, seed=0); t1 = time.perf_counter()
print(f"generate_random_regular(n={n}, d=8): {t1 - t0:.2f} s", flush=True)
emb = gra.create_graphem(adj, n_components=3, backend="hip", verbose=False, seed=0); emb._engine.sync(); t2 = time.perf_counter()
print(f"create_graphem (edge extraction, engine, spectral start): {t2 - t1:.2f} s", flush=True)
pos = emb.run_layout(num_iterations=iters); t3 = time.perf_counter()
print(f"run_layout({iters}) incl. download: {t3 - t2:.3f} s ({1e3 * (t3 - t2) / iters:.3f} ms per iteration); positions {pos.shape}, finite {np.isfinite(pos).all()}", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
emb2 = gra.create_graphem(adj, n_components=3, backend="hip", verbose=False, seed=0)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
