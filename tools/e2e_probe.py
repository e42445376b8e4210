import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import graphem_rapids_amd as gra
adj = gra.generate_random_regular(1000000, 8, seed=0)
emb = gra.create_graphem(adj, n_components=3, backend="hip", verbose=False, seed=0); emb._engine.sync()
for rep in range(3):
    t0=time.perf_counter(); emb._engine.run(100, None); t1=time.perf_counter(); emb._engine.sync(); t2=time.perf_counter()
    p = emb._engine.get_positions(); t3=time.perf_counter(); q = emb.positions; t4=time.perf_counter()
    print("run enqueue %.1f ms, sync %.1f ms, get_positions %.1f ms, .positions %.1f ms" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3)))
t0=time.perf_counter(); r = emb.run_layout(100); t1=time.perf_counter(); print("run_layout(100) %.1f ms" % (1e3*(t1-t0)))
