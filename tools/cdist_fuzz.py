"""Parity-mode fuzz: random graphs, embeddings with and without ties, random k / sample sizes -- single engines and
emulated row partitions -- every neighbour row against ATen's (oracle.knn_midpoints_aten).  Usage:
python tools/cdist_fuzz.py [configs = 120] [seed = 1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import graphem_rapids_amd as gra
from graphem_rapids_amd import _native
from graphem_rapids_amd.distributed import HipShardEngine, partition_rows
import oracle
from test_hip_cdist import _positions

N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
kinds = ["gauss", "start", "lattice", "lattice_fine", "collapsed"]
bad_total, t0 = 0, time.time()
for c in range(N):
    n = int(rng.integers(2000, 60000))
    deg = int(rng.choice([4, 6, 8, 12]))
    D = int(rng.integers(2, 17))
    k = int(rng.choice([1, 2, 3, 5, 8, 10, 10, 10, 14, 15, 16, 20, 32, 40]))
    S = int(rng.choice([64, 256, 256, 700]))
    kind = str(rng.choice(kinds))
    world = int(rng.choice([1, 1, 1, 2, 3, 5]))
    if (n * deg) % 2:
        n += 1
    edges = np.ascontiguousarray(gra.random_regular_edges(n, deg, seed=int(rng.integers(1 << 30))), dtype=np.int32)
    E = len(edges)
    S = min(S, E)
    if (k + 1) * 64 > E:
        continue
    pos = _positions(kind, n, D, rng)
    sampled = rng.permutation(E)[:S].astype(np.int32)
    want = oracle.knn_midpoints_aten(pos, edges, sampled, k)
    if world == 1:
        eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, knn_distance="cdist")
        eng.set_positions(pos)
        got = [eng.knn_midpoints(sampled)]
        full, unresolved = eng.knn_cdist_stats()
        eng.close()
    else:
        shards = []
        for r in range(world):
            chunk, lo, hi = partition_rows(n, world, r)
            sh = HipShardEngine(n, D, edges, 1.0, 0.2, 0.5, k, S, 0, (lo, hi, 0, 0, _native.EDGES_HASHED), 0, knn_distance="cdist")
            sh.rank_layout(world, r, chunk)
            sh.set_positions(pos)
            shards.append(sh)
        for sh in shards:
            sh.step_begin(sampled)
        gathered = torch.stack([sh.partial.clone() for sh in shards]).contiguous()
        for sh in shards:
            sh.step_merge(gathered, world)
        got = [sh.merged_knn() for sh in shards]
        full, unresolved = shards[0].eng.knn_cdist_stats()
        for sh in shards:
            sh.eng.close()
    nbad = max(int((~(g == want).all(axis=1)).sum()) for g in got)
    bad_total += nbad + unresolved
    print(f"{c:3d} n={n} deg={deg} D={D} k={k} S={S} {kind:12s} world={world}: listed {full:3d} differing rows {nbad} unresolved {unresolved}", flush=True)
print(f"{N} configurations, {bad_total} differing or unresolved rows, {time.time() - t0:.0f} s")
sys.exit(1 if bad_total else 0)
