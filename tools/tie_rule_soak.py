"""Parity mode, the loop's tie rule at full size: two knn_distance='cdist' engines on a bench workload, one replaying only
the ties that can change a force (the default), one replaying every tie (gh_set_cdist_replay(1)), fed the same host-drawn ids:
positions must stay bit-identical; counts of replayed rows on both.  python tools/tie_rule_soak.py [workload] [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from graphem_rapids_amd import _native

wl = sys.argv[1] if len(sys.argv) > 1 else "rr1m"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n, D, k, S, edges, pos = bench.make_workload(wl)
E = len(edges)
few = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, knn_distance="cdist")
every = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, knn_distance="cdist")
every.set_cdist_replay(True)
few.set_positions(pos); every.set_positions(pos)
rng = np.random.default_rng(11)
lf = le = 0
for t in range(iters):
    ids = rng.permutation(E)[:S].astype(np.int32)
    few.step(ids); every.step(ids)
    lf += few.knn_cdist_stats()[0]; le += every.knn_cdist_stats()[0]
    if (t + 1) % 50 == 0:
        same = np.array_equal(few.get_positions(), every.get_positions())
        print(f"{wl} iteration {t + 1}: positions bit-identical {same}; rows replayed so far: {lf} (ties that matter) / {le} (every tie)", flush=True)
        assert same
print("ok")
