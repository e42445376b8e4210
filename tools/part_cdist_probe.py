"""Parity mode on row partitions, what rank 0 computes after the all-gather of the keys: `world` engines on ONE GPU (the
all-gather emulated by stacking the ranks' records), a few iterations from the reference's start, then the engine's
HIP-event timers of rank 0's merge + prefix + replay.  Usage: python tools/part_cdist_probe.py [workload] [world ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from graphem_rapids_amd.distributed import HipShardEngine, partition_rows
from graphem_rapids_amd import _native

wl = sys.argv[1] if len(sys.argv) > 1 else "rr1m"
worlds = [int(x) for x in sys.argv[2:]] or [2, 4, 8]
n, D, k, S, edges, pos = bench.make_workload(wl)
E = len(edges)
rng = np.random.default_rng(3)
for world in worlds:
    shards = []
    for r in range(world):
        chunk, lo, hi = partition_rows(n, world, r)
        sh = HipShardEngine(n, D, edges, 1.0, 0.2, 0.5, k, S, 0, (lo, hi, 0, 0, _native.EDGES_HASHED), 0, knn_distance="cdist")
        sh.rank_layout(world, r, chunk)
        sh.set_positions(pos)
        shards.append(sh)
    iters, listed = 12, []
    shards[0].eng.timing_enable(True)
    for t in range(iters):
        sampled = rng.permutation(E)[:S].astype(np.int32)
        if t == 2:
            shards[0].eng.timing_reset()
        for sh in shards:
            sh.step_begin(sampled)
        gathered = torch.stack([sh.partial.clone() for sh in shards]).contiguous()
        for sh in shards:
            sh.step_merge(gathered, world)
        listed.append(shards[0].eng.knn_cdist_stats()[0])
        stats_all = torch.stack([sh.stats.clone() for sh in shards]).contiguous()
        for sh in shards:
            sh.step_finish_own(stats_all)
        blocks = torch.stack([sh.pos_blocks[r].clone() for r, sh in enumerate(shards)])
        for sh in shards:
            sh.pos_blocks.copy_(blocks)
    torch.cuda.synchronize()
    tm = {kk: round(1e3 * tot / cnt, 1) for kk, (tot, cnt) in shards[0].eng.timings().items()}
    print(wl, "world", world, "listed rows per iteration", listed, "rank 0 kernels (us):", tm)
    for sh in shards:
        sh.eng.close()
