"""Per-rank compute time of the split step for world sizes 1,2,4,8 on ONE GPU (collectives replaced by
device copies of this rank's own keys): what the multi-GPU iteration costs before communication."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from graphem_rapids_amd.distributed import HipShardEngine, partition_edges, partition_rows
wl = sys.argv[1] if len(sys.argv) > 1 else "rr1m"
n, D, k, S, edges, pos = bench.make_workload(wl)
rules = sys.argv[2].split(",") if len(sys.argv) > 2 else ["range", "hashed"]
for world, rule, rank in [(w, r, k_) for w in (1, 2, 4, 8) for r in rules for k_ in sorted({0, w - 1})]:
    chunk, lo, hi = partition_rows(n, world, rank)
    elo, ehi = partition_edges(edges, lo, hi)
    part = (lo, hi, elo, ehi, 0) if rule == "range" else (lo, hi, 0, 0, 1)
    sh = HipShardEngine(n, D, edges, 1.0, 0.2, 0.5, k, S, 0, part, 0)
    sh.gather_layout(world, rank, chunk)
    sh.set_positions(pos)
    gathered = torch.empty((world, S, k + 1), dtype=torch.int64, device="cuda")
    def it():
        sh.step_begin(None)
        for w in range(world):
            gathered[w].copy_(sh.partial)
        sh.step_merge(gathered, world)
        sh.gbuf.copy_(sh.gbuf[rank].expand_as(sh.gbuf).clone())  # stand-in for the all-gather of the slots
        sh.step_finish_gathered()
    for _ in range(5): it()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): it()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    sh.eng.timing_enable(True); sh.eng.timing_reset()
    for _ in range(10): it()
    torch.cuda.synchronize()
    tm = {a: round(1e3 * b[0] / b[1], 1) for a, b in sh.eng.timings().items()}
    print(f"world={world} {rule} rank {rank}: {1e6*dt:.0f} us per iteration (rows {hi-lo});", tm, flush=True)
    sh.eng.close()
