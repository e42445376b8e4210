"""Per-rank KERNEL time of one partitioned iteration for world sizes 1, 2, 4, 8 on ONE GPU: rank r's engine runs the
split step with the collectives replaced by device copies of its own data (same sizes), and the engine's HIP-event
timers give what every kernel costs.  This is what a rank computes between collectives -- the input of the scaling model
in DESIGN.md section 6; the collectives themselves cannot be measured on a one-GPU box.

  python tools/rank_compute_time.py [workload] [finish: own|gathered|overlap] > profiles/rNN/rank_compute_<workload>.json

finish=overlap (form D): `before_rows_us` = what runs before the rows' all-gather can start (thresholds, fused kernel),
`beside_rows_us` = the kernels that run beside it (select, merge + intersection, corrections; + the two small collectives),
`after_rows_us` = patch + normalisation of all n rows (+ next set-up); `side_stream_us` = the pack kernel in front of the
all-gather on the side stream."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from graphem_rapids_amd.distributed import HipShardEngine, partition_rows

wl = sys.argv[1] if len(sys.argv) > 1 else "rr1m"
finish = sys.argv[2] if len(sys.argv) > 2 else "own"
n, D, k, S, edges, pos = bench.make_workload(wl)
out = {"workload": wl, "finish": finish, "n": n, "E": int(len(edges)), "rows": []}
for world in (1, 2, 4, 8):
    for rank in sorted({0, world // 2, world - 1}):
        chunk, lo, hi = partition_rows(n, world, rank)
        sh = HipShardEngine(n, D, edges, 1.0, 0.2, 0.5, k, S, 0, (lo, hi, 0, 0, 1), 0)
        {"own": sh.rank_layout, "gathered": sh.gather_layout, "overlap": sh.overlap_layout}[finish](world, rank, chunk)
        sh.set_positions(pos)
        gathered = torch.empty((world, S, k + 1), dtype=torch.int64, device="cuda")

        def it_overlap():
            sh.step_begin(None)
            sh.step_pack_rows()
            # stand-in for the early all-gather: every rank's block like this one's (as the form-B stand-in below: the statistics
            # are world x this rank's sums, so the layout stays a layout; torch copies, not on the timers)
            sh.rows_all.copy_(sh.rows_all[rank].expand_as(sh.rows_all).clone())
            for w in range(world):
                gathered[w].copy_(sh.partial)
            sh.step_merge(gathered, world)
            st = sh.stats_all[rank].clone()
            for w in range(world):
                sh.stats_all[w].copy_(st)                  # every rank's sums like this one's: the totals stay those of a layout
            sh.step_finish_overlap()

        def it():
            if finish == "overlap":
                return it_overlap()
            sh.step_begin(None)
            for w in range(world):
                gathered[w].copy_(sh.partial)              # stand-in for the all-gather of the keys
            sh.step_merge(gathered, world)
            if finish == "own":
                for w in range(world):
                    sh.stats_all[w].copy_(sh.stats)        # stand-in for the all-gather of the statistics
                sh.stats_all[1:].zero_()                   # (one rank's sums count once: rows of the others are absent anyway)
                sh.step_finish_own(sh.stats_all)
                if sh.packed_blocks is not None:           # the receiving side of the unpadded block exchange
                    # (stand-in for the all-gather: the other ranks' blocks = their current rows, so that the layout this
                    # rank keeps computing on stays a layout; torch copies, not on the engine's timers)
                    pk = sh.packed_blocks.view(world, chunk, D)
                    own = pk[rank].clone()
                    pk.copy_(sh.pos[: world * chunk].view(world, chunk, sh.ld)[:, :, :D])
                    pk[rank].copy_(own)
                    sh.step_unpack_rows()
            else:
                sh.gbuf.copy_(sh.gbuf[rank].expand_as(sh.gbuf).clone())
                sh.step_finish_gathered()
        for _ in range(5):
            it()
        torch.cuda.synchronize()
        sh.eng.timing_enable(True); sh.eng.timing_reset()
        for _ in range(20):
            it()
        torch.cuda.synchronize()
        tm = {a: round(1e3 * b[0] / b[1], 2) for a, b in sh.eng.timings().items()}
        rec = {"world": world, "rank": rank, "own_rows": hi - lo, "kernel_us": tm, "kernels_total_us": round(sum(tm.values()), 1),
               "bytes_sent_rows": (hi - lo) * (D if D < sh.ld else sh.ld) * 4 if world > 1 else 0}
        if finish == "overlap":
            grp = lambda names: round(sum(v for a, v in tm.items() if a in names), 1)
            before = {"knn_tau", "knn_setup", "spring_scan", "spring_long", "spring_mid", "new0"}
            after = {"patch_rows", "normalise_gathered"}
            side = {"pack_rows"}
            rec["before_rows_us"] = grp(before)
            rec["after_rows_us"] = grp(after)
            rec["side_stream_us"] = grp(side)
            rec["beside_rows_us"] = round(sum(v for a, v in tm.items() if a not in before | after | side), 1)
        out["rows"].append(rec)
        print(f"world={world} rank {rank}: kernels {rec['kernels_total_us']} us", tm, file=sys.stderr, flush=True)
        sh.eng.close()
print(json.dumps(out, indent=1))
