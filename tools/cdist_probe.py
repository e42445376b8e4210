"""How many rows of a knn_distance='cdist' run get partial_sort's heap replayed, iteration by iteration
(gh_knn_cdist_stats), what an iteration costs (wall clock, no instrumentation), what the per-iteration kernels cost
(HIP events), and -- with GRAPHEM_HIP_STAMPS=1 -- where the replay kernel of the last iteration spent its time.
python tools/cdist_probe.py [workload] [iters]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from graphem_rapids_amd import _native

name = sys.argv[1] if len(sys.argv) > 1 else "rr1m"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n, D, k, S, edges, pos = bench.make_workload(name)
k = int(os.environ.get("PROBE_K", k))
eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=0, knn_distance="cdist")
eng.set_positions(pos)
rows = []
for t in range(iters):
    eng.step(None)
    full, unres = eng.knn_cdist_stats()
    _, _, ovf = eng.knn_last_counts()
    rows.append((full, int((ovf == 1).sum()), int((ovf >= 2).sum())))   # (ovf: 0 once the next set-up has run ahead)
    if os.environ.get("GRAPHEM_HIP_STAMPS") and t >= iters - 3 and full:
        buf = np.zeros(32 * full, dtype=np.uint64)
        eng.lib.gh_debug_stamps(eng.handle, buf.ctypes.data_as(ctypes.c_void_p), buf.size)
        for r in buf.reshape(full, 32):
            t0 = int(r[0])
            print("  replay us: heap %.1f prefix %.1f tail %.1f sort %.1f intersect %.1f | batches %d entered %d chunks %d P %d tail %d"
                  % tuple([(int(r[i + 1]) - int(r[i])) / 100.0 for i in range(5)] + [int(x) for x in r[6:11]]))
            print('     batches (us from heap-ready: listed, staged, processed, chunks):',
                  [((int(r[16 + 4 * b]) - int(r[1])) / 100.0, (int(r[17 + 4 * b]) - int(r[1])) / 100.0,
                    (int(r[18 + 4 * b]) - int(r[1])) / 100.0, int(r[19 + 4 * b])) for b in range(min(int(r[6]), 4))])
print(name, "per iteration (listed rows, of which ties only, of which list not provably complete):")
print(" ".join(f"{a}/{b}/{c}" for a, b, c in rows))
eng.run(10); eng.sync()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); eng.run(50); eng.sync(); ts.append((time.perf_counter() - t0) / 50 * 1e6)
print("us per iteration, 3 passes of 50:", " ".join("%.1f" % x for x in ts))
eng.timing_enable(True); eng.timing_reset(); eng.run(20); eng.sync()
print({kname: round(1e3 * tot / cnt, 1) for kname, (tot, cnt) in eng.timings().items()})
