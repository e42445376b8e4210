"""Where a fused spring+scan workgroup -- and the workgroups of the normalise launch after it -- spend their time:
wall-clock stamps (100 MHz) written by thread 0 at the phase boundaries of the last launch (diagnostic:
GRAPHEM_HIP_STAMPS).  Usage: python tools/stamp_probe.py [workload]"""
import ctypes, os, sys
os.environ["GRAPHEM_HIP_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from graphem_rapids_amd import _native
wl = sys.argv[1] if len(sys.argv) > 1 else "rr1m"
n, D, k, S, edges, pos = bench.make_workload(wl)
if os.environ.get("GRAPHEM_PROBE_DIM"):   # another number of components on the same graph
    D = int(os.environ["GRAPHEM_PROBE_DIM"])
    pos = (np.random.default_rng(0).standard_normal((n, D)) * 0.1).astype(np.float32)
eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S)
eng.set_positions(pos)
eng.run(8)
eng.sync()
EXTRA = 8192   # records of the normalise launch after the fused kernel's (engine.h GH_STAMP_EXTRA)
buf = np.full(1 << 22, np.iinfo(np.uint64).max, dtype=np.uint64)
st = eng.lib.gh_debug_stamps(eng.handle, buf.ctypes.data_as(ctypes.c_void_p), buf.size)
assert st == 0
have = int(np.nonzero(buf != np.iinfo(np.uint64).max)[0].max()) // 8 + 1
nvb = have - EXTRA
norm = buf[nvb * 8:have * 8].reshape(-1, 8).astype(np.int64)
buf = buf[:nvb * 8]
t = buf.reshape(-1, 8)
t = t[t[:, 0] > 0].astype(np.int64)
t0 = t[:, 0].min()
names = ["spring phase (gathers)", "block sums + barrier", "scan operands", "scan over all queries", "flush hits"]
print(wl, "workgroups", len(t), "kernel span %.1f us" % ((t[:, 5].max() - t0) / 100.0))
for i, nm in enumerate(names):
    d = (t[:, i + 1] - t[:, i]) / 100.0
    print("  %-26s median %6.2f us   p90 %6.2f   max %6.2f   sum/CU %.1f us" % (nm, np.median(d), np.quantile(d, 0.9), d.max(), d.sum() / 256))
life = (t[:, 5] - t[:, 0]) / 100.0
print("  workgroup lifetime median %.2f us, p90 %.2f; start times: p10 %.1f p50 %.1f p90 %.1f us" % (
    np.median(life), np.quantile(life, 0.9), *[(np.quantile(t[:, 0], q) - t0) / 100.0 for q in (0.1, 0.5, 0.9)]))
embedded = len(t) > 0 and int(np.median(t[:, 6])) > (1 << 20)   # slots 6, 7: wait for this launch's thresholds (else rows, owned edges)
idx = np.argsort(-life)[:5]
for i in idx:
    print("  slowest: workgroup", int(np.nonzero(buf.reshape(-1, 8)[:, 0] > 0)[0][i]), "lifetime %.1f us, phases" % life[i],
          [round(float(x), 2) for x in (t[i, 1:6] - t[i, 0:5]) / 100.0], "start %.1f us" % ((t[i, 0] - t0) / 100.0),
          "" if embedded else "rows %d owned edges %d" % (int(t[i, 6]), int(t[i, 7])))
if not embedded:
    print("  rows per workgroup: median", np.median(t[:, 6]), "max", t[:, 6].max(), "; owned edges median", np.median(t[:, 7]))

# ---- the normalise launch that followed: set-up workgroups (next iteration's queries and group minima) first
for kind, nm in ((1, "set-up"), (2, "normalising")):
    w = norm[norm[:, 6] == kind]
    if len(w) == 0:
        continue
    z = norm[(norm[:, 6] == 1) | (norm[:, 6] == 2)][:, 0].min()   # (the threshold producers' records share the region)
    print("normalise launch, %d %s workgroups (of the first %d): start p50 %.1f us after the first, span to last end %.1f us" % (
        len(w), nm, EXTRA, (np.median(w[:, 0]) - z) / 100.0, (w[:, 3].max() - z) / 100.0))
    print("  end times: p10 %.1f p50 %.1f p90 %.1f p99 %.1f us; start times p90 %.1f max %.1f us" % (
        *[(np.quantile(w[:, 3], q) - z) / 100.0 for q in (0.1, 0.5, 0.9, 0.99)], (np.quantile(w[:, 0], 0.9) - z) / 100.0, (w[:, 0].max() - z) / 100.0))
    print("  mean/std known after     median %6.2f us  max %6.2f" % (np.median(w[:, 1] - w[:, 0]) / 100.0, (w[:, 1] - w[:, 0]).max() / 100.0))
    if kind == 1 and (w[:, 2] > 0).any():
        print("  queries + tile staged    median %6.2f us  max %6.2f" % (np.median(w[:, 2] - w[:, 1]) / 100.0, (w[:, 2] - w[:, 1]).max() / 100.0))
        print("  distances, group minima  median %6.2f us  max %6.2f" % (np.median(w[:, 3] - w[:, 2]) / 100.0, (w[:, 3] - w[:, 2]).max() / 100.0))
    else:
        print("  rest                     median %6.2f us  max %6.2f" % (np.median(w[:, 3] - w[:, 1]) / 100.0, (w[:, 3] - w[:, 1]).max() / 100.0))

# ---- thresholds inside the fused launch (tau_core.h): the producers' stamps sit in the last records of the buffer
prod = norm[-64:]
prod = prod[prod[:, 0] > 0]
if len(prod):
    names2 = ["group minima loaded", "K-th smallest", "records stored (issued)", "stores acknowledged", "counter moved"]
    print("threshold producers: %d workgroups, start %.1f us after the first workgroup, counter complete at %.1f us" % (
        len(prod), (prod[:, 0].min() - t0) / 100.0, (prod[:, 5].max() - t0) / 100.0))
    for i, nm in enumerate(names2):
        d = (prod[:, i + 1] - prod[:, i]) / 100.0
        print("  %-26s median %6.2f us   max %6.2f" % (nm, np.median(d), d.max()))
    w = (t[:, 7] - t[:, 6]) / 100.0
    print("  consumers: waited median %.2f us, max %.2f, %d of %d workgroups more than 1 us; wait began %.1f us (median) after the first workgroup" % (
        np.median(w), w.max(), int((w > 1.0).sum()), len(w), (np.median(t[:, 6]) - t0) / 100.0))
