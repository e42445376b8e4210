"""Queries per iteration whose candidate list overflowed (exact fallback inside the select launch) over a long run.
python tools/overflow_probe.py [workload] [iters]   (runs with GRAPHEM_HIP_NO_PRESETUP=1 so the flags survive the step)"""
import os, sys
os.environ["GRAPHEM_HIP_NO_PRESETUP"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, bench
from graphem_rapids_amd import _native
name = sys.argv[1] if len(sys.argv) > 1 else "rr16m"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 260
n, D, k, S, edges, pos = bench.make_workload(name)
eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=0)
eng.set_positions(pos)
rows = []
for t in range(iters):
    eng.step(None)
    sub, fin, ovf = eng.knn_last_counts()
    rows.append((int(ovf.sum()), int(fin.max()), int(np.median(fin))))
    if ovf.sum():
        print("iteration", t, "overflowed queries", int(ovf.sum()), "largest list", int(fin.max()), "median list", int(np.median(fin)), flush=True)
print(name, "iterations with an overflow:", sum(1 for r in rows if r[0]), "of", iters, "; median list length over the run", int(np.median([r[2] for r in rows])),
      "max list", max(r[1] for r in rows))
p = eng.get_positions()
print("layout: std", p.std(0), "max |x|", np.abs(p).max(), "median radius", np.median(np.linalg.norm(p, axis=1)))
