"""The device's power state against the layout's: 20-step passes of rr1m from a fresh process, back to back, after 3 s of
idle on the same layout, and on the early layout with a busy device (what bench.py's spin-up phase is for).
python tools/warm_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from graphem_rapids_amd import _native
n, D, k, S, edges, pos = bench.make_workload("rr1m")
eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=0)
eng.set_positions(pos)
def timed(steps):
    eng.sync(); t0 = time.perf_counter(); eng.run(steps); eng.sync(); return (time.perf_counter() - t0) / steps * 1e6
eng.run(5); eng.sync()
print("iterations 5..: 5 passes of 20:", [round(timed(20), 1) for _ in range(5)])
eng.run(500); eng.sync()
print("after 500 more, back to back:", [round(timed(20), 1) for _ in range(3)])
time.sleep(3.0)
print("same layout after 3 s idle:", [round(timed(20), 1) for _ in range(5)])
eng.set_positions(pos); eng.run(5); eng.sync()
print("early layout again, device busy just before:", [round(timed(20), 1) for _ in range(5)])
