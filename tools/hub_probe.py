"""Iteration time on graphs with hubs (long pull lists walked by one thread)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import graphem_rapids_amd as gra
from graphem_rapids_amd import _native


def hub_graph(n, base_deg, hubs, seed=0):
    rng = np.random.default_rng(seed)
    base = gra.random_regular_edges(n, base_deg, seed=seed).astype(np.int64)
    extra = []
    for hub, deg in hubs:
        nb = rng.choice(n, size=deg, replace=False); nb = nb[nb != hub]
        extra.append(np.stack([np.minimum(hub, nb), np.maximum(hub, nb)], axis=1))
    e = np.unique(np.concatenate([np.sort(base, axis=1)] + extra), axis=0)
    return np.ascontiguousarray(e, dtype=np.int32)


def main():
  for name, n, D, k, hubs in [("fb-like n=4039 D=16 k=32, no hub", 4039, 16, 32, []),
                              ("fb-like n=4039 D=16 k=32, hub 1045", 4039, 16, 32, [(107, 1045), (1684, 792), (1912, 755)]),
                              ("n=100K D=3, no hub", 100000, 3, 10, []),
                              ("n=100K D=3, hubs 20000/2000", 100000, 3, 10, [(17, 20000), (4021, 2000)]),
                              ("n=1M D=3, hub 100000", 1000000, 3, 10, [(5, 100000)])]:
      edges = hub_graph(n, 8 if n > 10000 else 40, hubs)
      pos = np.random.default_rng(1).standard_normal((n, D)).astype(np.float32)
      eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, 256)
      eng.set_positions(pos)
      eng.run(5); eng.sync()
      t0 = time.perf_counter(); eng.run(30); eng.sync(); dt = (time.perf_counter() - t0) / 30
      eng.timing_enable(True); eng.timing_reset(); eng.run(10); eng.sync()
      tm = {a: round(1e3 * b[0] / b[1], 1) for a, b in eng.timings().items()}
      print(f"{name}: E={len(edges)} {1e6 * dt:.0f} us/iter", tm, flush=True)
      eng.close()


if __name__ == "__main__":
    main()
