"""gh_run_torch_sampled over thousands of iterations (many wraps of its host, pinned and device rings) against gh_run over the
ids torch.randperm itself draws: positions bit for bit, generator state identical.  python tools/sampler_soak.py [workload] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from graphem_rapids_amd import _native
wl = sys.argv[1] if len(sys.argv) > 1 else "rr100k"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
n, D, k, S, edges, pos = bench.make_workload(wl)
E = len(edges)
a = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, knn_distance="cdist")
b = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, knn_distance="cdist")
a.set_positions(pos); b.set_positions(pos)
torch.manual_seed(2024)
state = torch.get_rng_state().numpy().copy()
done = 0
for chunk in (1, 7, 129, 1000, iters - 1137):
    t0 = time.perf_counter()
    ids = np.stack([torch.randperm(E)[:S].numpy() for _ in range(chunk)]).astype(np.int32)
    t1 = time.perf_counter()
    a.run(chunk, ids); a.sync()
    t2 = time.perf_counter()
    b.run_torch_sampled(chunk, state); b.sync()
    t3 = time.perf_counter()
    done += chunk
    same = np.array_equal(a.get_positions(), b.get_positions()) and np.array_equal(state, torch.get_rng_state().numpy())
    print(f"{wl}: {done} iterations, positions and generator state identical: {same}; torch.randperm {1e3 * (t1 - t0) / chunk:.2f} ms per draw, "
          f"gh_run {1e6 * (t2 - t1) / chunk:.1f} us/it, gh_run_torch_sampled {1e6 * (t3 - t2) / chunk:.1f} us/it", flush=True)
    assert same
print("ok")
