import sys; sys.path.insert(0, '.')
import numpy as np, torch
import graphem_rapids_amd as gra
from graphem_rapids_amd import _native
n, D, k, S = 300, 2, 5, 64
edges = gra.random_regular_edges(n, 4, seed=3).astype(np.int32)
E = len(edges)
pos = (np.random.default_rng(2).standard_normal((n, D)) * 0.1)
for dtype in ("float32", "float64"):
    torch.manual_seed(3)
    ids = np.stack([torch.randperm(E)[:S].numpy() for _ in range(40)]).astype(np.int32)
    outs = []
    for rep in range(3):
        a = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, dtype=dtype)
        a.set_positions(pos); a.run(40, ids); outs.append(a.get_positions()); a.close()
    print(dtype, "run reproducible:", np.array_equal(outs[0], outs[1]), np.array_equal(outs[0], outs[2]), np.abs(outs[0]-outs[1]).max())
    for it in (1, 2, 5, 33):
        torch.manual_seed(3); st = torch.get_rng_state().numpy().copy()
        a = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, dtype=dtype); b = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, dtype=dtype)
        a.set_positions(pos); b.set_positions(pos)
        a.run(it, ids[:it]); b.run_torch_sampled(it, st)
        print(dtype, it, np.array_equal(a.get_positions(), b.get_positions()), np.abs(a.get_positions()-b.get_positions()).max())
