"""Host time per iteration of the Python-driven partitioned loop (how fast the host can enqueue an iteration: the floor of the
N > 1 default loop), world 1 over real RCCL.  python tools/python_loop_host_time.py [workload] [finish]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist, bench
from graphem_rapids_amd.distributed import PartitionedLayout
wl = sys.argv[1] if len(sys.argv) > 1 else "rr100k"
finish = sys.argv[2] if len(sys.argv) > 2 else "overlap"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n, D, k, S, edges, pos = bench.make_workload(wl)
for native in (False, True):
    lay = PartitionedLayout(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=0, rank=0, world=1, device_id=0, finish=finish, native=native)
    lay.set_positions(pos); lay.run(20); lay.sync(); torch.cuda.synchronize()
    iters = 300
    t0 = time.perf_counter(); lay.run(iters); t1 = time.perf_counter(); lay.sync(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{wl} {finish} native={native}: host enqueue {1e6 * (t1 - t0) / iters:.1f} us per iteration, with the GPU {1e6 * (t2 - t0) / iters:.1f}", flush=True)
    lay.engine.eng.close()
dist.destroy_process_group()
