"""The multi-GPU driver with REAL processes and REAL HIP engines: `world` processes share this one GPU, each with its own
HipShardEngine for its row block, collectives by torch.distributed's gloo backend on the engines' device buffers (RCCL
refuses two ranks on one device) -- everything of an N > 1 run except RCCL itself: process groups (form D's second one),
the side stream, in-place all-gathers on views of engine memory, the device sampler agreeing across processes.
Must reproduce the single engine and leave identical positions on every rank."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, finish, knn_distance, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from graphem_rapids_amd.distributed import PartitionedLayout
    d = np.load(os.path.join(out_dir, "case.npz"))
    n, D, k, S = int(d["n"]), int(d["D"]), int(d["k"]), int(d["S"])
    lay = PartitionedLayout(n, D, d["edges"], 1.0, 0.2, 0.5, k, S, seed=4, rank=rank, world=world, device_id=0,
                            finish=finish, knn_distance=knn_distance)
    lay.set_positions(d["pos"])
    lay.run(len(d["stream"]), d["stream"])
    lay.run(2)                      # engine-drawn ids: the same on every process
    lay.sync()
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"pos_r{rank}.npy"), lay.get_positions())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,finish,knn_distance", [(2, "overlap", "exact"), (3, "overlap", "exact"), (4, "overlap", "exact"), (2, "own", "exact"),
                                                        (2, "gathered", "exact"), (2, "overlap", "cdist")])
def test_processes_on_one_gpu_equal_the_single_engine(world, finish, knn_distance, tmp_path):
    import torch.multiprocessing as mp
    import graphem_rapids_amd as gra
    from graphem_rapids_amd import _native
    n, D, k, S = 30011, 3, 10, 256
    edges = np.ascontiguousarray(gra.random_regular_edges(n - 1, 8, seed=5), dtype=np.int32)   # last vertex isolated
    rng = np.random.default_rng(6)
    pos = (rng.standard_normal((n, D)) * 0.1).astype(np.float32)
    stream = np.stack([rng.permutation(len(edges))[:S] for _ in range(3)]).astype(np.int32)
    np.savez(tmp_path / "case.npz", n=n, D=D, k=k, S=S, edges=edges, pos=pos, stream=stream)
    single = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=4, knn_distance=knn_distance)
    single.set_positions(pos)
    single.run(3, stream)
    single.run(2)
    ref = single.get_positions()
    single.close()
    mp.spawn(_worker, args=(world, _free_port(), finish, knn_distance, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(tmp_path / f"pos_r{r}.npy") for r in range(world)]
    assert np.abs(outs[0] - ref).max() <= 4e-6   # (five iterations from the 0.1-sigma start: the statistics' summation order, amplified)
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


@pytest.mark.parametrize("finish", ["overlap", "own"])
def test_bench_multi_rank_code_path_on_one_gpu(finish):
    """bench.py as the driver launches it for N > 1 -- torch.distributed.run, one rank per process, barrier + max-over-ranks
    timing, rank 0's JSON line with rank0_us_per_step -- rehearsed with two ranks on this one GPU over gloo
    (GRAPHEM_BENCH_REHEARSAL=gloo: every rank on cuda:0).  The numbers mean nothing; the code path must run and report."""
    import json
    import subprocess
    env = dict(os.environ, GRAPHEM_BENCH_REHEARSAL="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--workload", "rr100k", "--finish", finish]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and d["config"]["finish"] == finish
    assert d["config"]["parallelism"].startswith("rows/2")
    r0 = d["rank0_us_per_step"]
    assert r0["kernels"] > 0 and r0["finish"] == finish
    if finish == "overlap":
        assert r0["rows_allgather_exposed"] is not None
