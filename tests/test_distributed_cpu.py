"""The multi-rank path (graphem_rapids_amd.distributed.PartitionedLayout) under gloo on CPUs,
world sizes 2 and 3, with the CPU stand-in engine: partition arithmetic, the collective sequence
and the in-place position all-gather must reproduce the single-rank result and the oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from graphem_rapids_amd.distributed import PartitionedLayout
    from cpu_shard_engine import CpuShardEngine
    n, D, edges, pos, stream, k, S = case[:7]
    lay = PartitionedLayout(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=3, rank=rank, world=world,
                            engine_factory=CpuShardEngine, edge_ownership=case[7] if len(case) > 7 else "auto",
                            finish=case[8] if len(case) > 8 else "own", knn_distance=case[9] if len(case) > 9 else "exact")
    lay.set_positions(pos)
    lay.run(len(stream), stream)
    np.save(os.path.join(out_dir, f"pos_w{world}_r{rank}.npy"), lay.get_positions())
    if len(case) > 9 and case[9] == "cdist":   # the rows of the last step, and how many of them needed the heap replayed
        np.save(os.path.join(out_dir, f"knn_w{world}_r{rank}.npy"), lay.engine.last_knn)
        np.save(os.path.join(out_dir, f"listed_w{world}_r{rank}.npy"), np.array([lay.engine.last_listed]))
    lay.set_positions(pos)
    lay.run(2)  # engine-drawn samples must agree across ranks too
    np.save(os.path.join(out_dir, f"auto_w{world}_r{rank}.npy"), lay.get_positions())
    dist.destroy_process_group()


def _case(n=403, D=3, deg=6, k=6, S=40, iters=3):
    import graphem_rapids_amd as gra
    edges = gra.random_regular_edges(n - 1, deg, seed=1).astype(np.int32)  # vertex n-1 stays isolated
    rng = np.random.default_rng(0)
    pos = rng.standard_normal((n, D)).astype(np.float32)
    stream = np.stack([rng.permutation(len(edges))[:S] for _ in range(iters)]).astype(np.int32)
    return n, D, edges, pos, stream, k, S


@pytest.mark.parametrize("world,rule,finish", [(2, "hashed", "own"), (3, "hashed", "own"), (2, "range", "own"),
                                               (3, "hashed", "gathered"), (2, "hashed", "overlap"), (3, "hashed", "overlap")])
def test_partitioned_layout_matches_single_rank_and_oracle(world, rule, finish, tmp_path):
    import oracle
    case = _case() + (rule, finish)
    n, D, edges, pos, stream, k, S = case[:7]
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    mp.spawn(_worker, args=(1, _free_port(), case, str(tmp_path)), nprocs=1, join=True)
    single = np.load(tmp_path / "pos_w1_r0.npy")
    ref = oracle.run_layout(pos, edges, stream, k)
    assert np.abs(single - ref).max() < 1e-4
    for r in range(world):
        got = np.load(tmp_path / f"pos_w{world}_r{r}.npy")
        assert np.abs(got - single).max() < 1e-5, f"rank {r}"       # N ranks == 1 rank
        assert np.abs(got - ref).max() < 1e-4                        # and == the oracle
        auto = np.load(tmp_path / f"auto_w{world}_r{r}.npy")
        assert np.array_equal(auto, np.load(tmp_path / f"auto_w{world}_r0.npy"))  # every rank holds the same positions


@pytest.mark.parametrize("world,finish", [(2, "own"), (3, "own"), (2, "overlap")])
@pytest.mark.parametrize("kind", ["gauss", "lattice"])
def test_parity_mode_on_row_partitions(world, finish, kind, tmp_path):
    """knn_distance='cdist' on row partitions: every rank sends its k + 2 best cdist keys and a flag, the merge decides the
    rows without a tie and hands the others to partial_sort's replay over all edges -- the rows must be the reference's
    (oracle.knn_midpoints_aten: torch.cdist + torch.topk, pt.py:580-583) on every rank, ties included (the lattice case
    has them in every row), and the positions those of the oracle's ATen-mode steps."""
    import oracle
    # E = 4808 >= 64 (k + 1): partial_sort's regime.  (One step from the lattice: its ties are gone after a normalisation.)
    n, D, edges, pos, stream, k, S = _case(n=1203, deg=8, k=6, S=48, iters=1 if kind == "lattice" else 2)
    if kind == "lattice":
        pos = (np.random.default_rng(5).integers(-5, 6, size=(n, D)) / 4.0).astype(np.float32)
    case = (n, D, edges, pos, stream, k, S, "hashed", finish, "cdist")
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    ref = pos
    for t in range(len(stream)):
        last_state = ref
        ref = oracle.step_aten(ref, edges, stream[t], k)
    want = oracle.knn_midpoints_aten(last_state, edges, stream[-1], k)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"knn_w{world}_r{r}.npy"), want), f"rank {r}: not the reference's rows"
        assert np.abs(np.load(tmp_path / f"pos_w{world}_r{r}.npy") - ref).max() < 1e-4
        if kind == "lattice":
            assert int(np.load(tmp_path / f"listed_w{world}_r{r}.npy")[0]) > 0   # the tie path was really taken


def test_hashed_ownership_partitions_the_edges_evenly():
    """GH_EDGES_HASHED: every edge has exactly one owner rank and the shares are near E/world even
    for a u<v edge list (where endpoint-0 ownership gives rank 0 of 2 three quarters of the edges)."""
    import graphem_rapids_amd as gra
    from graphem_rapids_amd.distributed import owned_edge_ids, partition_edges, partition_rows
    n, world = 20000, 4
    edges = gra.random_regular_edges(n, 8, seed=2)
    ids = [owned_edge_ids(edges, *partition_rows(n, world, r)[1:], n) for r in range(world)]
    allids = np.sort(np.concatenate(ids))
    assert np.array_equal(allids, np.arange(len(edges)))
    shares = np.array([len(x) for x in ids]) / len(edges)
    assert np.abs(shares - 1 / world).max() < 0.02
    lo, hi = partition_edges(edges, *partition_rows(n, world, 0)[1:])
    assert (hi - lo) / len(edges) > 0.4      # the imbalance the hashed rule removes


def test_partition_helpers():
    from graphem_rapids_amd.distributed import partition_edges, partition_rows
    assert partition_rows(10, 4, 0) == (3, 0, 3)
    assert partition_rows(10, 4, 3) == (3, 9, 10)
    assert partition_rows(5, 8, 7) == (1, 5, 5)          # more ranks than rows: empty shard
    e = np.array([[0, 1], [0, 5], [2, 3], [2, 9], [7, 8]])
    assert partition_edges(e, 0, 2) == (0, 2)
    assert partition_edges(e, 2, 7) == (2, 4)
    assert partition_edges(e, 7, 10) == (4, 5)
    covered = sum(b - a for a, b in (partition_edges(e, lo, hi) for lo, hi in [(0, 3), (3, 6), (6, 10)]))
    assert covered == len(e)
    with pytest.raises(ValueError):
        partition_edges(np.array([[3, 4], [1, 2]]), 0, 5)
