"""Row-partitioned layout across the GPUs of one node (the reference has no
multi-GPU code, so this is new functionality whose oracle is "N ranks == 1 rank").

One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).  Rank r owns the
vertex rows [r*chunk, (r+1)*chunk) and a disjoint share of the edges: by default each edge
belongs to one of its two endpoints, picked by a fixed hash of the edge id, and so to the rank
holding that endpoint's row (`edge_ownership="hashed"`: ~E/world edges per rank whatever the
vertex numbering); with `edge_ownership="range"` a rank owns the edges whose first endpoint it
holds (a contiguous range of the sorted edge list -- for a u<v edge list rank 0 then holds
most of the edges).  Every rank holds all n positions and the whole edge list.  Per iteration:

    part 1  (local)   spring pull of own rows + midpoints of own edges; exact KNN of the S
                      sampled midpoints among the OWN edges -> S x (k+1) keys
    gather  (RCCL)    all-gather of the keys                       S*(k+1)*8 B per rank
    part 2  (local)   merge keys -> global KNN; intersection forces (redundant on every rank,
                      O(S*k)); integrate own rows; own column sums
  finish="own" (round 3-4; nothing passes over all n rows, three collectives in a row):
    gather  (RCCL)    all-gather of the ranks' column statistics          18*ld*8 B per rank
    part 3  (local)   normalise the OWN rows into their block of the position array; the per-rank sums
                      are added in rank order, so every rank derives the same mean / std
    gather  (RCCL)    in-place all-gather of the finished position blocks     chunk*D*4 B per rank
                      (without their pad columns when D < ld -- 12 instead of 16 B per row at D = 3 --, then
                      expanded into the position array by gh_step_unpack_rows; chunk*ld*4 B when D == ld)
  finish="overlap" (default; round 5, form D): the LAST collective of finish="gathered" moved to the front
    part 1  (local)   as above; the fused kernel leaves new0 = pos + Fs of the own rows in this rank's block
    gather  (RCCL, 2nd stream + 2nd group)   all-gather of the new0 blocks (chunk*D*4 B per rank) -- IN FLIGHT while
    gather  (RCCL)    all-gather of the keys
    part 2  (local)   merge; intersection forces; own corrections to the statistics; the finished values of the own touched
                      rows (pos + (Fs + Fi), <= 4*S*k of them) into the rank's patch list, behind its statistics
    gather  (RCCL)    all-gather of statistics + patch lists                    ~ (18*ld*8 + min(4*S*k, chunk)*(1+ld)*4) B per rank
    part 3  (local)   wait for the rows; every rank's patch list over the gathered rows; normalise ALL n rows (next set-up in
                      the same launch)
  finish="gathered" (two collectives, every rank normalises all n rows -- 22-51 us per rank at 1 M vertices):
    gather  (RCCL)    in-place all-gather of slots [un-normalised rows | statistics]
    part 3  (local)   normalise ALL n rows from the gathered slots

The loop can also run inside the C library (`native=True`): gh_run_partitioned (csrc/comm.hip) enqueues kernels
and ncclAllGather calls of all iterations on one stream -- RCCL opened by the library itself, its communicator
bootstrapped here by broadcasting rank 0's unique id through torch.distributed (+14 us per iteration at world = 1
against +100 us for the Python-driven step).  It is OPT-IN: ncclCommInitRank / ncclAllGather with more than one rank
have not run on hardware yet (the development boxes have one GPU and RCCL refuses two ranks on one device); the
Python-driven step below, on torch.distributed's own communicator, is the default.

The sample ids are the same on every rank: either passed in, or drawn by the engine's counter
based sampler from (seed, iteration).  The compute engine is injectable so that the collective
choreography can be tested with the gloo backend on CPUs (tests/test_distributed_cpu.py).
"""
import numpy as np
import torch
import torch.distributed as dist


def partition_rows(n, world, rank):
    """Equal chunks of ceil(n / world) rows; the last ranks may own fewer (or no) real rows."""
    chunk = (n + world - 1) // world
    lo = min(n, rank * chunk)
    hi = min(n, lo + chunk)
    return chunk, lo, hi


def partition_edges(edges, row_lo, row_hi):
    """Edge range owned by the rows [row_lo, row_hi): edges are sorted by first endpoint."""
    e0 = np.asarray(edges)[:, 0]
    if len(e0) > 1 and np.any(e0[1:] < e0[:-1]):
        raise ValueError("the edge list must be sorted by first endpoint (CSR row order)")
    return int(np.searchsorted(e0, row_lo, side="left")), int(np.searchsorted(e0, row_hi, side="left"))


def edge_owner_is_second(edge_ids):
    """The hash of the C library's GH_EDGES_HASHED rule (csrc/api.hip, gh_create): True where an
    edge is owned by its second endpoint."""
    x = (np.asarray(edge_ids, dtype=np.uint64) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    return (x >> np.uint64(31)) != 0


LONG_DEG = 128  # csrc/common.h GH_LONG_DEG: rows with more neighbours are hubs


def long_degree(n, n_edges):
    """csrc/common.h gh_long_degree: small dense graphs (<= 2**20 edges, mean degree >= 24) treat every row above 16
    neighbours as a hub."""
    return 16 if (n_edges <= (1 << 20) and 2 * n_edges >= 24 * n) else LONG_DEG


def owned_edge_ids(edges, row_lo, row_hi, n):
    """Ids of the edges a rank with rows [row_lo, row_hi) owns under the hashed rule (vertex numbers as
    given: an engine that reorders vertices internally partitions its internal rows the same way).
    An edge between a hub and a short row belongs to the short row, one between two hubs of different degree to the
    smaller hub; otherwise the hash decides."""
    edges = np.asarray(edges).reshape(-1, 2)
    second = edge_owner_is_second(np.arange(len(edges)))
    if len(edges):
        deg = np.bincount(edges.ravel())
        is_hub = deg > long_degree(n, len(edges))   # n as the engine sees it (trailing isolated vertices count)
        hub_u, hub_v = is_hub[edges[:, 0]], is_hub[edges[:, 1]]
        if is_hub.any():
            du, dv = deg[edges[:, 0]], deg[edges[:, 1]]
            second = np.where(hub_u & hub_v & (du != dv), du > dv, second)   # two hubs: the smaller one owns the edge
            second = np.where(hub_u != hub_v, hub_u, second)
    owner = np.where(second, edges[:, 1], edges[:, 0])
    return np.nonzero((owner >= row_lo) & (owner < row_hi))[0]


class HipShardEngine:
    """The product engine of one rank: libgraphem_hip.so on this rank's GPU and torch tensor views of its device
    buffers.  Python-driven steps run on torch's current stream (stream-ordered with torch.distributed's collectives);
    the native loop runs on the engine's own stream and run_partitioned() returns only when it has drained, so the
    views are safe to read afterwards."""

    def __init__(self, n, D, edges, L_min, k_attr, k_inter, k, S, seed, partition, device_id, knn_distance="exact"):
        from . import _native
        from .embedder_hip import device_view
        self.eng = _native.Engine(n, D, edges, L_min, k_attr, k_inter, k, S, seed=seed, device_id=device_id,
                                  partition=partition, knn_distance=knn_distance)
        self.device = torch.device("cuda", device_id)
        self.eng.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        e = self.eng
        self.ld = e.ld
        rows = e.positions_rows_allocated()
        self.pos = device_view(e.positions_device_ptr(), (rows, e.ld), torch.float32, self.device, e)
        self.key_cols = e.knn_partial_cols()   # k + 1 keys per query; knn_distance='cdist': k + 2 keys and a flag
        self.partial = device_view(e.knn_partial_device_ptr(), (e.S, self.key_cols), torch.int64, self.device, e)
        self.stats = device_view(e.stats_partial_device_ptr(), (e.stats_rows(), e.ld), torch.float64, self.device, e)
        self.gbuf = None

    def gather_layout(self, world, rank, chunk):
        """Places the rank's new rows and statistics in slot `rank` of a (world, slot_bytes) buffer."""
        from .embedder_hip import device_view
        e = self.eng
        e.gather_layout(world, rank, chunk)
        self.gbuf = device_view(e.gather_buffer_device_ptr(), (world, e.gather_slot_bytes()), torch.uint8, self.device, e)
        self.stats = device_view(e.stats_partial_device_ptr(), (e.stats_rows(), e.ld), torch.float64, self.device, e)

    def rank_layout(self, world, rank, chunk, packed=None):
        """finish="own": per-rank statistics gathered into stats_all, position blocks gathered in place in pos.
        packed: force the unpadded block exchange on / off (None: the library's default, on from 2 M vertices)."""
        e = self.eng
        e.rank_layout(world, rank, chunk)
        if packed is not None and ((world > 1 and e.D < e.ld) or not packed):
            e.set_packed_rows(packed)
        self.world, self.rank, self.chunk = world, rank, chunk
        self.stats_all = torch.zeros((world, e.stats_rows(), e.ld), dtype=torch.float64, device=self.device)
        self.pos_blocks = self.pos[: world * chunk].view(world, chunk * e.ld)
        # fewer components than the row stride: the finished blocks travel without their pad columns (12 B per row at D = 3)
        from .embedder_hip import device_view
        ptr = e.rows_packed_device_ptr()
        self.packed_blocks = device_view(ptr, (world, chunk * e.D), torch.float32, self.device, e) if ptr else None

    def overlap_layout(self, world, rank, chunk):
        """finish="overlap" (form D): new0 blocks gathered early into rows_all, statistics into stats_all."""
        from .embedder_hip import device_view
        e = self.eng
        e.overlap_layout(world, rank, chunk)
        self.world, self.rank, self.chunk = world, rank, chunk
        rf = e.rows_all_row_floats()
        self.rows_all = device_view(e.rows_all_device_ptr(), (world, chunk * rf), torch.float32, self.device, e)
        # a rank's block: its statistics rows, then its patch list (the own rows the intersection phase touched, finished)
        self.stats_all = device_view(e.stats_all_device_ptr(), (world, e.stats_all_block_doubles()), torch.float64, self.device, e)
        self.stats = self.stats_all[rank]
        self.side = torch.cuda.Stream(self.device)

    def step_rows_early(self):
        return self.eng.step_rows_early()

    def step_pack_rows(self, stream=None):
        self.eng.step_pack_rows(None if stream is None else stream.cuda_stream)

    def step_finish_overlap(self):
        self.eng.step_finish_overlap()

    def step_unpack_rows(self):
        self.eng.step_unpack_rows()

    def step_finish_own(self, stats_all):
        self.eng.step_finish_own(stats_all.data_ptr(), self.world)

    def step_finish_gathered(self):
        self.eng.step_finish_gathered()

    def set_positions(self, pos):
        self.eng.set_positions(pos)

    def get_positions(self):
        return self.eng.get_positions()

    def step_begin(self, sampled):
        self.eng.step_begin(sampled)

    def step_merge(self, gathered, world):
        self.eng.step_merge(gathered.data_ptr(), world)

    def merged_knn(self):
        """(S, k) neighbour edge ids of the last merge (column 0 dropped, pt.py:421), as a host array."""
        from .embedder_hip import device_view
        e = self.eng
        keys = device_view(e.knn_merged_device_ptr(), (e.S, e.k + 1), torch.int64, self.device, e)
        return (keys[:, 1:] & 0xFFFFFFFF).to(torch.int32).cpu().numpy()

    def step_finish(self):
        self.eng.step_finish()

    def sync(self):
        self.eng.sync()

    # native loop (csrc/comm.hip)
    def comm_init_rccl(self, world, rank, unique_id):
        self.eng.set_stream(0, use_own=True)   # the library's own stream: nothing of torch's is in the loop any more
        self.eng.comm_init_rccl(world, rank, unique_id)

    def run_partitioned(self, iters, sample_stream=None):
        self.eng.run_partitioned(iters, sample_stream)
        self.eng.sync()   # the engine's own stream: torch knows nothing of it, so nothing may be pending when we return


class PartitionedLayout:
    def __init__(self, n, D, edges, L_min=1.0, k_attr=0.2, k_inter=0.5, n_neighbors=10, sample_size=256, seed=0,
                 rank=None, world=None, device_id=0, engine_factory=None, group=None, edge_ownership="auto", native=False,
                 finish="overlap", knn_distance="exact"):
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.group = group
        self.n, self.D = int(n), int(D)
        edges = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 2)
        self.chunk, self.row_lo, self.row_hi = partition_rows(self.n, self.world, self.rank)
        if edge_ownership == "auto":  # balanced shares, and the rule the engine's internal BFS vertex order needs
            edge_ownership = "hashed"
        if edge_ownership not in ("hashed", "range"):
            raise ValueError(f"edge_ownership must be 'auto', 'hashed' or 'range', got {edge_ownership!r}")
        self.edge_ownership = edge_ownership
        if edge_ownership == "hashed":
            part = (self.row_lo, self.row_hi, 0, 0, 1)
        else:
            self.edge_lo, self.edge_hi = partition_edges(edges, self.row_lo, self.row_hi)
            part = (self.row_lo, self.row_hi, self.edge_lo, self.edge_hi, 0)
        factory = engine_factory or HipShardEngine
        if knn_distance not in ("exact", "cdist"):
            raise ValueError(f"knn_distance must be 'exact' or 'cdist', got {knn_distance!r}")
        self.knn_distance = knn_distance
        # knn_distance='cdist' (parity mode, the reference's torch.cdist + torch.topk rows, pt.py:580-583): a rank sends its
        # k + 2 best cdist keys per query and whether they are provably its best; the merge decides the rows whose values
        # are pairwise different and replays partial_sort's heap for the others over all edges, identically on every rank
        self.engine = factory(self.n, self.D, edges, L_min, k_attr, k_inter, n_neighbors, min(sample_size, len(edges)),
                              seed, part, device_id, **({"knn_distance": "cdist"} if knn_distance == "cdist" else {}))
        if finish not in ("own", "gathered", "overlap"):
            raise ValueError(f"finish must be 'own', 'gathered' or 'overlap', got {finish!r}")
        self.finish = finish
        self.rows_group = None
        self.exposed_events = []   # (before, after) CUDA events around the wait for the early all-gather (timing runs)
        self.time_overlap = False
        if finish == "own":
            self.engine.rank_layout(self.world, self.rank, self.chunk)
        elif finish == "overlap":
            self.engine.overlap_layout(self.world, self.rank, self.chunk)
            # the early all-gather is in flight beside the keys' and the statistics': a process group of its own
            if self.world > 1 and dist.is_initialized():
                ranks = list(range(dist.get_world_size())) if group is None else dist.get_process_group_ranks(group)
                self.rows_group = dist.new_group(ranks=ranks)
        else:
            self.engine.gather_layout(self.world, self.rank, self.chunk)
        self.K = n_neighbors + 1
        self.S = min(sample_size, len(edges))
        self.key_cols = getattr(self.engine, "key_cols", self.K)
        self.gathered = torch.empty((self.world, self.S, self.key_cols), dtype=torch.int64, device=self.engine.pos.device)
        # the loop in the C library over its own RCCL communicator: opt-in (module docstring)
        self.native = bool(native) and hasattr(self.engine, "comm_init_rccl")
        if self.native:
            self.native = self._init_native_comm()

    def _init_native_comm(self):
        """Bootstraps the library's own RCCL communicator: rank 0's unique id goes to every rank through
        torch.distributed.  Every step is agreed on by all ranks (a rank that cannot open librccl.so must not leave the
        others waiting inside ncclCommInitRank); on failure all ranks fall back to the Python-driven loop together."""
        from . import _native
        dev = self.engine.pos.device
        uid = torch.zeros(129, dtype=torch.uint8, device=dev)       # 128 bytes of id + 1 "rank 0 could make one"
        if self.rank == 0:
            try:
                raw = _native.comm_unique_id()
                uid[:128].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
                uid[128] = 1
            except RuntimeError:
                pass
        if self.world > 1:
            src = 0 if self.group is None else dist.get_global_rank(self.group, 0)
            dist.broadcast(uid, src=src, group=self.group)
        host = uid.cpu().numpy()
        # every rank must be able to open RCCL BEFORE anybody enters the collective ncclCommInitRank (a rank that fails
        # there would leave the others waiting inside it)
        ready = torch.ones(1, dtype=torch.int32, device=dev)
        if not _native.comm_available():
            ready.zero_()
        if host[128] != 1:
            ready.zero_()
        if self.world > 1:
            dist.all_reduce(ready, op=dist.ReduceOp.MIN, group=self.group)
        if int(ready.item()) != 1:
            return False
        ok = torch.ones(1, dtype=torch.int32, device=dev)
        try:
            self.engine.comm_init_rccl(self.world, self.rank, host[:128].tobytes())
        except (RuntimeError, ValueError):
            ok.zero_()
        if self.world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 1:
            return True
        try:                                   # some rank failed: everybody back onto torch's stream and collectives
            self.engine.eng.comm_destroy()
        except (RuntimeError, ValueError, AttributeError):
            pass
        self.engine.eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        return False

    def set_positions(self, pos):
        self.engine.set_positions(np.ascontiguousarray(pos, dtype=np.float32))

    def get_positions(self):
        return self.engine.get_positions()

    def step(self, sampled=None):
        if self.native:
            self.engine.run_partitioned(1, None if sampled is None else np.asarray(sampled, dtype=np.int32)[None, :])
            return
        e = self.engine
        e.step_begin(sampled)
        if self.finish == "overlap":
            self._step_overlap(e)
            return
        # output in concatenated form (world*S, K): accepted by both the RCCL and the gloo backend
        dist.all_gather_into_tensor(self.gathered.view(self.world * self.S, self.key_cols), e.partial, group=self.group)
        e.step_merge(self.gathered, self.world)
        if self.finish == "own":
            dist.all_gather_into_tensor(e.stats_all.view(-1), e.stats.view(-1), group=self.group)   # concatenated form: RCCL and gloo
            e.step_finish_own(e.stats_all)
            packed = getattr(e, "packed_blocks", None)
            if packed is not None:   # the blocks without their pad columns, then expanded into the position array
                dist.all_gather_into_tensor(packed.view(-1), packed[self.rank], group=self.group)
                e.step_unpack_rows()
                return
            # in-place all-gather of the finished blocks: rank r's rows are block r of the position array
            dist.all_gather_into_tensor(e.pos_blocks.view(-1), e.pos_blocks[self.rank], group=self.group)
            return
        # in-place all-gather of the slots: rank r's new rows + statistics sit in row r of gbuf
        dist.all_gather_into_tensor(e.gbuf.view(-1), e.gbuf[self.rank], group=self.group)
        e.step_finish_gathered()

    def _step_overlap(self, e):
        """Form D: the all-gather of the new0 blocks goes out first -- on the engine's side stream and the rows' own process
        group when there is a GPU, so that it is in flight beside everything up to the statistics."""
        early = e.step_rows_early()
        side = getattr(e, "side", None) if early else None
        rows_group = self.rows_group if self.rows_group is not None else self.group

        def send_rows(stream):
            e.step_pack_rows(stream)
            dist.all_gather_into_tensor(e.rows_all.view(-1), e.rows_all[self.rank], group=rows_group if stream is not None else self.group)

        if early and side is not None:
            side.wait_stream(torch.cuda.current_stream(side.device))   # the fused kernel's new0
            with torch.cuda.stream(side):
                send_rows(side)
        elif early:
            send_rows(None)
        dist.all_gather_into_tensor(self.gathered.view(self.world * self.S, self.key_cols), e.partial, group=self.group)
        e.step_merge(self.gathered, self.world)
        if not early:   # no fused kernel in part 1: the rows travel with their intersection forces in them
            send_rows(None)
        dist.all_gather_into_tensor(e.stats_all.view(-1), e.stats.view(-1), group=self.group)
        if early and side is not None:
            cur = torch.cuda.current_stream(side.device)
            if self.time_overlap:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(cur)
                cur.wait_stream(side)
                b.record(cur)
                self.exposed_events.append((a, b))
            else:
                cur.wait_stream(side)
        e.step_finish_overlap()

    def run(self, iters, sample_stream=None):
        if self.native:
            self.engine.run_partitioned(iters, sample_stream)
            return
        for t in range(iters):
            self.step(None if sample_stream is None else sample_stream[t])

    def sync(self):
        self.engine.sync()
