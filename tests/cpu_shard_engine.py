"""TEST INFRASTRUCTURE: a CPU stand-in for one rank's engine, built on the oracle's functions and
numpy, with the same interface as graphem_rapids_amd.distributed.HipShardEngine.  It lets the
multi-rank choreography of PartitionedLayout (partitioning, collective sequence, in-place
position all-gather) run under the gloo backend without a GPU.  Never used by the product."""
import numpy as np
import torch

import oracle


class CpuShardEngine:
    PAD_ROWS = 1024

    def __init__(self, n, D, edges, L_min, k_attr, k_inter, k, S, seed, partition, device_id, knn_distance="exact"):
        self.n, self.D, self.k, self.S = n, D, k, S
        self.cdist = knn_distance == "cdist"   # a rank then sends its k + 2 best cdist keys and a "provably my best" flag
        self.key_cols = k + 3 if self.cdist else k + 1
        self.edges = np.ascontiguousarray(edges, dtype=np.int32)
        self.prm = (L_min, k_attr, k_inter)
        self.row_lo, self.row_hi, edge_lo, edge_hi = partition[:4]
        if len(partition) > 4 and partition[4] == 1:  # hashed ownership (include/graphem_hip.h GH_EDGES_HASHED)
            from graphem_rapids_amd.distributed import owned_edge_ids
            self.own_ids = owned_edge_ids(self.edges, self.row_lo, self.row_hi, n).astype(np.int64)
        else:
            self.own_ids = np.arange(edge_lo, edge_hi, dtype=np.int64)
        self.ld = 4 if D <= 4 else 8 if D <= 8 else 16 if D <= 16 else (D + 3) // 4 * 4
        self.pos = torch.zeros((n + self.PAD_ROWS, self.ld), dtype=torch.float32)
        self.partial = torch.zeros((S, self.key_cols), dtype=torch.int64)
        self.stats = torch.zeros((2, self.ld), dtype=torch.float64)
        self.seed, self.iter = seed, 0
        self.gbuf = None
        self.overlap = False

    def gather_layout(self, world, rank, chunk):
        """Slot = [chunk rows of new positions (ld floats each) | (2, ld) float64 statistics], as the HIP engine."""
        self.world, self.rank, self.chunk = world, rank, chunk
        self.slot = (chunk * self.ld * 4 + 2 * self.ld * 8 + 15) // 16 * 16
        self.gbuf = torch.zeros((world, self.slot), dtype=torch.uint8)

    def rank_layout(self, world, rank, chunk):
        """finish="own": statistics of every rank in stats_all, position blocks gathered in place (HipShardEngine.rank_layout)."""
        self.world, self.rank, self.chunk = world, rank, chunk
        assert world * chunk <= self.pos.shape[0]
        self.stats_all = torch.zeros((world, 2, self.ld), dtype=torch.float64)
        self.pos_blocks = self.pos[: world * chunk].view(world, chunk * self.ld)

    def overlap_layout(self, world, rank, chunk):
        """finish="overlap" (form D): new0 = pos + Fs blocks gathered early; statistics + patch list late
        (HipShardEngine.overlap_layout).  Rows travel without pad columns when D < ld and world > 1, like the HIP engine's.
        A rank's late block: (2, ld) statistics, a count, then up to `cap` records (row, D values) -- the own rows the
        intersection phase touched, finished as pos + (Fs + Fi)."""
        self.world, self.rank, self.chunk = world, rank, chunk
        self.rf = self.D if (self.D < self.ld and world > 1) else self.ld
        self.rows_all = torch.zeros((world, chunk * self.rf), dtype=torch.float32)
        self.cap = max(1, min(4 * self.S * self.k, chunk))
        self.stats_all = torch.zeros((world, 2 * self.ld + 1 + self.cap * (1 + self.D)), dtype=torch.float64)
        self.stats = self.stats_all[rank]
        self.overlap = True

    def step_rows_early(self):
        return True

    def step_pack_rows(self, stream=None):
        pass   # (step_begin wrote the own block in its travelling form)

    def step_finish_overlap(self):
        """Every rank's patch list over the gathered new0 rows, then all n rows normalised from the ranks' statistics."""
        n, D = self.n, self.D
        rows = self.rows_all.numpy().reshape(self.world * self.chunk, self.rf)[:n, :D].copy()
        tot = np.zeros((2, self.ld))
        for r in range(self.world):
            blk = self.stats_all[r].numpy()
            tot += blk[: 2 * self.ld].reshape(2, self.ld)
            cnt = int(blk[2 * self.ld])
            rec = blk[2 * self.ld + 1: 2 * self.ld + 1 + cnt * (1 + D)].reshape(cnt, 1 + D)
            rows[rec[:, 0].astype(np.int64)] = rec[:, 1:].astype(np.float32)
        mean = tot[0, :D] / n
        var = np.maximum((tot[1, :D] - tot[0, :D] * mean) / (n - 1), 0.0)
        sd = np.sqrt(var).astype(np.float32) + np.float32(1e-6)
        self.pos[:n, :D] = torch.from_numpy(((rows - mean.astype(np.float32)) / sd).astype(np.float32))
        self.iter += 1

    def step_finish_own(self, stats_all):
        """The ranks' statistics added in rank order; own rows normalised into their block of pos."""
        tot = np.zeros((2, self.ld))
        for r in range(self.world):
            tot += stats_all[r].numpy()
        n = self.n
        mean = tot[0, : self.D] / n
        var = np.maximum((tot[1, : self.D] - tot[0, : self.D] * mean) / (n - 1), 0.0)
        sd = np.sqrt(var).astype(np.float32) + np.float32(1e-6)
        out = (self.new - mean.astype(np.float32)) / sd
        self.pos[self.row_lo:self.row_hi, : self.D] = torch.from_numpy(out.astype(np.float32))
        self.iter += 1

    def _slot_views(self, r):
        raw = self.gbuf[r].numpy()
        rows = raw[: self.chunk * self.ld * 4].view(np.float32).reshape(self.chunk, self.ld)
        stats = raw[self.chunk * self.ld * 4: self.chunk * self.ld * 4 + 2 * self.ld * 8].view(np.float64).reshape(2, self.ld)
        return rows, stats

    def _p(self):
        return self.pos[: self.n, : self.D].numpy()

    def set_positions(self, pos):
        self.pos[: self.n, : self.D] = torch.from_numpy(np.asarray(pos, dtype=np.float32))

    def get_positions(self):
        return self._p().copy()

    def step_begin(self, sampled):
        E = len(self.edges)
        if self.S >= E:
            sampled = np.arange(E, dtype=np.int32)
        elif sampled is None:  # same ids on every rank from (seed, iteration)
            sampled = np.random.default_rng([self.seed, self.iter]).permutation(E)[: self.S].astype(np.int32)
        self.sampled = np.asarray(sampled, dtype=np.int32)
        p = self._p()
        self.Fs = oracle.spring_forces(p, self.edges, self.prm[0], self.prm[1])[self.row_lo:self.row_hi]
        if self.overlap:   # form D: new0 of the own rows is in its block before anything else happens
            blk = self.rows_all[self.rank].numpy().reshape(self.chunk, self.rf)
            self.new0 = p[self.row_lo:self.row_hi] + self.Fs
            blk[: self.row_hi - self.row_lo, : self.D] = self.new0
        mid = oracle.midpoints(p, self.edges)
        q = mid[self.sampled]
        loc = mid[self.own_ids]
        if self.cdist:   # this rank's k + 2 smallest (cdist value, id) keys among its own edges; always "proven" here
            keys = np.full((self.S, self.k + 3), np.iinfo(np.int64).max, dtype=np.int64)
            if len(self.own_ids):
                v = oracle.cdist_values_aten(p, self.edges, self.sampled)[:, self.own_ids]
                key = (v.view(np.uint32).astype(np.int64) << 32) | self.own_ids[None, :]
                key.sort(axis=1)
                m = min(self.k + 2, key.shape[1])
                keys[:, :m] = key[:, :m]
            keys[:, self.k + 2] = 1 if len(self.own_ids) >= self.k + 2 else 0
            self.partial[:] = torch.from_numpy(keys)
            return
        keys = np.full((self.S, self.k + 1), np.iinfo(np.int64).max, dtype=np.int64)  # INF key pattern 0x7FFF..: sorts last
        if len(loc):
            d2 = ((q[:, None, :] - loc[None, :, :]) ** 2).sum(-1, dtype=np.float32)
            ids = self.own_ids
            key = (d2.view(np.uint32).astype(np.int64) << 32) | ids[None, :]
            key.sort(axis=1)
            m = min(self.k + 1, key.shape[1])
            keys[:, :m] = key[:, :m]
        self.partial[:] = torch.from_numpy(keys)

    def step_merge(self, gathered, world):
        p = self._p()
        if self.cdist:
            # rows whose merged k + 2 smallest values are pairwise different (every rank's list proven): ascending values;
            # the others: partial_sort's heap over all edges = the oracle's ATen rows (csrc/cdist.hip knn_merge_cdist_kernel)
            ga = gathered.numpy()
            flags_ok = (ga[:, :, self.k + 2] == 1).all(axis=0)
            g = np.sort(ga[:, :, : self.k + 2].transpose(1, 0, 2).reshape(self.S, -1), axis=1)[:, : self.k + 2]
            vals = g >> 32
            decided = flags_ok & (vals[:, 1:] != vals[:, :-1]).all(axis=1)
            knn = (g[:, 1: self.k + 1] & 0xFFFFFFFF).astype(np.int32)
            if not decided.all():
                rows = np.nonzero(~decided)[0]
                knn[rows] = oracle.knn_midpoints_aten(p, self.edges, self.sampled[rows], self.k)
            self.last_knn, self.last_listed = knn.copy(), int((~decided).sum())
        else:
            g = gathered.numpy().transpose(1, 0, 2).reshape(self.S, -1)  # (S, world*K)
            g = np.sort(g, axis=1)[:, : self.k + 1]
            knn = (g[:, 1:] & 0xFFFFFFFF).astype(np.int32)               # drop column 0 (pt.py:421)
        Fi_all = oracle.intersection_forces(p, self.edges, self.sampled, knn, self.prm[2])
        Fi = Fi_all[self.row_lo:self.row_hi]
        if self.overlap:   # the own rows' finished values pos + (Fs + Fi): statistics, and the touched ones into the patch list
            new = p[self.row_lo:self.row_hi] + (self.Fs + Fi)
            st = np.zeros((2, self.ld))
            st[0, : self.D] = new.astype(np.float64).sum(0)
            st[1, : self.D] = (new.astype(np.float64) ** 2).sum(0)
            own_touched = np.nonzero((Fi != 0).any(axis=1))[0]
            assert len(own_touched) <= self.cap
            blk = self.stats.numpy()
            blk[: 2 * self.ld] = st.reshape(-1)
            blk[2 * self.ld] = len(own_touched)
            rec = np.concatenate([(own_touched + self.row_lo)[:, None].astype(np.float64), new[own_touched].astype(np.float64)], axis=1)
            blk[2 * self.ld + 1: 2 * self.ld + 1 + rec.size] = rec.reshape(-1)
            return
        tot = self.Fs + Fi
        self.new = p[self.row_lo:self.row_hi] + tot
        st = np.zeros((2, self.ld))
        st[0, : self.D] = self.new.astype(np.float64).sum(0)
        st[1, : self.D] = (self.new.astype(np.float64) ** 2).sum(0)
        self.stats[:] = torch.from_numpy(st)
        if self.gbuf is not None:
            rows, stats = self._slot_views(self.rank)
            rows[: self.row_hi - self.row_lo, : self.D] = self.new
            stats[:] = st

    def step_finish(self):
        st = self.stats.numpy()
        n = self.n
        mean = st[0, : self.D] / n
        var = np.maximum((st[1, : self.D] - st[0, : self.D] * mean) / (n - 1), 0.0)
        sd = np.sqrt(var).astype(np.float32) + np.float32(1e-6)
        out = (self.new - mean.astype(np.float32)) / sd
        self.pos[self.row_lo:self.row_hi, : self.D] = torch.from_numpy(out.astype(np.float32))
        self.iter += 1

    def step_finish_gathered(self):
        """After the slots were all-gathered: normalise all n rows, statistics added in rank order."""
        n = self.n
        tot = np.zeros((2, self.ld))
        for r in range(self.world):
            tot += self._slot_views(r)[1]
        mean = tot[0, : self.D] / n
        var = np.maximum((tot[1, : self.D] - tot[0, : self.D] * mean) / (n - 1), 0.0)
        sd = np.sqrt(var).astype(np.float32) + np.float32(1e-6)
        for r in range(self.world):
            lo, hi = min(n, r * self.chunk), min(n, (r + 1) * self.chunk)
            new = self._slot_views(r)[0][: hi - lo, : self.D]
            self.pos[lo:hi, : self.D] = torch.from_numpy(((new - mean.astype(np.float32)) / sd).astype(np.float32))
        self.iter += 1

    def sync(self):
        pass
