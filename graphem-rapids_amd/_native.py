"""ctypes binding of libgraphem_hip.so (include/graphem_hip.h).

There is no CPU fallback: if the library is missing or no MI355X is visible the HIP
backend raises.  Nothing here imports the oracle.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgraphem_hip.so")

GH_OK, GH_ERR_INVALID, GH_ERR_RUNTIME, GH_ERR_K_TOO_LARGE, GH_ERR_HIP, GH_ERR_NOMEM = range(6)

# Every symbol include/graphem_hip.h declares.
SYMBOLS = [
    "gh_create", "gh_destroy", "gh_last_error", "gh_set_positions", "gh_get_positions", "gh_positions_device",
    "gh_row_stride", "gh_step", "gh_run", "gh_sync", "gh_spring_forces", "gh_knn_midpoints",
    "gh_intersection_forces", "gh_integrate_normalise", "gh_step_begin", "gh_knn_partial_device", "gh_knn_partial_cols", "gh_knn_merged_device", "gh_rows_packed_device", "gh_step_unpack_rows", "gh_set_packed_rows", "gh_step_merge",
    "gh_stats_partial_device", "gh_step_finish", "gh_timing_enable", "gh_timing_reset", "gh_timing_count",
    "gh_timing_get", "gh_device_count", "gh_version", "gh_knn_last_counts", "gh_set_stream",
    "gh_positions_rows_allocated", "gh_knn_points", "gh_stats_rows", "gh_spmv_symnorm",
    "gh_spectral_last_error", "gh_gather_layout", "gh_gather_buffer_device", "gh_gather_slot_bytes",
    "gh_step_finish_gathered", "gh_vertex_order", "gh_positions_unpadded_device", "gh_radial_topk",
    "gh_comm_unique_id", "gh_comm_init_rccl", "gh_loopback_group_create", "gh_loopback_group_destroy",
    "gh_comm_init_loopback", "gh_comm_destroy", "gh_run_partitioned", "gh_comm_last_error", "gh_debug_stamps",
    "gh_knn_cdist_stats", "gh_rank_layout", "gh_step_finish_own", "gh_comm_available", "gh_selftest_arith",
    "gh_create_f64", "gh_set_positions_f64", "gh_get_positions_f64", "gh_positions_device_f64", "gh_spring_forces_f64",
    "gh_intersection_forces_f64", "gh_trlan_sweep", "gh_knn_ivf_config", "gh_knn_ivf_list_sizes",
    "gh_torch_randperm_prefix", "gh_torch_randperm_isa", "gh_run_torch_sampled", "gh_set_cdist_replay", "gh_sampler_stats",
    "gh_overlap_layout", "gh_rows_all_device", "gh_rows_all_row_floats", "gh_stats_all_device", "gh_stats_all_block_doubles", "gh_step_rows_early",
    "gh_step_pack_rows", "gh_step_finish_overlap",
]


class GhParams(ctypes.Structure):
    _fields_ = [("L_min", ctypes.c_float), ("k_attr", ctypes.c_float), ("k_inter", ctypes.c_float),
                ("n_neighbors", ctypes.c_int32), ("sample_size", ctypes.c_int32), ("seed", ctypes.c_uint64),
                ("reorder", ctypes.c_int32), ("knn_method", ctypes.c_int32), ("knn_distance", ctypes.c_int32),
                ("ivf_lists", ctypes.c_int32), ("ivf_probes", ctypes.c_int32)]


REORDER = {"auto": 0, "off": 1, "bfs": 2}  # gh_params.reorder (include/graphem_hip.h GH_REORDER_*)
KNN_METHOD = {"auto": 0, "scan": 1, "grid": 2, "ivf": 3}  # gh_params.knn_method (GH_KNN_*)
KNN_DISTANCE = {"exact": 0, "cdist": 1}  # gh_params.knn_distance (GH_DIST_*)


class GhPartition(ctypes.Structure):
    _fields_ = [("row_lo", ctypes.c_int64), ("row_hi", ctypes.c_int64),
                ("edge_lo", ctypes.c_int64), ("edge_hi", ctypes.c_int64), ("edge_rule", ctypes.c_int32)]


EDGES_RANGE, EDGES_HASHED = 0, 1  # gh_partition.edge_rule (include/graphem_hip.h)


_lib = None


def load():
    """Load the shared library (does not touch the GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    global LIB_PATH
    if os.environ.get("GRAPHEM_HIP_LIB"):   # A/B runs against another build of the same library
        LIB_PATH = os.environ["GRAPHEM_HIP_LIB"]
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python graphem-rapids_amd/build.py` "
            "(hipcc --offload-arch=gfx950). The HIP backend has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    L.gh_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int, i64, i32, i64, vp, ctypes.POINTER(GhParams),
                            ctypes.POINTER(GhPartition)]
    L.gh_create.restype = ctypes.c_int
    L.gh_destroy.argtypes = [vp]
    L.gh_destroy.restype = None
    L.gh_last_error.argtypes = [vp]
    L.gh_last_error.restype = ctypes.c_char_p
    for name in ("gh_set_positions", "gh_get_positions", "gh_spring_forces", "gh_step", "gh_step_begin"):
        getattr(L, name).argtypes = [vp, vp]
        getattr(L, name).restype = ctypes.c_int
    L.gh_positions_device.argtypes = [vp]
    L.gh_positions_device.restype = vp
    L.gh_row_stride.argtypes = [vp]
    L.gh_row_stride.restype = i32
    L.gh_run.argtypes = [vp, i32, vp]
    L.gh_run.restype = ctypes.c_int
    L.gh_torch_randperm_prefix.argtypes = [vp, i64, i64, i64, i32, vp]
    L.gh_torch_randperm_prefix.restype = ctypes.c_int
    L.gh_torch_randperm_isa.argtypes = []
    L.gh_torch_randperm_isa.restype = ctypes.c_char_p
    L.gh_run_torch_sampled.argtypes = [vp, i32, vp, i64]
    L.gh_run_torch_sampled.restype = ctypes.c_int
    L.gh_radial_topk.argtypes = [vp, i32, vp]
    L.gh_radial_topk.restype = ctypes.c_int
    L.gh_vertex_order.argtypes = [vp, vp]
    L.gh_vertex_order.restype = ctypes.c_int
    L.gh_positions_unpadded_device.argtypes = [vp]
    L.gh_positions_unpadded_device.restype = vp
    L.gh_gather_layout.argtypes = [vp, i32, i32, i64]
    L.gh_gather_layout.restype = ctypes.c_int
    L.gh_rank_layout.argtypes = [vp, i32, i32, i64]
    L.gh_rank_layout.restype = ctypes.c_int
    L.gh_overlap_layout.argtypes = [vp, i32, i32, i64]
    L.gh_overlap_layout.restype = ctypes.c_int
    L.gh_rows_all_device.argtypes = [vp]
    L.gh_rows_all_device.restype = vp
    L.gh_rows_all_row_floats.argtypes = [vp]
    L.gh_rows_all_row_floats.restype = i32
    L.gh_stats_all_device.argtypes = [vp]
    L.gh_stats_all_device.restype = vp
    L.gh_stats_all_block_doubles.argtypes = [vp]
    L.gh_stats_all_block_doubles.restype = i64
    L.gh_step_rows_early.argtypes = [vp]
    L.gh_step_rows_early.restype = i32
    L.gh_step_pack_rows.argtypes = [vp, vp, i32]
    L.gh_step_pack_rows.restype = ctypes.c_int
    L.gh_step_finish_overlap.argtypes = [vp]
    L.gh_step_finish_overlap.restype = ctypes.c_int
    L.gh_step_finish_own.argtypes = [vp, vp, i32]
    L.gh_step_finish_own.restype = ctypes.c_int
    L.gh_create_f64.argtypes = [ctypes.POINTER(vp), ctypes.c_int, i64, i32, i64, vp, ctypes.POINTER(GhParams),
                                ctypes.c_double, ctypes.c_double, ctypes.c_double]
    L.gh_create_f64.restype = ctypes.c_int
    for name in ("gh_set_positions_f64", "gh_get_positions_f64", "gh_spring_forces_f64"):
        getattr(L, name).argtypes = [vp, vp]
        getattr(L, name).restype = ctypes.c_int
    L.gh_positions_device_f64.argtypes = [vp]
    L.gh_positions_device_f64.restype = vp
    L.gh_intersection_forces_f64.argtypes = [vp, vp, vp, vp]
    L.gh_intersection_forces_f64.restype = ctypes.c_int
    L.gh_selftest_arith.argtypes = [ctypes.c_int, ctypes.c_uint64, i64, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    L.gh_selftest_arith.restype = ctypes.c_int
    L.gh_comm_available.argtypes = []
    L.gh_comm_available.restype = i32
    L.gh_gather_buffer_device.argtypes = [vp]
    L.gh_gather_buffer_device.restype = vp
    L.gh_gather_slot_bytes.argtypes = [vp]
    L.gh_gather_slot_bytes.restype = i64
    for name in ("gh_sync", "gh_step_finish", "gh_step_finish_gathered", "gh_timing_reset"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = ctypes.c_int
    L.gh_knn_midpoints.argtypes = [vp, vp, vp]
    L.gh_knn_midpoints.restype = ctypes.c_int
    L.gh_intersection_forces.argtypes = [vp, vp, vp, vp]
    L.gh_intersection_forces.restype = ctypes.c_int
    L.gh_integrate_normalise.argtypes = [vp, vp, vp, vp]
    L.gh_integrate_normalise.restype = ctypes.c_int
    L.gh_knn_partial_device.argtypes = [vp]
    L.gh_knn_partial_device.restype = vp
    L.gh_knn_partial_cols.argtypes = [vp]
    L.gh_knn_partial_cols.restype = i32
    L.gh_knn_merged_device.argtypes = [vp]
    L.gh_knn_merged_device.restype = vp
    L.gh_rows_packed_device.argtypes = [vp]
    L.gh_rows_packed_device.restype = vp
    L.gh_step_unpack_rows.argtypes = [vp]
    L.gh_step_unpack_rows.restype = ctypes.c_int
    L.gh_set_packed_rows.argtypes = [vp, i32]
    L.gh_set_packed_rows.restype = ctypes.c_int
    L.gh_step_merge.argtypes = [vp, vp, i32]
    L.gh_step_merge.restype = ctypes.c_int
    L.gh_stats_partial_device.argtypes = [vp]
    L.gh_stats_partial_device.restype = vp
    L.gh_timing_enable.argtypes = [vp, i32]
    L.gh_timing_enable.restype = ctypes.c_int
    L.gh_timing_count.argtypes = [vp]
    L.gh_timing_count.restype = i32
    L.gh_timing_get.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_double),
                                ctypes.POINTER(i64)]
    L.gh_timing_get.restype = ctypes.c_int
    L.gh_set_stream.argtypes = [vp, vp, i32]
    L.gh_set_stream.restype = ctypes.c_int
    L.gh_positions_rows_allocated.argtypes = [vp]
    L.gh_positions_rows_allocated.restype = i64
    L.gh_spmv_symnorm.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    L.gh_spmv_symnorm.restype = ctypes.c_int
    L.gh_trlan_sweep.argtypes = [vp, i64, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.gh_trlan_sweep.restype = ctypes.c_int
    L.gh_spectral_last_error.argtypes = []
    L.gh_spectral_last_error.restype = ctypes.c_char_p
    L.gh_stats_rows.argtypes = [vp]
    L.gh_stats_rows.restype = i32
    L.gh_knn_points.argtypes = [ctypes.c_int, vp, i64, vp, i64, i32, i32, vp]
    L.gh_knn_points.restype = ctypes.c_int
    L.gh_knn_last_counts.argtypes = [vp, vp, vp, vp]
    L.gh_knn_last_counts.restype = ctypes.c_int
    L.gh_knn_cdist_stats.argtypes = [vp, vp, vp]
    L.gh_knn_cdist_stats.restype = ctypes.c_int
    L.gh_sampler_stats.argtypes = [vp, vp]
    L.gh_sampler_stats.restype = ctypes.c_int
    L.gh_set_cdist_replay.argtypes = [vp, i32]
    L.gh_set_cdist_replay.restype = ctypes.c_int
    L.gh_knn_ivf_config.argtypes = [vp, vp, vp]
    L.gh_knn_ivf_config.restype = ctypes.c_int
    L.gh_knn_ivf_list_sizes.argtypes = [vp, vp, ctypes.c_int32]
    L.gh_knn_ivf_list_sizes.restype = ctypes.c_int
    L.gh_comm_unique_id.argtypes = [vp]
    L.gh_comm_unique_id.restype = ctypes.c_int
    L.gh_comm_init_rccl.argtypes = [vp, i32, i32, vp]
    L.gh_comm_init_rccl.restype = ctypes.c_int
    L.gh_loopback_group_create.argtypes = [i32]
    L.gh_loopback_group_create.restype = vp
    L.gh_loopback_group_destroy.argtypes = [vp]
    L.gh_loopback_group_destroy.restype = None
    L.gh_comm_init_loopback.argtypes = [vp, vp, i32]
    L.gh_comm_init_loopback.restype = ctypes.c_int
    L.gh_comm_destroy.argtypes = [vp]
    L.gh_comm_destroy.restype = ctypes.c_int
    L.gh_run_partitioned.argtypes = [vp, i32, vp]
    L.gh_run_partitioned.restype = ctypes.c_int
    L.gh_comm_last_error.argtypes = []
    L.gh_comm_last_error.restype = ctypes.c_char_p
    L.gh_debug_stamps.argtypes = [vp, vp, i64]
    L.gh_debug_stamps.restype = ctypes.c_int
    L.gh_device_count.argtypes = []
    L.gh_device_count.restype = i32
    L.gh_version.argtypes = []
    L.gh_version.restype = ctypes.c_char_p
    _lib = L
    return L


def raise_for(status, handle):
    """Map a gh_status to the exception type the reference raises for the same condition."""
    if status == GH_OK:
        return
    msg = load().gh_last_error(handle)
    msg = msg.decode() if msg else f"gh_status {status}"
    if status == GH_ERR_INVALID:
        raise ValueError(msg)
    if status == GH_ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Engine:
    """Thin RAII wrapper over a gh_handle."""

    def __init__(self, n, D, edges, L_min, k_attr, k_inter, n_neighbors, sample_size, seed=0, device_id=0,
                 partition=None, reorder="auto", knn_method="auto", knn_distance="exact", dtype="float32", ivf_lists=0,
                 ivf_probes=0):
        """dtype='float64': the engine of csrc/f64.hip -- every phase in double; positions, spring and intersection forces
        cross the boundary as float64 arrays (whole graph only; reorder / knn_method / knn_distance do not apply)."""
        self.lib = load()
        self.handle = ctypes.c_void_p()
        self.n, self.D = int(n), int(D)
        if dtype not in ("float32", "float64"):
            raise ValueError(f"dtype must be 'float32' or 'float64', got {dtype!r}")
        self.f64 = dtype == "float64"
        self.np_dtype = np.float64 if self.f64 else np.float32
        if self.f64 and partition is not None:
            raise ValueError("the float64 engine takes the whole graph (no partition)")
        edges = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 2)
        self.E = edges.shape[0]
        prm = GhParams(float(L_min), float(k_attr), float(k_inter), int(n_neighbors), int(sample_size),
                       int(seed) & 0xFFFFFFFFFFFFFFFF, REORDER[reorder], KNN_METHOD[knn_method], KNN_DISTANCE[knn_distance],
                       int(ivf_lists), int(ivf_probes))
        part = None
        if partition is not None:
            vals = [int(x) for x in partition]  # (row_lo, row_hi, edge_lo, edge_hi[, edge_rule])
            part = ctypes.pointer(GhPartition(*(vals + [EDGES_RANGE] * (5 - len(vals)))))
        if self.f64:
            st = self.lib.gh_create_f64(ctypes.byref(self.handle), int(device_id), self.n, self.D, self.E, ptr(edges),
                                        ctypes.byref(prm), float(L_min), float(k_attr), float(k_inter))
        else:
            st = self.lib.gh_create(ctypes.byref(self.handle), int(device_id), self.n, self.D, self.E, ptr(edges),
                                    ctypes.byref(prm), part)
        if st != GH_OK:
            self.handle = ctypes.c_void_p()
            raise_for(st, None)
        self.k = int(n_neighbors)
        self.S = min(int(sample_size), self.E)
        self.ld = self.lib.gh_row_stride(self.handle)

    def close(self):
        if getattr(self, "handle", None) and self.handle.value:
            self.lib.gh_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-exception-caught
            pass

    def _chk(self, st):
        raise_for(st, self.handle)

    def set_positions(self, pos):
        pos = np.ascontiguousarray(pos, dtype=self.np_dtype)
        if pos.shape != (self.n, self.D):
            raise ValueError(f"positions must have shape {(self.n, self.D)}, got {pos.shape}")
        self._chk((self.lib.gh_set_positions_f64 if self.f64 else self.lib.gh_set_positions)(self.handle, ptr(pos)))

    def get_positions(self):
        out = np.empty((self.n, self.D), dtype=self.np_dtype)
        self._chk((self.lib.gh_get_positions_f64 if self.f64 else self.lib.gh_get_positions)(self.handle, ptr(out)))
        return out

    def _ids(self, sampled):
        if sampled is None:
            return None
        sampled = np.ascontiguousarray(sampled, dtype=np.int32).ravel()
        if self.S < self.E and sampled.shape[0] != self.S:
            raise ValueError(f"expected {self.S} sampled edge ids, got {sampled.shape[0]}")
        return sampled

    def step(self, sampled=None):
        s = self._ids(sampled)
        self._chk(self.lib.gh_step(self.handle, ptr(s)))

    def run(self, iters, sample_stream=None):
        ss = None
        if sample_stream is not None:
            ss = np.ascontiguousarray(sample_stream, dtype=np.int32)
            if self.S < self.E and ss.shape != (iters, self.S):
                raise ValueError(f"sample_stream must have shape {(iters, self.S)}, got {ss.shape}")
        self._chk(self.lib.gh_run(self.handle, int(iters), ptr(ss)))

    def run_torch_sampled(self, iters, rng_state):
        """gh_run_torch_sampled: rng_state = uint8 array of torch.get_rng_state() (5056 bytes), updated in place."""
        if rng_state.dtype != np.uint8 or not rng_state.flags.c_contiguous or not rng_state.flags.writeable:
            raise ValueError("rng_state must be a writable contiguous uint8 array")
        self._chk(self.lib.gh_run_torch_sampled(self.handle, int(iters), ptr(rng_state), rng_state.size))

    def sampler_stats(self):
        """Host ms of the last run_torch_sampled: producer drawing, caller waiting for an upload slot, caller waiting for ids, the call."""
        out = np.zeros(4, dtype=np.float64)
        self._chk(self.lib.gh_sampler_stats(self.handle, ptr(out)))
        return {"draw_ms": out[0], "slot_wait_ms": out[1], "caller_wait_ms": out[2], "call_ms": out[3]}

    def sync(self):
        self._chk(self.lib.gh_sync(self.handle))

    def spring_forces(self):
        F = np.empty((self.n, self.D), dtype=self.np_dtype)
        self._chk((self.lib.gh_spring_forces_f64 if self.f64 else self.lib.gh_spring_forces)(self.handle, ptr(F)))
        return F

    def knn_midpoints(self, sampled=None):
        s = self._ids(sampled)
        knn = np.empty((self.S, self.k), dtype=np.int32)
        self._chk(self.lib.gh_knn_midpoints(self.handle, ptr(s), ptr(knn)))
        return knn

    def intersection_forces(self, sampled, knn):
        s = self._ids(sampled)
        knn = np.ascontiguousarray(knn, dtype=np.int32)
        if knn.shape != (self.S, self.k):
            raise ValueError(f"knn must have shape {(self.S, self.k)}, got {knn.shape}")
        F = np.empty((self.n, self.D), dtype=self.np_dtype)
        self._chk((self.lib.gh_intersection_forces_f64 if self.f64 else self.lib.gh_intersection_forces)(self.handle, ptr(s), ptr(knn), ptr(F)))
        return F

    def integrate_normalise(self, Fs, Fi):
        Fs = np.ascontiguousarray(Fs, dtype=np.float32)
        Fi = np.ascontiguousarray(Fi, dtype=np.float32)
        if Fs.shape != (self.n, self.D) or Fi.shape != (self.n, self.D):
            raise ValueError("force arrays must have the shape of positions")
        out = np.empty((self.n, self.D), dtype=np.float32)
        self._chk(self.lib.gh_integrate_normalise(self.handle, ptr(Fs), ptr(Fi), ptr(out)))
        return out

    # multi-GPU split step
    def step_begin(self, sampled=None):
        s = self._ids(sampled)
        self._chk(self.lib.gh_step_begin(self.handle, ptr(s)))

    def step_merge(self, gathered_ptr, world):
        self._chk(self.lib.gh_step_merge(self.handle, ctypes.c_void_p(gathered_ptr), int(world)))

    def step_finish(self):
        self._chk(self.lib.gh_step_finish(self.handle))

    def set_stream(self, stream_ptr, use_own=False):
        """Enqueue on the given raw HIP stream (0 = the default stream) or back on the engine's own."""
        self._chk(self.lib.gh_set_stream(self.handle, ctypes.c_void_p(stream_ptr), 1 if use_own else 0))

    def positions_rows_allocated(self):
        return int(self.lib.gh_positions_rows_allocated(self.handle))

    def stats_rows(self):
        return int(self.lib.gh_stats_rows(self.handle))

    def radial_topk(self, k):
        """The k vertices farthest from the origin, farthest first (include/graphem_hip.h gh_radial_topk)."""
        out = np.empty(int(k), dtype=np.int32)
        self._chk(self.lib.gh_radial_topk(self.handle, int(k), ptr(out)))
        return out

    def vertex_order(self):
        """order[v] = row of vertex v in the device position array."""
        out = np.empty(self.n, dtype=np.int32)
        self._chk(self.lib.gh_vertex_order(self.handle, ptr(out)))
        return out

    def positions_unpadded_device_ptr(self):
        p = self.lib.gh_positions_unpadded_device(self.handle)
        if not p:
            raise RuntimeError("gh_positions_unpadded_device failed")
        return p

    def gather_layout(self, world, rank, chunk):
        self._chk(self.lib.gh_gather_layout(self.handle, int(world), int(rank), int(chunk)))

    def rank_layout(self, world, rank, chunk):
        self._chk(self.lib.gh_rank_layout(self.handle, int(world), int(rank), int(chunk)))

    # form D (include/graphem_hip.h gh_overlap_layout)
    def overlap_layout(self, world, rank, chunk):
        self._chk(self.lib.gh_overlap_layout(self.handle, int(world), int(rank), int(chunk)))

    def rows_all_device_ptr(self):
        return self.lib.gh_rows_all_device(self.handle)

    def rows_all_row_floats(self):
        return int(self.lib.gh_rows_all_row_floats(self.handle))

    def stats_all_device_ptr(self):
        return self.lib.gh_stats_all_device(self.handle)

    def stats_all_block_doubles(self):
        return int(self.lib.gh_stats_all_block_doubles(self.handle))

    def step_rows_early(self):
        return bool(self.lib.gh_step_rows_early(self.handle))

    def step_pack_rows(self, stream_ptr=None):
        """Own block of new0 -> its packed slot, on the given raw HIP stream (None: the engine's stream)."""
        self._chk(self.lib.gh_step_pack_rows(self.handle, ctypes.c_void_p(stream_ptr or 0), 1 if stream_ptr is None else 0))

    def step_finish_overlap(self):
        self._chk(self.lib.gh_step_finish_overlap(self.handle))

    def step_finish_own(self, stats_all_ptr, world):
        self._chk(self.lib.gh_step_finish_own(self.handle, ctypes.c_void_p(stats_all_ptr), int(world)))

    def rows_packed_device_ptr(self):
        """(world, chunk, D) float32: the finished blocks without pad columns (0 when D == ld or world == 1)."""
        return self.lib.gh_rows_packed_device(self.handle)

    def step_unpack_rows(self):
        self._chk(self.lib.gh_step_unpack_rows(self.handle))

    def set_packed_rows(self, on):
        """Finished blocks travel without pad columns + expansion kernel (default: from 2 M vertices on)."""
        self._chk(self.lib.gh_set_packed_rows(self.handle, 1 if on else 0))

    def gather_buffer_device_ptr(self):
        return self.lib.gh_gather_buffer_device(self.handle)

    def gather_slot_bytes(self):
        return int(self.lib.gh_gather_slot_bytes(self.handle))

    def step_finish_gathered(self):
        self._chk(self.lib.gh_step_finish_gathered(self.handle))

    # the whole partitioned run in one call (csrc/comm.hip)
    def comm_init_rccl(self, world, rank, unique_id):
        """unique_id: the 128 bytes rank 0 got from comm_unique_id(), the same on every rank."""
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        self._chk(self.lib.gh_comm_init_rccl(self.handle, int(world), int(rank), ctypes.cast(buf, ctypes.c_void_p)))

    def comm_init_loopback(self, group, rank):
        self._chk(self.lib.gh_comm_init_loopback(self.handle, group, int(rank)))

    def comm_destroy(self):
        self._chk(self.lib.gh_comm_destroy(self.handle))

    def run_partitioned(self, iters, sample_stream=None):
        ss = None
        if sample_stream is not None:
            ss = np.ascontiguousarray(sample_stream, dtype=np.int32)
            if self.S < self.E and ss.shape != (iters, self.S):
                raise ValueError(f"sample_stream must have shape {(iters, self.S)}, got {ss.shape}")
        self._chk(self.lib.gh_run_partitioned(self.handle, int(iters), ptr(ss)))

    def positions_device_ptr(self):
        return (self.lib.gh_positions_device_f64 if self.f64 else self.lib.gh_positions_device)(self.handle)

    def knn_partial_device_ptr(self):
        return self.lib.gh_knn_partial_device(self.handle)

    def knn_merged_device_ptr(self):
        """(S, k + 1) keys of the global KNN after step_merge (0 before the first merge)."""
        return self.lib.gh_knn_merged_device(self.handle)

    def knn_partial_cols(self):
        """64-bit words per query of the record a rank sends after part 1 of a split step: k + 1 keys; a
        knn_distance='cdist' engine on a row partition sends k + 2 keys and a flag (include/graphem_hip.h)."""
        return int(self.lib.gh_knn_partial_cols(self.handle))

    def stats_partial_device_ptr(self):
        return self.lib.gh_stats_partial_device(self.handle)

    def knn_last_counts(self):
        """(subset_counts, final_counts, overflow) of the last KNN search, each (S,) int32."""
        a, b, c = (np.zeros(self.S, dtype=np.int32) for _ in range(3))
        self._chk(self.lib.gh_knn_last_counts(self.handle, ptr(a), ptr(b), ptr(c)))
        return a, b, c

    def knn_cdist_stats(self):
        """(rows that took the pass over all edges, rows with a tie ATen's nth_element path decides) of the last
        KNN search of a knn_distance='cdist' engine."""
        a, b = ctypes.c_int32(0), ctypes.c_int32(0)
        self._chk(self.lib.gh_knn_cdist_stats(self.handle, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def set_cdist_replay(self, all_ties):
        """The loop of a knn_distance='cdist' engine replays every tie (True) or only those that can change a force (False, default)."""
        self._chk(self.lib.gh_set_cdist_replay(self.handle, 1 if all_ties else 0))

    def knn_ivf_config(self):
        """(lists, probes per query) of a knn_method='ivf' engine; (0, 0) otherwise."""
        a, b = ctypes.c_int32(0), ctypes.c_int32(0)
        self._chk(self.lib.gh_knn_ivf_config(self.handle, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def knn_ivf_list_sizes(self):
        """(lists,) int32 members of every inverted list after the last search."""
        out = np.zeros(self.knn_ivf_config()[0], dtype=np.int32)
        self._chk(self.lib.gh_knn_ivf_list_sizes(self.handle, ptr(out), len(out)))
        return out

    # instrumentation
    def timing_enable(self, on=True):
        self._chk(self.lib.gh_timing_enable(self.handle, 1 if on else 0))

    def timing_reset(self):
        self._chk(self.lib.gh_timing_reset(self.handle))

    def timings(self):
        """{kernel name: (total_ms, launches)} since the last reset."""
        out = {}
        for i in range(self.lib.gh_timing_count(self.handle)):
            name, ms, cnt = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int64()
            self._chk(self.lib.gh_timing_get(self.handle, i, ctypes.byref(name), ctypes.byref(ms), ctypes.byref(cnt)))
            out[name.value.decode()] = (ms.value, cnt.value)
        return out


def knn_points(query, reference, k, device_id=0):
    """(n_query, k) int64 ids of the k nearest reference rows (gh_knn_points)."""
    query = np.ascontiguousarray(query, dtype=np.float32)
    reference = np.ascontiguousarray(reference, dtype=np.float32)
    if query.ndim != 2 or reference.ndim != 2 or query.shape[1] != reference.shape[1]:
        raise ValueError("query and reference must be 2-D with the same number of columns")
    out = np.empty((query.shape[0], int(k)), dtype=np.int64)
    st = load().gh_knn_points(int(device_id), ptr(query), query.shape[0], ptr(reference), reference.shape[0],
                              query.shape[1], int(k), ptr(out))
    raise_for(st, None)
    return out


def torch_randperm_prefix(rng_state, n, S, iters=1):
    """(iters, S) int32 = `iters` successive torch.randperm(n)[:S] of the CPU generator state in `rng_state` (uint8 array
    of torch.get_rng_state(), 5056 bytes), which is moved on in place exactly as those calls would (pure host code)."""
    if rng_state.dtype != np.uint8 or not rng_state.flags.c_contiguous or not rng_state.flags.writeable:
        raise ValueError("rng_state must be a writable contiguous uint8 array")
    out = np.empty((int(iters), int(S)), dtype=np.int32)
    st = load().gh_torch_randperm_prefix(ptr(rng_state), rng_state.size, int(n), int(S), int(iters), ptr(out))
    if st != GH_OK:
        raise ValueError("gh_torch_randperm_prefix: not a torch CPU generator state, or sizes out of range")
    return out


def torch_randperm_isa():
    return load().gh_torch_randperm_isa().decode()


def comm_unique_id():
    """128 bytes identifying a new RCCL communicator (ncclGetUniqueId); rank 0 makes it, every rank gets a copy."""
    buf = ctypes.create_string_buffer(128)
    st = load().gh_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p))
    if st != GH_OK:
        raise RuntimeError(load().gh_comm_last_error().decode())
    return buf.raw


def comm_available():
    """True when librccl.so opens with every entry point the native loop uses (no communicator is made)."""
    return bool(load().gh_comm_available())


def selftest_arith(samples, seed=1, device_id=0):
    """(mismatches of the lean sqrt, of the lean division) against sqrtf and '/' on `samples` operand sets."""
    a, b = ctypes.c_int64(-1), ctypes.c_int64(-1)
    raise_for(load().gh_selftest_arith(int(device_id), int(seed), int(samples), ctypes.byref(a), ctypes.byref(b)), None)
    return a.value, b.value


def device_count():
    return int(load().gh_device_count())
