"""ctypes binding of the CPU oracle (oracle/graphem_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgraphem_oracle.so")
_lib = None

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("graphem_oracle.c", "aten_cdist_topk.cpp", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, i32, f32 = ctypes.c_int64, ctypes.c_int, ctypes.c_float
        L.go_spring_forces.argtypes = [_f32p, i64, i32, _i32p, i64, f32, f32, _f32p]
        L.go_spring_forces.restype = None
        L.go_midpoints.argtypes = [_f32p, i32, _i32p, i64, _f32p]
        L.go_midpoints.restype = None
        L.go_knn_midpoints.argtypes = [_f32p, i32, _i32p, i64, _i32p, i64, i32, _i32p, ctypes.c_void_p]
        L.go_knn_midpoints.restype = i32
        L.go_knn_midpoints_tiled.argtypes = [_f32p, i32, _i32p, i64, _i32p, i64, i32, _i32p]
        L.go_knn_midpoints_tiled.restype = i32
        L.go_knn_midpoints_cdist_mm.argtypes = [_f32p, i32, _i32p, i64, _i32p, i64, i32, _i32p]
        L.go_knn_midpoints_cdist_mm.restype = i32
        L.go_knn_midpoints_aten.argtypes = [_f32p, i32, _i32p, i64, _i32p, i64, i32, _i32p, ctypes.c_void_p]
        L.go_knn_midpoints_aten.restype = i32
        L.go_cdist_values_aten.argtypes = [_f32p, i32, _i32p, i64, _i32p, i64, _f32p]
        L.go_cdist_values_aten.restype = i32
        L.go_intersection_forces.argtypes = [_f32p, i64, i32, _i32p, _i32p, i64, _i32p, i32, f32, _f32p,
                                             ctypes.POINTER(i64)]
        L.go_intersection_forces.restype = None
        L.go_integrate_normalise.argtypes = [_f32p, _f32p, _f32p, i64, i32, i32, _f32p, _f32p, _f32p]
        L.go_integrate_normalise.restype = None
        L.go_step.argtypes = [_f32p, i64, i32, _i32p, i64, _i32p, i64, i32, f32, f32, f32, i32]
        L.go_step.restype = i32
        L.go_run_layout.argtypes = [_f32p, i64, i32, _i32p, i64, _i32p, i64, i32, i32, f32, f32, f32, i32]
        L.go_column_sums.argtypes = [_f32p, i64, i32, _f32p]
        L.go_column_sums.restype = None
        L.go_column_sums_colmajor.argtypes = [_f32p, i64, i32, _f32p]
        L.go_column_sums_colmajor.restype = None
        L.go_run_layout.restype = i32
        L.go_num_threads.restype = i32
        L.go_set_threads.argtypes = [i32]
        L.go_set_threads.restype = None
        _i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
        _i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
        L.go_pull_lists.argtypes = [_i32p, i64, i64, _i64p, _i32p, _i8p]
        L.go_pull_lists.restype = None
        L.go_spring_forces_pull.argtypes = [_f32p, i64, i32, _i64p, _i32p, _i8p, f32, f32, _f32p]
        L.go_spring_forces_pull.restype = None
        L.go_step_omp.argtypes = [_f32p, i64, i32, _i32p, i64, _i64p, _i32p, _i8p, _i32p, i64, i32, f32, f32, f32]
        L.go_step_omp.restype = i32
        _lib = L
    return _lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def spring_forces(pos, edges, L_min=1.0, k_attr=0.2):
    pos, edges = _c(pos, np.float32), _c(edges, np.int32)
    F = np.empty_like(pos)
    lib().go_spring_forces(pos, pos.shape[0], pos.shape[1], edges, edges.shape[0], L_min, k_attr, F)
    return F


def midpoints(pos, edges):
    pos, edges = _c(pos, np.float32), _c(edges, np.int32)
    mid = np.empty((edges.shape[0], pos.shape[1]), dtype=np.float32)
    lib().go_midpoints(pos, pos.shape[1], edges, edges.shape[0], mid)
    return mid


def knn_midpoints(pos, edges, sampled, k, return_dist=False, cdist_mm=False, tiled=False):
    """(S, k) int32 neighbour edge ids; RuntimeError when k+1 > E like torch.topk (pt.py:583).
    tiled=True: the blocked all-cores search of bench.py's baseline (go_knn_midpoints_tiled), same result."""
    pos, edges, sampled = _c(pos, np.float32), _c(edges, np.int32), _c(sampled, np.int32)
    S = sampled.shape[0]
    knn = np.empty((S, k), dtype=np.int32)
    if tiled:
        err = lib().go_knn_midpoints_tiled(pos, pos.shape[1], edges, edges.shape[0], sampled, S, k, knn)
        dist = None
    elif cdist_mm:
        err = lib().go_knn_midpoints_cdist_mm(pos, pos.shape[1], edges, edges.shape[0], sampled, S, k, knn)
        dist = None
    else:
        dist = np.empty((S, k + 1), dtype=np.float32) if return_dist else None
        err = lib().go_knn_midpoints(pos, pos.shape[1], edges, edges.shape[0], sampled, S, k, knn,
                                     dist.ctypes.data if dist is not None else None)
    if err == 1:
        raise RuntimeError("selected index k out of range")
    if err:
        raise MemoryError("oracle allocation failed")
    return (knn, dist) if return_dist else knn


def knn_midpoints_aten(pos, edges, sampled, k, return_col0=False):
    """The reference's own neighbour ids: ATen's cdist (matmul form) + topk, rounding and tie order included
    (oracle/aten_cdist_topk.cpp; pt.py:580-583).  With return_col0 also the id the reference drops as
    "self" (pt.py:421) -- not always the sampled edge, SURVEY quirk Q3."""
    pos, edges, sampled = _c(pos, np.float32), _c(edges, np.int32), _c(sampled, np.int32)
    S = sampled.shape[0]
    knn = np.empty((S, k), dtype=np.int32)
    col0 = np.empty(S, dtype=np.int32)
    err = lib().go_knn_midpoints_aten(pos, pos.shape[1], edges, edges.shape[0], sampled, S, k, knn, col0.ctypes.data)
    if err == 1:
        raise RuntimeError("selected index k out of range")
    return (knn, col0) if return_col0 else knn


def cdist_values_aten(pos, edges, sampled):
    """(S, E) float32: ATen's cdist values of the sampled midpoints against every midpoint (the numbers
    knn_midpoints_aten ranks; oracle/aten_cdist_topk.cpp go_cdist_values_aten)."""
    pos, edges, sampled = _c(pos, np.float32), _c(edges, np.int32), _c(sampled, np.int32)
    out = np.empty((sampled.shape[0], edges.shape[0]), dtype=np.float32)
    lib().go_cdist_values_aten(pos, pos.shape[1], edges, edges.shape[0], sampled, sampled.shape[0], out)
    return out


def step_aten(pos, edges, sampled, k, L_min=1.0, k_attr=0.2, k_inter=0.5, colmajor=False):
    """One iteration with the KNN phase as ATen computes it (knn_midpoints_aten); the other phases as step()."""
    pos = _c(pos, np.float32)
    Fs = spring_forces(pos, edges, L_min, k_attr)
    knn = knn_midpoints_aten(pos, edges, sampled, k)
    Fi = intersection_forces(pos, edges, sampled, knn, k_inter)
    return integrate_normalise(pos, Fs, Fi, colmajor=colmajor)


def intersection_forces(pos, edges, sampled, knn, k_inter=0.5, return_count=False):
    pos, edges = _c(pos, np.float32), _c(edges, np.int32)
    sampled, knn = _c(sampled, np.int32), _c(knn, np.int32)
    F = np.empty_like(pos)
    cnt = ctypes.c_int64(0)
    lib().go_intersection_forces(pos, pos.shape[0], pos.shape[1], edges, sampled, sampled.shape[0], knn,
                                 knn.shape[1], k_inter, F, ctypes.byref(cnt))
    return (F, cnt.value) if return_count else F


def integrate_normalise(pos, Fs, Fi, return_stats=False, colmajor=False):
    pos, Fs, Fi = _c(pos, np.float32), _c(Fs, np.float32), _c(Fi, np.float32)
    out = np.empty_like(pos)
    mean = np.empty(pos.shape[1], dtype=np.float32)
    std = np.empty(pos.shape[1], dtype=np.float32)
    lib().go_integrate_normalise(pos, Fs, Fi, pos.shape[0], pos.shape[1], int(colmajor), out, mean, std)
    return (out, mean, std) if return_stats else out


def step(pos, edges, sampled, k, L_min=1.0, k_attr=0.2, k_inter=0.5, colmajor=False):
    pos = np.array(pos, dtype=np.float32, order="C", copy=True)
    edges, sampled = _c(edges, np.int32), _c(sampled, np.int32)
    err = lib().go_step(pos, pos.shape[0], pos.shape[1], edges, edges.shape[0], sampled, sampled.shape[0], k,
                        L_min, k_attr, k_inter, int(colmajor))
    if err == 1:
        raise RuntimeError("selected index k out of range")
    return pos


def run_layout(pos, edges, sample_stream, k, L_min=1.0, k_attr=0.2, k_inter=0.5, colmajor=False):
    pos = np.array(pos, dtype=np.float32, order="C", copy=True)
    edges, ss = _c(edges, np.int32), _c(sample_stream, np.int32)
    err = lib().go_run_layout(pos, pos.shape[0], pos.shape[1], edges, edges.shape[0], ss, ss.shape[1],
                              ss.shape[0], k, L_min, k_attr, k_inter, int(colmajor))
    if err == 1:
        raise RuntimeError("selected index k out of range")
    return pos


class OmpStepper:
    """bench.py's 'port_omp' baseline: the same algorithm with every phase over all cores (graphem_oracle.c,
    go_step_omp).  The pull lists are built once per graph, like the GPU engine builds its own at creation."""

    def __init__(self, n, edges):
        self.n = int(n)
        self.edges = _c(edges, np.int32)
        E = self.edges.shape[0]
        self.rowptr = np.empty(self.n + 1, dtype=np.int64)
        self.adj = np.empty(2 * E, dtype=np.int32)
        self.sign = np.empty(2 * E, dtype=np.int8)
        lib().go_pull_lists(self.edges, E, self.n, self.rowptr, self.adj, self.sign)

    def spring_forces(self, pos, L_min=1.0, k_attr=0.2):
        pos = _c(pos, np.float32)
        F = np.empty_like(pos)
        lib().go_spring_forces_pull(pos, self.n, pos.shape[1], self.rowptr, self.adj, self.sign, L_min, k_attr, F)
        return F

    def step(self, pos, sampled, k, L_min=1.0, k_attr=0.2, k_inter=0.5):
        pos = np.array(pos, dtype=np.float32, order="C", copy=True)
        sampled = _c(sampled, np.int32)
        err = lib().go_step_omp(pos, self.n, pos.shape[1], self.edges, self.edges.shape[0], self.rowptr, self.adj,
                                self.sign, sampled, sampled.shape[0], k, L_min, k_attr, k_inter)
        if err == 1:
            raise RuntimeError("selected index k out of range")
        return pos


def num_threads():
    return lib().go_num_threads()


def set_threads(n):
    lib().go_set_threads(int(n))


def column_sums(x, colmajor=False):
    """torch.sum(x, dim=0) as ATen's CPU kernel orders it, for a row-major or column-major (n, D) tensor."""
    x = _c(x, np.float32)
    out = np.empty(x.shape[1], dtype=np.float32)
    (lib().go_column_sums_colmajor if colmajor else lib().go_column_sums)(x, x.shape[0], x.shape[1], out)
    return out
