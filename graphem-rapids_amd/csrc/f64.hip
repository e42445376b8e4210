// The layout iteration in float64: what the reference computes when it is created with dtype=torch.float64
// (pt.py:56, 372-376; tests/test_pytorch_backend.py:169-181).  A second, plain engine behind the same handle type:
// positions (n, D) doubles without padding, every phase its own kernel, one thread per vertex / edge / pair -- the fp32
// engine's fusion, matrix-pipe pre-filter and reference-order bit tricks have no counterpart here; what counts is that
// every operation of pt.py:595-806 is carried out in double, in the reference's order where an order is visible
// (the two index_add_ passes of the spring forces, the four endpoints of a crossing pair).
//
//   spring forces   f64_spring_kernel: pull lists in the reference's summation order, as the fp32 engine builds them
//   midpoints       f64_mid_kernel: materialised, (E, D) doubles (coalesced reads for the search)
//   KNN             f64_knn_kernel: one workgroup per query, exact squared distances in double.  The K smallest of E
//                   doubles through the fp32 engine's 64-bit-key extraction: pass 1 ranks (distance ROUNDED DOWN to
//                   float, id) keys -- rounding is monotone, so the K smallest doubles are among the elements whose
//                   rounded distance does not exceed the K-th smallest rounded distance; pass 2 collects those (K plus
//                   the few that share the last float) and ranks them on (double distance, id).
//                   From 131072 edges on (round 5) the two passes run over a FILTERED list instead of all E midpoints:
//                   pass 1 over every stride-th midpoint gives an exclusive bound per query (MODE 1); a reference-major
//                   pass parks what lies below it -- two and three components: f64_filter_mf_kernel, the fp32 engine's
//                   split-f16 matrix-pipe pre-filter on the float-rounded midpoints with a threshold that provably covers
//                   every pair whose DOUBLE distance is below the bound, then the double chain on what passes; other
//                   dimensions: f64_filter_kernel, a packed-fp32 pre-check in front of the double chain --; MODE 2 ranks
//                   the parked midpoints (same doubles, same chain).  Rows identical to the full passes.
//   gathers         two to four components: the positions once more with 32-byte rows (f64_pad4_kernel) for the spring
//                   and midpoint kernels -- a neighbour's row is two 16-byte loads from one sector (321 -> 136 us).
//   intersection    f64_intersect_kernel: pt.py:638-774 per candidate pair, double atomics into a dense (n, D) array
//   update          f64_sum_kernel / f64_centre_kernel / f64_scale_kernel: new = pos + (Fs + Fi); column means, then
//                   centred sums of squares (two passes, fixed-order reductions), unbiased std + 1e-6, divide.
// Per-iteration cost at a million vertices (rocprofv3, round 5): spring 136 us, filter 134, midpoints 90, thresholds 61,
// update 57, padded copy ~15, ranking 14, intersection 14: 0.52 ms (round 4, two full passes per query: 6.7 ms).
#include "common.h"
#include "engine.h"
#include "scan_core.h"   // the split-f16 matrix-pipe pre-filter of the fp32 engine (gh_mf_query_row / gh_mf_ref_col): reused by the filtered search

#include <algorithm>
#include <new>
#include <vector>

struct gh_f64 {
    double *pos = nullptr, *nw = nullptr, *Fs = nullptr, *Fi = nullptr, *mid = nullptr, *io = nullptr;
    double2 *pos4 = nullptr;      // (n, 2) double2 = (n, 4) doubles: the positions with 32-byte rows (2..4 components; rebuilt every step)
    double *part = nullptr;       // (blocks, D) partial sums of the reductions
    double *colstat = nullptr;    // (2, D): mean, std + 1e-6
    int32_t *rowptr = nullptr, *adj = nullptr, *edges = nullptr, *sampled = nullptr, *knn = nullptr;
    int32_t *fail = nullptr;      // a query whose boundary ties exceeded the pass-2 buffer (never seen; reported)
    // filtered search (graphs from F64_FILTER_MIN_EDGES edges on): per query a bound from every `stride`-th midpoint, one
    // reference-major pass over all midpoints that parks what passes it, the exact ranking over the parked ones
    int64_t stride = 0;           // 0: the two full passes per query (small graphs)
    double *sub = nullptr;        // (ceil(E / stride), D) every stride-th midpoint, contiguous (written beside mid)
    gh_h8 *qA = nullptr;          // (S, 2) the queries' A-operand rows of the matrix-pipe pre-filter (2 or 3 components)
    double *pmax = nullptr;       // (ceil(n / 256)) per-block maxima of |coordinate| over the positions (f64_pad4_kernel)
    double *qaux = nullptr;       // (S, 2) sqrt of that bound (-1: the query takes the full passes), |q|: for the filter's fp32 pre-check
    double *tq = nullptr;         // (S) exclusive bound on the double distance: the float above the K-th smallest rounded-down subset distance
    int32_t *cnt = nullptr;       // (S * F64_CNT_STRIDE) parked midpoints per query, one counter per 128-byte line (may exceed F64_CAND_CAP: then that query takes the full passes)
    double *cand_d = nullptr;     // (S, F64_CAND_CAP) their squared distances ...
    int32_t *cand_i = nullptr;    // ... and edge ids
    int nblocks = 0;
    double L_min = 1.0, k_attr = 0.2, k_inter = 0.5;   // the constructor's constants as doubles (gh_params holds floats)
};

namespace {

#define F64_MAXD 32

__global__ __launch_bounds__(256) void f64_spring_kernel(const double *__restrict__ pos, int D, const int32_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ adj, int64_t n, double L_min, double neg_k,
                                                        double *__restrict__ F) {
    const int64_t x = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (x >= n) return;
    double acc[F64_MAXD], diff[F64_MAXD];
    for (int d = 0; d < D; ++d) acc[d] = 0.0;
    const int jend = rowptr[x + 1];
    double px[F64_MAXD];
    for (int d = 0; d < D; ++d) px[d] = pos[x * D + d];
    auto pull = [&](const double *py) __attribute__((always_inline)) {   // one neighbour, the reference's chain; forces added in list order
        double s = 0.0;
        for (int d = 0; d < D; ++d) { diff[d] = py[d] - px[d]; s = fma(diff[d], diff[d], s); }
        const double dist = sqrt(s) + 1e-6;                 // pt.py:623
        const double fm = neg_k * (dist - L_min);           // pt.py:626
        for (int d = 0; d < D; ++d) acc[d] = acc[d] + fm * (diff[d] / dist);   // pt.py:629, 633-634
    };
    int j = rowptr[x];
    if (D <= 4) {
        // four neighbours' rows in flight per thread (one at a time: 362 us for 8 M gathers at a million vertices, a dependent
        // memory round trip per neighbour; four: 321; eight: 379 -- the registers cost more waves than the loads win)
        for (; j + 4 <= jend; j += 4) {
            double py[4][4];
            int64_t ys[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ys[u] = adj[j + u];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                for (int d = 0; d < D; ++d) py[u][d] = pos[ys[u] * D + d];
#pragma unroll
            for (int u = 0; u < 4; ++u) pull(py[u]);
        }
    }
    for (; j < jend; ++j) pull(pos + (int64_t)adj[j] * D);
    for (int d = 0; d < D; ++d) F[x * D + d] = acc[d];
}

// Two to four components: the positions once more as (n, 4) doubles (32-byte rows, pad 0), so that a neighbour's row is two
// 16-byte loads from one 32-byte sector instead of D separate 8-byte loads from a 24-byte row that straddles sectors (the
// spring kernel's 8 M gathers at a million vertices: 321 us with the (n, D) rows).  Same arithmetic, same order.
// (also the block's largest |coordinate| -> pmax[blockIdx.x]: the matrix-pipe pre-filter's bound on the midpoints' norms)
__global__ __launch_bounds__(256) void f64_pad4_kernel(const double *__restrict__ pos, int64_t n, int D, double2 *__restrict__ pos4,
                                                      double *__restrict__ pmax) {
    __shared__ double red[4];
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (i < n) {
        for (int d = 0; d < D; ++d) v[d] = pos[i * D + d];
        pos4[2 * i] = make_double2(v[0], v[1]);
        pos4[2 * i + 1] = make_double2(v[2], v[3]);
    }
    double m = fmax(fmax(fabs(v[0]), fabs(v[1])), fmax(fabs(v[2]), fabs(v[3])));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0 && pmax) pmax[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
template <int DT>
__global__ __launch_bounds__(256) void f64_spring4_kernel(const double2 *__restrict__ pos4, const int32_t *__restrict__ rowptr,
                                                         const int32_t *__restrict__ adj, int64_t n, double L_min, double neg_k,
                                                         double *__restrict__ F) {
    const int64_t x = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (x >= n) return;
    double acc[DT], diff[DT], px[4];
    { const double2 a = pos4[2 * x], b = pos4[2 * x + 1]; px[0] = a.x; px[1] = a.y; px[2] = b.x; px[3] = b.y; }
#pragma unroll
    for (int d = 0; d < DT; ++d) acc[d] = 0.0;
    auto pull = [&](const double2 &a, const double2 &b) __attribute__((always_inline)) {
        const double py[4] = {a.x, a.y, b.x, b.y};
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < DT; ++d) { diff[d] = py[d] - px[d]; s = fma(diff[d], diff[d], s); }
        const double dist = sqrt(s) + 1e-6;                 // pt.py:623
        const double fm = neg_k * (dist - L_min);           // pt.py:626
#pragma unroll
        for (int d = 0; d < DT; ++d) acc[d] = acc[d] + fm * (diff[d] / dist);   // pt.py:629, 633-634
    };
    int j = rowptr[x];
    const int jend = rowptr[x + 1];
    for (; j + 4 <= jend; j += 4) {
        int64_t ys[4];
        double2 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) ys[u] = adj[j + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = pos4[2 * ys[u]]; if (DT > 2) b[u] = pos4[2 * ys[u] + 1]; else b[u] = make_double2(0.0, 0.0); }
#pragma unroll
        for (int u = 0; u < 4; ++u) pull(a[u], b[u]);
    }
    for (; j < jend; ++j) {
        const int64_t y = adj[j];
        pull(pos4[2 * y], DT > 2 ? pos4[2 * y + 1] : make_double2(0.0, 0.0));
    }
#pragma unroll
    for (int d = 0; d < DT; ++d) F[x * DT + d] = acc[d];
}
// midpoints from the padded rows, one thread per edge
template <int DT>
__global__ __launch_bounds__(256) void f64_mid4_kernel(const double2 *__restrict__ pos4, const int32_t *__restrict__ edges, int64_t E,
                                                      double *__restrict__ mid, int64_t stride, double *__restrict__ sub) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int2 uv = reinterpret_cast<const int2 *>(edges)[e];
    const double2 ua = pos4[2 * (int64_t)uv.x], va = pos4[2 * (int64_t)uv.y];
    double2 ub = make_double2(0.0, 0.0), vb = ub;
    if (DT > 2) { ub = pos4[2 * (int64_t)uv.x + 1]; vb = pos4[2 * (int64_t)uv.y + 1]; }
    const double pu[4] = {ua.x, ua.y, ub.x, ub.y}, pv[4] = {va.x, va.y, vb.x, vb.y};
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        const double v = (pu[d] + pv[d]) / 2.0;   // pt.py:785
        mid[e * DT + d] = v;
        if (sub && e % stride == 0) sub[(e / stride) * DT + d] = v;
    }
}

__global__ void f64_mid_kernel(const double *__restrict__ pos, const int32_t *__restrict__ edges, int64_t E, int D, double *__restrict__ mid,
                               int64_t stride = 0, double *__restrict__ sub = nullptr /* every stride-th midpoint once more, contiguous: the
                               thresholds' subset (read by every query's workgroup: 1 MB in lines of its own instead of one line per midpoint) */) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= E * D) return;
    const int64_t e = t / D;
    const int d = (int)(t % D);
    const double v = (pos[(int64_t)edges[2 * e] * D + d] + pos[(int64_t)edges[2 * e + 1] * D + d]) / 2.0;   // pt.py:785
    mid[t] = v;
    if (sub && e % stride == 0) sub[(e / stride) * D + d] = v;
}

__global__ void f64_sample_kernel(int64_t E, int64_t S, uint64_t seed, uint64_t iter, int mode, int32_t *__restrict__ sampled) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < S) sampled[t] = mode == 2 ? (int32_t)t : gh_sample_id(E, seed, iter, t);
}

// float not above x (x >= 0): the largest float <= x.
__device__ __forceinline__ float f64_round_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = __uint_as_float(__float_as_uint(f) - 1u);
    return f;
}

__device__ __forceinline__ uint64_t f64_min_u64(uint64_t a, uint64_t b) { return b < a ? b : a; }

// One workgroup per query (file comment).  knn[q][0..k): ids of columns 1..k (column 0 dropped, pt.py:421).
#define F64_POOL 1024
#define F64_CAND_CAP 16384
#define F64_CNT_STRIDE 32   /* one list counter per 128-byte line: adjacent counters serialise their returning atomics (3.2 ms of a 3.4 ms iteration at 400 K edges) */
#define F64_FILTER_MIN_EDGES 131072
// MODE 0: the search over all E midpoints (two passes).  MODE 1: pass 1 over every stride-th midpoint only -> tq[q], the
// exclusive bound the filtered pass parks against (the K-th smallest of a subset is no smaller than the K-th smallest of
// all, and rounding down is monotone: whatever ranks among the K smallest has a rounded distance <= the subset's K-th, i.e.
// a distance below the next float); cnt[q] = 0.  MODE 2: both passes over the midpoints f64_filter_kernel parked for the
// query -- the same doubles, computed by the same fma chain -- or, if they overflowed their list, over all E as MODE 0.
template <int MODE>
__global__ __launch_bounds__(256) void f64_knn_kernel(const double *__restrict__ mid, int64_t E, int D, const int32_t *__restrict__ sampled,
                                                     int K, int32_t *__restrict__ knn, int32_t *__restrict__ fail,
                                                     int64_t stride, double *__restrict__ tq, int32_t *__restrict__ cnt,
                                                     const double *__restrict__ cand_d, const int32_t *__restrict__ cand_i,
                                                     const double *__restrict__ sub /* MODE 1: the compact subset */,
                                                     double *__restrict__ qaux = nullptr /* MODE 1: (S, 2) sqrt of the bound (-1: none), |q| */,
                                                     gh_h8 *__restrict__ qA = nullptr /* MODE 1, D <= 3: the query's A-operand row of the matrix-pipe
                                                     pre-filter (f64_filter_mf_kernel) */, const double *__restrict__ pmax = nullptr, int npmax = 0) {
    __shared__ double q[F64_MAXD];
    __shared__ double s_pm[4];
    __shared__ uint64_t wmin[4];
    __shared__ double pool_d[F64_POOL];
    __shared__ int32_t pool_i[F64_POOL];
    __shared__ int pool_n;
    __shared__ uint64_t thr;
    const int64_t qi = blockIdx.x;
    const int64_t qe = sampled[qi];
    if ((int)threadIdx.x < D) q[threadIdx.x] = mid[qe * D + threadIdx.x];
    if (threadIdx.x == 0) pool_n = 0;
    __syncthreads();
    // A scan of the thread's stripe (e = threadIdx.x, + 256, ...): the squared distances of F64_UNR edges at a time, every
    // load of the round requested before the first is used.  (One edge per trip left a single load in flight per thread:
    // 15.6 K dependent memory round trips per scan at a million vertices, 99 of the engine's 100 ms per iteration.)  The
    // chain of one edge is the same fma chain in coordinate order as before.
    constexpr int F64_UNR = 8;
    const int parked = MODE == 2 ? cnt[qi * F64_CNT_STRIDE] : 0;
    const bool from_list = MODE == 2 && parked <= F64_CAND_CAP;
    const int64_t step = MODE == 1 ? stride : 1;           // MODE 1: midpoints 0, stride, 2 stride, ... (`src` holds them contiguously)
    const int64_t NE = MODE == 1 ? (E + stride - 1) / stride : E;
    const double *src = MODE == 1 ? sub : mid;
    auto scan = [&](auto visit) __attribute__((always_inline)) {
        if (from_list) {
            const double *ld = cand_d + qi * F64_CAND_CAP;
            const int32_t *li = cand_i + qi * F64_CAND_CAP;
            for (int i = threadIdx.x; i < parked; i += 256) visit((int64_t)li[i], ld[i]);
            return;
        }
        for (int64_t e0 = threadIdx.x; e0 < NE; e0 += 256 * F64_UNR) {
            double s[F64_UNR];
#pragma unroll
            for (int u = 0; u < F64_UNR; ++u) s[u] = 0.0;
            for (int d = 0; d < D; ++d) {
                double t[F64_UNR];
#pragma unroll
                for (int u = 0; u < F64_UNR; ++u) {
                    const int64_t e = e0 + (int64_t)u * 256;
                    t[u] = e < NE ? src[e * D + d] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < F64_UNR; ++u) { const double df = q[d] - t[u]; s[u] = fma(df, df, s[u]); }
            }
#pragma unroll
            for (int u = 0; u < F64_UNR; ++u) {
                const int64_t e = e0 + (int64_t)u * 256;
                if (e < NE) visit(e * step, s[u]);
            }
        }
    };
    // pass 1: the K-th smallest (rounded-down distance, id) key.  One scan leaves every thread the F64_BEST smallest keys
    // of its stripe, sorted; the block then extracts K times from the threads' current smallest.  A thread whose F64_BEST
    // keys have all been taken (more than F64_BEST of the K best in one stripe of 1/256 of the ids: rare) rescans its
    // stripe for the smallest key above the last one it gave away -- the first version's refill, kept as the fallback.
    constexpr int F64_BEST = 4;
    uint64_t best[F64_BEST];
#pragma unroll
    for (int i = 0; i < F64_BEST; ++i) best[i] = GH_KEY_INF;
    scan([&](int64_t e, double d2) {
        uint64_t key = gh_key(f64_round_down(d2), (uint32_t)e);
        if (key < best[F64_BEST - 1]) {
#pragma unroll
            for (int i = 0; i < F64_BEST; ++i) {   // sorted insertion: the new key bubbles down to its place
                const uint64_t lo = key < best[i] ? key : best[i];
                key = key < best[i] ? best[i] : key;
                best[i] = lo;
            }
        }
    });
    uint64_t mine = best[0], taken = 0;   // taken: the largest key this thread has handed over (exclusive lower bound)
    int given = 0;
    auto refill = [&]() {
        uint64_t m = GH_KEY_INF;
        scan([&](int64_t e, double d2) {
            const uint64_t key = gh_key(f64_round_down(d2), (uint32_t)e);
            if (key > taken && key < m) m = key;
        });
        mine = m;
    };
    uint64_t kth = GH_KEY_INF;
    for (int r = 0; r < K; ++r) {
        uint64_t m = mine;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint64_t o = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(m >> 32), off, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)m, off, 64);
            m = f64_min_u64(m, o);
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) wmin[threadIdx.x >> 6] = m;
        __syncthreads();
        const uint64_t g = f64_min_u64(f64_min_u64(wmin[0], wmin[1]), f64_min_u64(wmin[2], wmin[3]));
        kth = g;
        if (g == GH_KEY_INF) break;
        if (mine == g) {   // keys are unique (the id is part of them)
            taken = g;
            ++given;
            if (given < F64_BEST) {
                uint64_t nx = best[1];
#pragma unroll
                for (int i = 2; i < F64_BEST; ++i) nx = given == i ? best[i] : nx;
                mine = nx;
            } else {
                refill();
            }
        }
    }
    if (MODE == 1) {
        if (qA && D <= 3) {   // the largest |coordinate| of any position, by all threads
            double pm = 0.0;
            for (int b = threadIdx.x; b < npmax; b += 256) pm = fmax(pm, pmax[b]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) pm = fmax(pm, __shfl_xor(pm, off, 64));
            __syncthreads();
            if ((threadIdx.x & 63) == 0) s_pm[threadIdx.x >> 6] = pm;
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const uint32_t vb = (uint32_t)(kth >> 32);
            // not a finite bound (fewer than K subset midpoints, or distances that overflow): the query takes the full passes
            const bool usable = vb < 0x7F800000u;
            const double bound = usable ? (double)__uint_as_float(vb + 1u) : 0.0;
            tq[qi] = bound;
            cnt[qi * F64_CNT_STRIDE] = usable ? 0 : F64_CAND_CAP + 1;
            bool mf_ok = false;
            if (qaux) {   // what the filter's fp32 pre-check needs of this query, once instead of once per workgroup
                double qn = 0.0;
                for (int d = 0; d < D; ++d) qn = fma(q[d], q[d], qn);
                const double qnorm = sqrt(qn) * (1.0 + 1e-15);
                if (qA && D <= 3) {
                    // the matrix-pipe form: threshold b32 of the fp32 difference chain (derivation at f64_filter_kernel) with
                    // M = sqrt(D) * the largest |coordinate| of any position (a midpoint's norm is at most its endpoints')
                    const double M = fmax(fmax(s_pm[0], s_pm[1]), fmax(s_pm[2], s_pm[3])) * sqrt((double)D) * (1.0 + 1e-15);
                    _Float16 row[16];
                    float qf3[3] = {(float)q[0], (float)q[1], D > 2 ? (float)q[2] : 0.0f};
                    float b32 = -1.0f;
                    if (usable) {
                        const double u24 = 5.9604644775390625e-08;
                        const double r = sqrt(bound) * (1.0 + 1e-15) + 2.01 * u24 * (qnorm + M);
                        const double b = r * r * (1.0 + (D + 2) * u24) * (1.0 + 1e-12);
                        b32 = (float)b;
                        if ((double)b32 < b) b32 = __uint_as_float(__float_as_uint(b32) + 1u);
                        if (!(b < 1e30)) b32 = __builtin_inff();
                    }
                    mf_ok = usable && gh_mf_query_row(qf3, 3, b32, row);   // false: outside the f16 range -> a never-pass row
                    if (!usable) { for (int k = 0; k < 16; ++k) row[k] = (_Float16)0.0f; row[12] = (_Float16)GH_MF_NEVER; }
                    gh_h8 lo, hi;
                    for (int k = 0; k < 8; ++k) { lo[k] = row[k]; hi[k] = row[8 + k]; }
                    qA[2 * qi] = lo;
                    qA[2 * qi + 1] = hi;
                    if (usable && !mf_ok) cnt[qi * F64_CNT_STRIDE] = F64_CAND_CAP + 1;   // out of the f16 range: the full passes
                }
                qaux[2 * qi] = (usable && (mf_ok || !(qA && D <= 3))) ? sqrt(bound) * (1.0 + 1e-15) : -1.0;
                qaux[2 * qi + 1] = qnorm;
            }
        }
        return;
    }
    if (threadIdx.x == 0) thr = kth;
    __syncthreads();
    // pass 2: every edge whose rounded distance is <= that of the K-th key, with its double distance
    const uint32_t vmax = (uint32_t)(thr >> 32);
    scan([&](int64_t e, double d2) {
        if (__float_as_uint(f64_round_down(d2)) <= vmax) {
            const int p = atomicAdd(&pool_n, 1);
            if (p < F64_POOL) { pool_d[p] = d2; pool_i[p] = (int32_t)e; }
        }
    });
    __syncthreads();
    const int m = min(pool_n, F64_POOL);
    if (pool_n > F64_POOL && threadIdx.x == 0) *fail = 1;
    // rank on (double distance, id); columns 1 .. K-1 are the neighbours
    for (int i = threadIdx.x; i < m; i += 256) {
        const double di = pool_d[i];
        const int32_t ii = pool_i[i];
        int rank = 0;
        for (int j = 0; j < m; ++j) rank += (pool_d[j] < di || (pool_d[j] == di && pool_i[j] < ii)) ? 1 : 0;
        if (rank >= 1 && rank < K) knn[qi * (K - 1) + rank - 1] = ii;
    }
}

// The filtered pass: a workgroup takes 256 * U consecutive midpoints into registers and runs every query past them (the
// queries' coordinates and bounds staged through LDS, F64_QC at a time): squared distance by the fma chain in coordinate
// order, as f64_knn_kernel computes it; what lies below the query's bound is parked with its distance.  Every midpoint is
// read ONCE per iteration (96 MB at four million edges and three components) where the per-query passes read all of them
// twice per query (49 GB).
#define F64_QC 128
typedef float f64_f2 __attribute__((ext_vector_type(2)));
// PRE: a conservative pre-check in packed fp32 in front of the double chain (two midpoints per v_pk_* instruction, half the
// double chain's issue slots).  With u = 2^-24, q and m rounded to float and the chain fl(sum of fl(q - m)^2) run in fp32:
// |fl(qf_d - mf_d) - (q_d - m_d)| <= 2.01 u (|q_d| + |m_d|), so the float sum s32 of a pair whose exact-chain double
// distance is s obeys  s32 <= (sqrt(s) + 2.01 u (|q| + |m|))^2 (1 + (D + 2) u).  A pair with s < bound therefore has
// s32 <= b32 := (sqrt(bound) + 2.01 u (|q| + M))^2 (1 + (D + 2) u) rounded UP to float, M = the largest |m| of the workgroup's
// own midpoints: whatever fails  s32 <= b32  cannot be below the bound, whatever passes gets the double chain (and is
// parked only if THAT is below the bound).  Underflow in fp32 only makes s32 smaller (passes); a b32 that is not finite
// lets everything pass.  (The expanded form |m|^2 - 2 q.m -- three packed fmas instead of six instructions -- was measured
// too: its error term scales with (|q| + |m|)^2 instead of with the distance, more pairs pass, 337 against 318 us.)
template <int DT, int U, bool PRE>
__global__ __launch_bounds__(256) void f64_filter_kernel(const double *__restrict__ mid, int64_t E, int Drt, const int32_t *__restrict__ sampled,
                                                        int64_t S, const double *__restrict__ tq, int32_t *__restrict__ cnt,
                                                        double *__restrict__ cand_d, int32_t *__restrict__ cand_i) {
    const int D = DT > 0 ? DT : Drt;
    constexpr int DM = DT > 0 ? DT : F64_MAXD;
    __shared__ double qs[F64_QC * DM];
    __shared__ double qb[F64_QC];
    __shared__ float qf[PRE ? F64_QC * DM : 1];
    __shared__ float qb32[PRE ? F64_QC : 1];
    __shared__ double wmax[4];
    double m[U][DM];
    const int64_t e0 = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t e = e0 + (int64_t)u * 256;
        for (int d = 0; d < D; ++d) m[u][d] = e < E ? mid[e * D + d] : 0.0;
    }
    f64_f2 mf[PRE ? U / 2 : 1][DM];
    double M = 0.0;
    if constexpr (PRE) {
        static_assert(U % 2 == 0, "the pre-check packs two midpoints per lane");
#pragma unroll
        for (int r = 0; r < U / 2; ++r)
            for (int d = 0; d < D; ++d) mf[r][d] = (f64_f2){(float)m[2 * r][d], (float)m[2 * r + 1][d]};
        double nm = 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double s = 0.0;
            for (int d = 0; d < D; ++d) s = fma(m[u][d], m[u][d], s);
            nm = fmax(nm, s);
        }
        nm = sqrt(nm) * (1.0 + 1e-15);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nm = fmax(nm, __shfl_xor(nm, off, 64));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = nm;
        __syncthreads();
        M = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
    }
    for (int64_t q0 = 0; q0 < S; q0 += F64_QC) {
        const int nq = (int)min((int64_t)F64_QC, S - q0);
        __syncthreads();
        for (int t = threadIdx.x; t < nq * D; t += 256) {
            const double v = mid[(int64_t)sampled[q0 + t / D] * D + t % D];
            qs[t] = v;
            if constexpr (PRE) qf[t] = (float)v;
        }
        for (int t = threadIdx.x; t < nq; t += 256) qb[t] = cnt[(q0 + t) * F64_CNT_STRIDE] > F64_CAND_CAP ? -1.0 : tq[q0 + t];   // (a query on the full passes parks nothing)
        __syncthreads();
        if constexpr (PRE) {
            for (int t = threadIdx.x; t < nq; t += 256) {
                float b32 = -1.0f;
                if (qb[t] >= 0.0) {
                    double qn = 0.0;
                    for (int d = 0; d < D; ++d) qn = fma(qs[t * D + d], qs[t * D + d], qn);
                    const double u24 = 5.9604644775390625e-08;
                    const double r = sqrt(qb[t]) + 2.01 * u24 * (sqrt(qn) * (1.0 + 1e-15) + M);
                    const double b = r * r * (1.0 + (D + 2) * u24) * (1.0 + 1e-12);
                    b32 = (float)b;
                    if ((double)b32 < b) b32 = __uint_as_float(__float_as_uint(b32) + 1u);   // round up (b > 0)
                    if (!(b < 1e37)) b32 = __builtin_inff();
                }
                qb32[t] = b32;
            }
            __syncthreads();
        }
#pragma unroll 4
        for (int j = 0; j < nq; ++j) {
            unsigned pass = (1u << U) - 1u;   // midpoints of this lane that take the double chain
            if constexpr (PRE) {
                const float b32 = qb32[j];
                pass = 0u;
#pragma unroll
                for (int r = 0; r < U / 2; ++r) {
                    f64_f2 acc = (f64_f2){0.0f, 0.0f};
                    for (int d = 0; d < D; ++d) {
                        const float qd = qf[j * D + d];
                        const f64_f2 df = (f64_f2){qd, qd} - mf[r][d];
                        acc = __builtin_elementwise_fma(df, df, acc);
                    }
                    pass |= (acc.x <= b32 ? 1u : 0u) << (2 * r);
                    pass |= (acc.y <= b32 ? 1u : 0u) << (2 * r + 1);
                }
                if (pass == 0u) continue;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!((pass >> u) & 1u)) continue;
                double s = 0.0;
                for (int d = 0; d < D; ++d) { const double df = qs[j * D + d] - m[u][d]; s = fma(df, df, s); }
                const int64_t e = e0 + (int64_t)u * 256;
                if (s < qb[j] && e < E) {
                    const int p = atomicAdd(&cnt[(q0 + j) * F64_CNT_STRIDE], 1);
                    if (p < F64_CAND_CAP) { cand_d[(q0 + j) * F64_CAND_CAP + p] = s; cand_i[(q0 + j) * F64_CAND_CAP + p] = (int32_t)e; }
                }
            }
        }
    }
}

// The filtered pass for two or three components on the MATRIX PIPE: the fp32 engine's split-f16 pre-filter (scan_core.h:
// F = C0_j - 2 q.m_j - T <= 0 for every pair whose fp32 difference-chain distance is <= the row's threshold, 1024 pairs per
// v_mfma_f32_32x32x16_f16) run on the float-rounded midpoints with the threshold b32 of f64_filter_kernel's derivation
// (every pair whose DOUBLE distance is below the query's bound has an fp32 chain value <= b32): what passes takes the double
// chain and is parked if that is below the bound.  A wave owns 4 column blocks of 32 midpoints (B operands in registers),
// the queries' A rows -- built once per query by the threshold launch -- stream past from LDS; a lane owns one midpoint
// column and 16 query rows of every 32x32 result.  Midpoints outside the f16 range (|coordinate| > 128) are scanned in
// double against every query by the workgroup; queries outside it take the full passes (their rows never pass).
#define F64_MF_QG 256
template <int DT>
__global__ __launch_bounds__(256) void f64_filter_mf_kernel(const double *__restrict__ mid, int64_t E, const int32_t *__restrict__ sampled, int S,
                                                           const double *__restrict__ tq, const double *__restrict__ qaux,
                                                           const gh_h8 *__restrict__ qA, int32_t *__restrict__ cnt,
                                                           double *__restrict__ cand_d, int32_t *__restrict__ cand_i) {
    constexpr int NB = 4, TILE = 512;
    __shared__ gh_h8 qa[2 * F64_MF_QG];
    __shared__ uint16_t badlist[TILE];
    __shared__ int nbad;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = lane & 31, hsel = lane >> 5;
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    if (threadIdx.x == 0) nbad = 0;
    __syncthreads();
    gh_h8 B[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = w * (32 * NB) + b * 32 + col;
        const int64_t e = tile0 + j;
        const bool valid = e < E;
        float mv[3] = {0.0f, 0.0f, 0.0f};
        if (valid) {
#pragma unroll
            for (int d = 0; d < DT; ++d) mv[d] = (float)mid[e * DT + d];
        }
        if (!gh_mf_ref_col(mv, valid, hsel, B[b]) && hsel == 0) badlist[atomicAdd(&nbad, 1)] = (uint16_t)j;
    }
    auto park = [&](int s, int j) {   // the double chain of pair (query s, midpoint j of the tile), as f64_knn_kernel computes it
        if (qaux[2 * s] < 0.0) return;   // a query on the full passes parks nothing
        const int64_t e = tile0 + j, qe = sampled[s];
        double sd = 0.0;
#pragma unroll
        for (int d = 0; d < DT; ++d) { const double df = mid[qe * DT + d] - mid[e * DT + d]; sd = fma(df, df, sd); }
        if (sd < tq[s]) {
            const int p = atomicAdd(&cnt[(int64_t)s * F64_CNT_STRIDE], 1);
            if (p < F64_CAND_CAP) { cand_d[(int64_t)s * F64_CAND_CAP + p] = sd; cand_i[(int64_t)s * F64_CAND_CAP + p] = (int32_t)e; }
        }
    };
    const int nvalid = (int)min((int64_t)TILE, E - tile0);
    for (int s_lo = 0; s_lo < S; s_lo += F64_MF_QG) {
        const int nq = min(S - s_lo, F64_MF_QG);
        __syncthreads();   // the previous group's rows are still being read
        for (int t = threadIdx.x; t < 2 * F64_MF_QG; t += 256) {
            const _Float16 z = (_Float16)0.0f;
            gh_h8 v = (t & 1) ? (gh_h8){z, z, z, z, (_Float16)GH_MF_NEVER, z, z, z} : (gh_h8){z, z, z, z, z, z, z, z};   // padding row: never passes
            if (t < 2 * nq) v = qA[2 * s_lo + t];
            qa[t] = v;
        }
        __syncthreads();
        const int nqb = (nq + 31) / 32;
        const gh_f16x zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int qb = 0; qb < nqb; ++qb) {
            const gh_h8 a = qa[2 * (qb * 32 + col) + hsel];
            gh_f16x f = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B[0], zero, 0, 0, 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                gh_f16x fn = zero;
                if (b + 1 < NB) fn = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B[b + 1], zero, 0, 0, 0);
                int mn = min(__float_as_int(f[0]), __float_as_int(f[1]));
#pragma unroll
                for (int i = 2; i < 16; ++i) mn = min(mn, __float_as_int(f[i]));
                asm volatile("" : "+v"(mn));
                if (mn <= 0) {   // rare: result row (i & 3) + 8 (i >> 2) + 4 * half, column = this lane's midpoint
                    uint32_t m = 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) m = __builtin_amdgcn_alignbit(m, __float_as_uint(f[i]), 31);
                    const int j = w * (32 * NB) + b * 32 + col;
                    if (j < nvalid) {
                        while (m) {
                            const int bit = 31 - __builtin_clz(m);
                            m &= ~(1u << bit);
                            const int i = 15 - bit;
                            const int sq = qb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hsel;
                            if (sq < nq) park(s_lo + sq, j);
                        }
                    }
                }
                f = fn;
            }
        }
        // midpoints outside the f16 range: every query of the group, one pair per thread and step
        const int nb_ = nbad;
        for (int p = threadIdx.x; p < nb_ * nq; p += 256) park(s_lo + p % nq, badlist[p / nq]);
    }
}

__device__ __forceinline__ double f64_orient(const double *a, const double *b, const double *c) {
    return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]);   // pt.py:760-763: coordinates 0 and 1 only
}

__global__ __launch_bounds__(256) void f64_intersect_kernel(const double *__restrict__ pos, int D, const int32_t *__restrict__ edges,
                                                           const int32_t *__restrict__ sampled, const int32_t *__restrict__ knn,
                                                           int64_t S, int k, double k_inter, double *__restrict__ Fi) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= S * k) return;
    const int32_t i = sampled[t / k], j = knn[t];
    if (!(i < j) || D < 2) return;                                                       // pt.py:672
    const int32_t v[4] = {edges[2 * (int64_t)i], edges[2 * (int64_t)i + 1], edges[2 * (int64_t)j], edges[2 * (int64_t)j + 1]};
    if (v[0] == v[2] || v[0] == v[3] || v[1] == v[2] || v[1] == v[3]) return;           // pt.py:685-692
    const double *p1 = pos + (int64_t)v[0] * D, *p2 = pos + (int64_t)v[1] * D, *q1 = pos + (int64_t)v[2] * D, *q2 = pos + (int64_t)v[3] * D;
    const double o1 = f64_orient(p1, p2, q1), o2 = f64_orient(p1, p2, q2), o3 = f64_orient(q1, q2, p1), o4 = f64_orient(q1, q2, p2);
    if (!(o1 * o2 < 0.0 && o3 * o4 < 0.0)) return;                                      // pt.py:768-772
    double diff[F64_MAXD];
    for (int role = 0; role < 4; ++role) {                                              // pt.py:722-734
        const double *x = pos + (int64_t)v[role] * D;
        double s = 0.0;
        for (int d = 0; d < D; ++d) {
            const double cen = (((p1[d] + p2[d]) + q1[d]) + q2[d]) / 4.0;
            diff[d] = x[d] - cen;
            s = fma(diff[d], diff[d], s);
        }
        const double dist = sqrt(s) + 1e-6;
        const double dd = dist * dist;
        for (int d = 0; d < D; ++d) atomicAdd(&Fi[(int64_t)v[role] * D + d], (k_inter * diff[d]) / dd);
    }
}

// new = pos + (Fs + Fi) (pt.py:796-799) and per-block column sums of it.
// Update kernels.  Flat over the (n, D) arrays, T = gridDim.x * 256 threads with T a multiple of D (f64_update_blocks): a
// thread's elements t, t + T, ... all belong to ONE column, so it keeps one running sum and every access is coalesced
// (round 4 walked the arrays column by column, and every workgroup added up the 1024 partial sums of the launch before one
// thread at a time: 107-112 us each for the two launches at a million vertices).  Sums are combined in a fixed order.
__device__ __forceinline__ void f64_block_columns(double s, int D, double *scratch /* shared, 256 */, double *__restrict__ part_row) {
    scratch[threadIdx.x] = s;
    __syncthreads();
    if ((int)threadIdx.x < D) {   // column c: the threads whose global id is c modulo D, in ascending order
        const int first = (int)(((int64_t)threadIdx.x - (int64_t)blockIdx.x * 256 % D + D) % D);
        double tot = 0.0;
        for (int t = first; t < 256; t += D) tot += scratch[t];
        part_row[threadIdx.x] = tot;
    }
    __syncthreads();
}
// totals of the (nparts, D) partial sums -> out[0 .. D) (shared), the same on every workgroup: thread t adds rows t, t + 256,
// ..., then a fixed tree
__device__ __forceinline__ void f64_totals(const double *__restrict__ part, int nparts, int D, double *out, double *scratch) {
    for (int d = 0; d < D; ++d) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nparts; b += 256) s += part[(int64_t)b * D + d];
        scratch[threadIdx.x] = s;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) scratch[threadIdx.x] += scratch[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[d] = scratch[0];
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void f64_sum_kernel(const double *__restrict__ pos, const double *__restrict__ Fs, const double *__restrict__ Fi,
                                                     int64_t n, int D, double *__restrict__ nw, double *__restrict__ part) {
    __shared__ double scratch[256];
    const int64_t T = (int64_t)gridDim.x * 256, total = n * D;
    double s = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += T) {
        const double tot = Fs[t] + Fi[t];          // pt.py:796-799
        const double v = pos[t] + tot;
        nw[t] = v;
        s += v;
    }
    f64_block_columns(s, D, scratch, part + (int64_t)blockIdx.x * D);
}
// mean from the partial sums (fixed order), centre the rows, per-block sums of squares of the centred values (pt.py:802-803).
__global__ __launch_bounds__(256) void f64_centre_kernel(double *__restrict__ nw, int64_t n, int D, const double *__restrict__ part_in, int nparts,
                                                        double *__restrict__ colstat, double *__restrict__ part_out) {
    __shared__ double mean[F64_MAXD];
    __shared__ double scratch[256];
    f64_totals(part_in, nparts, D, mean, scratch);
    if ((int)threadIdx.x < D) {
        mean[threadIdx.x] = mean[threadIdx.x] / (double)n;
        if (blockIdx.x == 0) colstat[threadIdx.x] = mean[threadIdx.x];
    }
    __syncthreads();
    const int64_t T = (int64_t)gridDim.x * 256, total = n * D;
    const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const double mu = mean[t0 % D];
    double s = 0.0;
    for (int64_t t = t0; t < total; t += T) {
        const double c = nw[t] - mu;
        nw[t] = c;
        s += c * c;
    }
    f64_block_columns(s, D, scratch, part_out + (int64_t)blockIdx.x * D);
}
// unbiased std + 1e-6 (pt.py:803), divide (pt.py:804).
__global__ __launch_bounds__(256) void f64_scale_kernel(const double *__restrict__ nw, int64_t n, int D, const double *__restrict__ part_in, int nparts,
                                                       double *__restrict__ colstat, double *__restrict__ pos) {
    __shared__ double sd[F64_MAXD];
    __shared__ double scratch[256];
    f64_totals(part_in, nparts, D, sd, scratch);
    if ((int)threadIdx.x < D) {
        sd[threadIdx.x] = sqrt(sd[threadIdx.x] / (double)(n - 1)) + 1e-6;
        if (blockIdx.x == 0) colstat[D + threadIdx.x] = sd[threadIdx.x];
    }
    __syncthreads();
    const int64_t T = (int64_t)gridDim.x * 256, total = n * D;
    const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const double sdv = sd[t0 % D];
    for (int64_t t = t0; t < total; t += T) pos[t] = nw[t] / sdv;
}

__global__ void f64_from_f32_kernel(const float *__restrict__ src, int64_t count, double *__restrict__ dst) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < count) dst[t] = (double)src[t];
}
__global__ void f64_to_f32_kernel(const double *__restrict__ src, int64_t count, float *__restrict__ dst) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < count) dst[t] = (float)src[t];
}

inline unsigned f64_grid(int64_t total) { return (unsigned)((total + 255) / 256); }

template <typename T>
gh_status f64_alloc(gh_engine *h, T **p, size_t count) {
    if (hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) {
        *p = nullptr;
        h->err = "hipMalloc failed (float64 engine)";
        return GH_ERR_NOMEM;
    }
    return GH_OK;
}

void f64_free(gh_engine *h) {
    gh_f64 *f = h->f64;
    if (!f) return;
    void *ptrs[] = {f->pos, f->nw, f->Fs, f->Fi, f->mid, f->io, f->part, f->colstat, f->rowptr, f->adj, f->edges, f->sampled, f->knn, f->fail,
                    f->tq, f->cnt, f->cand_d, f->cand_i, f->sub, f->qaux, f->qA, f->pmax, f->pos4};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete f;
    h->f64 = nullptr;
}

// (n, 4) copy of the positions for the gathering kernels (2..4 components)
void f64_refresh_pos4(gh_engine *h) {
    gh_f64 *f = h->f64;
    if (f->pos4) f64_pad4_kernel<<<dim3(f64_grid(h->n)), dim3(256), 0, h->stream>>>(f->pos, h->n, h->D, f->pos4, f->pmax);
}
template <typename K2, typename K3, typename K4>
void f64_by_dim(int D, K2 k2, K3 k3, K4 k4) { if (D == 2) k2(); else if (D == 3) k3(); else k4(); }

gh_status f64_knn(gh_engine *h, const int32_t *host_ids, bool pos4_fresh = false) {
    gh_f64 *f = h->f64;
    if ((int64_t)h->K > h->E) { h->err = "selected index k out of range"; return GH_ERR_K_TOO_LARGE; }
    if (h->S >= h->E) {
        f64_sample_kernel<<<dim3(f64_grid(h->S)), dim3(256), 0, h->stream>>>(h->E, h->S, h->prm.seed, h->iter, 2, f->sampled);
    } else if (host_ids) {
        for (int64_t i = 0; i < h->S; ++i)
            if (host_ids[i] < 0 || host_ids[i] >= h->E) { h->err = "sampled edge id out of range"; return GH_ERR_INVALID; }
        GH_HIP(hipMemcpyAsync(f->sampled, host_ids, sizeof(int32_t) * (size_t)h->S, hipMemcpyHostToDevice, h->stream));
        GH_HIP(hipStreamSynchronize(h->stream));
    } else {
        f64_sample_kernel<<<dim3(f64_grid(h->S)), dim3(256), 0, h->stream>>>(h->E, h->S, h->prm.seed, h->iter, 1, f->sampled);
    }
    if (f->pos4) {
        if (!pos4_fresh) f64_refresh_pos4(h);
#define F64_MID4(DD) f64_mid4_kernel<DD><<<dim3(f64_grid(h->E)), dim3(256), 0, h->stream>>>(f->pos4, f->edges, h->E, f->mid, std::max<int64_t>(f->stride, 1), f->stride > 0 ? f->sub : nullptr)
        f64_by_dim(h->D, [&] { F64_MID4(2); }, [&] { F64_MID4(3); }, [&] { F64_MID4(4); });
#undef F64_MID4
    } else {
        f64_mid_kernel<<<dim3(f64_grid(h->E * h->D)), dim3(256), 0, h->stream>>>(f->pos, f->edges, h->E, h->D, f->mid, f->stride, f->sub);
    }
    if (f->stride > 0) {
        f64_knn_kernel<1><<<dim3((unsigned)h->S), dim3(256), 0, h->stream>>>(f->mid, h->E, h->D, f->sampled, h->K, f->knn, f->fail, f->stride, f->tq, f->cnt,
                                                                             f->cand_d, f->cand_i, f->sub, f->qaux, (h->D == 2 || h->D == 3) ? f->qA : nullptr, f->pmax, (int)f64_grid(h->n));
#define F64_FILTER(DD, UU) f64_filter_kernel<DD, UU, (DD > 0 && UU % 2 == 0)><<<dim3((unsigned)((h->E + 256 * UU - 1) / (256 * UU))), dim3(256), 0, h->stream>>>( \
        f->mid, h->E, h->D, f->sampled, h->S, f->tq, f->cnt, f->cand_d, f->cand_i)
#define F64_FILTER3(DD) f64_filter_mf_kernel<DD><<<dim3((unsigned)((h->E + 511) / 512)), dim3(256), 0, h->stream>>>( \
        f->mid, h->E, f->sampled, (int)h->S, f->tq, f->qaux, f->qA, f->cnt, f->cand_d, f->cand_i)
        switch (h->D) {
            case 2: F64_FILTER3(2); break;
            case 3: F64_FILTER3(3); break;
            case 4: F64_FILTER(4, 4); break;
            case 5: F64_FILTER(5, 2); break;
            case 6: F64_FILTER(6, 2); break;
            case 8: F64_FILTER(8, 2); break;
            case 16: F64_FILTER(16, 2); break;
            default: F64_FILTER(0, 1); break;
        }
#undef F64_FILTER
#undef F64_FILTER3
        f64_knn_kernel<2><<<dim3((unsigned)h->S), dim3(256), 0, h->stream>>>(f->mid, h->E, h->D, f->sampled, h->K, f->knn, f->fail, f->stride, f->tq, f->cnt,
                                                                             f->cand_d, f->cand_i, nullptr);
    } else {
        f64_knn_kernel<0><<<dim3((unsigned)h->S), dim3(256), 0, h->stream>>>(f->mid, h->E, h->D, f->sampled, h->K, f->knn, f->fail, 1, nullptr, nullptr,
                                                                             nullptr, nullptr, nullptr);
    }
    GH_LAUNCH_CHECK();
    return GH_OK;
}

// spring forces of the current positions -> f->Fs (refreshes the padded copy of the positions)
void f64_launch_spring(gh_engine *h) {
    gh_f64 *f = h->f64;
    if (f->pos4) {
        f64_refresh_pos4(h);
#define F64_SPR4(DD) f64_spring4_kernel<DD><<<dim3(f64_grid(h->n)), dim3(256), 0, h->stream>>>(f->pos4, f->rowptr, f->adj, h->n, f->L_min, -f->k_attr, f->Fs)
        f64_by_dim(h->D, [&] { F64_SPR4(2); }, [&] { F64_SPR4(3); }, [&] { F64_SPR4(4); });
#undef F64_SPR4
    } else {
        f64_spring_kernel<<<dim3(f64_grid(h->n)), dim3(256), 0, h->stream>>>(f->pos, h->D, f->rowptr, f->adj, h->n, f->L_min, -f->k_attr, f->Fs);
    }
}

gh_status f64_step(gh_engine *h, const int32_t *host_ids) {
    gh_f64 *f = h->f64;
    const size_t bytes = sizeof(double) * (size_t)h->n * h->D;
    f64_launch_spring(h);
    GH_HIP(hipMemsetAsync(f->Fi, 0, bytes, h->stream));
    if (h->S > 0 && h->k > 0) {
        GH_TRY_ST(f64_knn(h, host_ids, true));
        f64_intersect_kernel<<<dim3(f64_grid(h->S * h->k)), dim3(256), 0, h->stream>>>(f->pos, h->D, f->edges, f->sampled, f->knn, h->S, h->k,
                                                                                      f->k_inter, f->Fi);
    }
    const int nb = f->nblocks;
    f64_sum_kernel<<<dim3(nb), dim3(256), 0, h->stream>>>(f->pos, f->Fs, f->Fi, h->n, h->D, f->nw, f->part);
    f64_centre_kernel<<<dim3(nb), dim3(256), 0, h->stream>>>(f->nw, h->n, h->D, f->part, nb, f->colstat, f->part + (size_t)nb * h->D);
    f64_scale_kernel<<<dim3(nb), dim3(256), 0, h->stream>>>(f->nw, h->n, h->D, f->part + (size_t)nb * h->D, nb, f->colstat, f->pos);
    GH_LAUNCH_CHECK();
    h->iter += 1;
    return GH_OK;
}

gh_status f64_check(gh_engine *h) {
    if (!h) return GH_ERR_INVALID;
    if (!h->f64) { h->err = "not a float64 engine (gh_create_f64)"; return GH_ERR_INVALID; }
    if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return GH_ERR_HIP; }
    return GH_OK;
}

}  // namespace

void gh_f64_free(gh_engine *h) { f64_free(h); }

extern "C" gh_status gh_create_f64(gh_handle *out, int device_id, int64_t n, int32_t D, int64_t E, const int32_t *edges, const gh_params *params,
                                   double L_min, double k_attr, double k_inter) {
    if (!out) return GH_ERR_INVALID;
    *out = nullptr;
    auto fail = [&](gh_status st, const std::string &msg) { gh_set_create_error(msg); return st; };
    if (n <= 0) return fail(GH_ERR_INVALID, "Adjacency matrix cannot be empty");
    if (D <= 0) return fail(GH_ERR_INVALID, "Number of components must be positive, got " + std::to_string(D));
    if (D > F64_MAXD) return fail(GH_ERR_INVALID, "the float64 engine takes up to 32 components");
    if (!params) return fail(GH_ERR_INVALID, "params is NULL");
    if (k_attr < 0) return fail(GH_ERR_INVALID, "Attractive force constant k_attr must be non-negative");
    if (E < 0 || (E > 0 && !edges)) return fail(GH_ERR_INVALID, "edges is NULL");
    if (params->n_neighbors < 0 || params->sample_size < 0) return fail(GH_ERR_INVALID, "negative n_neighbors / sample_size");
    if (params->n_neighbors + 1 > 256) return fail(GH_ERR_INVALID, "the float64 engine takes up to 255 neighbours");
    if (E >= ((int64_t)1 << 30) || n >= ((int64_t)1 << 31)) return fail(GH_ERR_INVALID, "graph too large for int32 ids");
    for (int64_t e = 0; e < E; ++e)
        if (edges[2 * e] < 0 || edges[2 * e + 1] < 0 || edges[2 * e] >= n || edges[2 * e + 1] >= n) return fail(GH_ERR_INVALID, "edge endpoint out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GH_ERR_HIP, "no HIP device available");
    if (device_id < 0 || device_id >= ndev) return fail(GH_ERR_RUNTIME, "invalid device ordinal " + std::to_string(device_id));
    gh_engine *h = new (std::nothrow) gh_engine();
    if (!h) return fail(GH_ERR_NOMEM, "out of host memory");
    h->device = device_id;
    h->n = n; h->E = E; h->D = D; h->LD = D;
    h->prm = *params;
    h->k = params->n_neighbors; h->K = h->k + 1; h->Ksel = h->K;
    h->S = std::min<int64_t>(params->sample_size, E);
    h->part = gh_partition{0, n, 0, E, GH_EDGES_RANGE};
    h->rows = n;
    auto bail = [&](gh_status st) { gh_set_create_error(h->err); f64_free(h); if (h->own_stream) (void)hipStreamDestroy(h->own_stream); delete h; return st; };
    if (hipSetDevice(device_id) != hipSuccess) { h->err = "hipSetDevice failed"; return bail(GH_ERR_HIP); }
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) { h->err = "hipStreamCreate failed"; return bail(GH_ERR_HIP); }
    h->stream = h->own_stream;
    h->f64 = new (std::nothrow) gh_f64();
    if (!h->f64) { h->err = "out of host memory"; return bail(GH_ERR_NOMEM); }
    gh_f64 *f = h->f64;
    f->L_min = L_min; f->k_attr = k_attr; f->k_inter = k_inter;
    // pull lists in the reference's summation order (pt.py:633-634): edges where the vertex is endpoint 0, then endpoint 1
    std::vector<int32_t> rowptr((size_t)n + 1, 0), adj((size_t)std::max<int64_t>(2 * E, 1));
    for (int64_t e = 0; e < E; ++e) { rowptr[(size_t)edges[2 * e] + 1]++; rowptr[(size_t)edges[2 * e + 1] + 1]++; }
    for (int64_t i = 0; i < n; ++i) rowptr[(size_t)i + 1] += rowptr[(size_t)i];
    {
        std::vector<int32_t> cur(rowptr.begin(), rowptr.end() - 1);
        for (int64_t e = 0; e < E; ++e) adj[(size_t)cur[(size_t)edges[2 * e]]++] = edges[2 * e + 1];
        for (int64_t e = 0; e < E; ++e) adj[(size_t)cur[(size_t)edges[2 * e + 1]]++] = edges[2 * e];
    }
    const size_t nD = (size_t)n * D;
    {   // update launches: T = nblocks * 256 threads with T a multiple of D (a thread then stays in one column)
        int64_t g = 256, dd = D;
        while (dd) { const int64_t r = g % dd; g = dd; dd = r; }      // gcd(256, D)
        const int64_t m = D / g;                                      // nblocks must be a multiple of this
        int64_t nb = std::min<int64_t>(2048, (n * D + 255) / 256);
        nb = std::max<int64_t>(m, nb / m * m);
        f->nblocks = (int)nb;
    }
    gh_status st;
    if ((st = f64_alloc(h, &f->pos, nD)) || (st = f64_alloc(h, &f->nw, nD)) || (st = f64_alloc(h, &f->Fs, nD)) || (st = f64_alloc(h, &f->Fi, nD)) ||
        (st = f64_alloc(h, &f->io, nD)) || (st = f64_alloc(h, &f->mid, (size_t)E * D)) || (st = f64_alloc(h, &f->part, (size_t)2 * f->nblocks * D)) ||
        (st = f64_alloc(h, &f->colstat, (size_t)2 * D)) || (st = f64_alloc(h, &f->rowptr, (size_t)n + 1)) || (st = f64_alloc(h, &f->adj, adj.size())) ||
        (st = f64_alloc(h, &f->edges, (size_t)std::max<int64_t>(2 * E, 1))) || (st = f64_alloc(h, &f->sampled, (size_t)std::max<int64_t>(h->S, 1))) ||
        (st = f64_alloc(h, &f->knn, (size_t)std::max<int64_t>(h->S * h->k, 1))) || (st = f64_alloc(h, &f->fail, 1)))
        return bail(st);
    // the filtered search: from F64_FILTER_MIN_EDGES edges on, while the parked lists stay below 1 GiB.  stride: about 1024
    // parked midpoints per query (the subset's K-th smallest sits near rank K * stride of all; every parked midpoint is a
    // returning atomic on its query's counter)
    if (E >= F64_FILTER_MIN_EDGES && h->S > 0 && h->k > 0 && (size_t)h->S * F64_CAND_CAP * 12 <= ((size_t)1 << 30)) {
        f->stride = std::min<int64_t>(1024, std::max<int64_t>(16, 1024 / h->K));
        if ((st = f64_alloc(h, &f->sub, (size_t)((E + f->stride - 1) / f->stride) * D)) || (st = f64_alloc(h, &f->tq, (size_t)h->S)) || (st = f64_alloc(h, &f->qaux, (size_t)2 * h->S)) || (st = f64_alloc(h, &f->qA, (size_t)2 * h->S)) ||  (st = f64_alloc(h, &f->cnt, (size_t)h->S * F64_CNT_STRIDE)) ||
            (st = f64_alloc(h, &f->cand_d, (size_t)h->S * F64_CAND_CAP)) || (st = f64_alloc(h, &f->cand_i, (size_t)h->S * F64_CAND_CAP)))
            return bail(st);
    }
    if (D >= 2 && D <= 4 && ((st = f64_alloc(h, &f->pos4, (size_t)2 * n)) || (st = f64_alloc(h, &f->pmax, (size_t)f64_grid(n))))) return bail(st);
    if (hipMemset(f->pos, 0, sizeof(double) * nD) != hipSuccess || hipMemset(f->fail, 0, sizeof(int32_t)) != hipSuccess ||
        hipMemcpy(f->rowptr, rowptr.data(), sizeof(int32_t) * rowptr.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(f->adj, adj.data(), sizeof(int32_t) * (size_t)(2 * E), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(f->edges, edges, sizeof(int32_t) * (size_t)(2 * E), hipMemcpyHostToDevice) != hipSuccess) {
        h->err = "upload of the graph failed";
        return bail(GH_ERR_HIP);
    }
    *out = h;
    return GH_OK;
}

extern "C" gh_status gh_set_positions_f64(gh_handle h, const double *pos) {
    GH_TRY_ST(f64_check(h));
    if (!pos) { h->err = "positions is NULL"; return GH_ERR_INVALID; }
    GH_HIP(hipMemcpyAsync(h->f64->pos, pos, sizeof(double) * (size_t)h->n * h->D, hipMemcpyHostToDevice, h->stream));
    GH_HIP(hipStreamSynchronize(h->stream));
    return GH_OK;
}
static gh_status f64_download(gh_engine *h, const double *d_src, double *host) {
    GH_HIP(hipMemcpyAsync(host, d_src, sizeof(double) * (size_t)h->n * h->D, hipMemcpyDeviceToHost, h->stream));
    GH_HIP(hipStreamSynchronize(h->stream));
    int32_t failed = 0;
    GH_HIP(hipMemcpyAsync(&failed, h->f64->fail, sizeof(failed), hipMemcpyDeviceToHost, h->stream));
    GH_HIP(hipStreamSynchronize(h->stream));
    if (failed) {   // reported once: the flag is cleared
        GH_HIP(hipMemsetAsync(h->f64->fail, 0, sizeof(int32_t), h->stream));
        GH_HIP(hipStreamSynchronize(h->stream));
        h->err = "float64 KNN: more than 1024 midpoints share the float value of the K-th distance";
        return GH_ERR_RUNTIME;
    }
    return GH_OK;
}
extern "C" gh_status gh_get_positions_f64(gh_handle h, double *pos) {
    GH_TRY_ST(f64_check(h));
    if (!pos) { h->err = "positions is NULL"; return GH_ERR_INVALID; }
    return f64_download(h, h->f64->pos, pos);
}
extern "C" double *gh_positions_device_f64(gh_handle h) { return h && h->f64 ? h->f64->pos : nullptr; }

// float32 accessors of the common ABI on a float64 engine: converted on the way.
gh_status gh_f64_set_positions_f32(gh_engine *h, const float *pos) {
    GH_TRY_ST(f64_check(h));
    float *tmp = reinterpret_cast<float *>(h->f64->io);
    GH_HIP(hipMemcpyAsync(tmp, pos, sizeof(float) * (size_t)h->n * h->D, hipMemcpyHostToDevice, h->stream));
    f64_from_f32_kernel<<<dim3(f64_grid(h->n * h->D)), dim3(256), 0, h->stream>>>(tmp, h->n * h->D, h->f64->pos);
    GH_HIP(hipStreamSynchronize(h->stream));
    return GH_OK;
}
gh_status gh_f64_get_positions_f32(gh_engine *h, float *pos) {
    GH_TRY_ST(f64_check(h));
    float *tmp = reinterpret_cast<float *>(h->f64->io);
    f64_to_f32_kernel<<<dim3(f64_grid(h->n * h->D)), dim3(256), 0, h->stream>>>(h->f64->pos, h->n * h->D, tmp);
    GH_HIP(hipMemcpyAsync(pos, tmp, sizeof(float) * (size_t)h->n * h->D, hipMemcpyDeviceToHost, h->stream));
    GH_HIP(hipStreamSynchronize(h->stream));
    return GH_OK;
}
gh_status gh_f64_step(gh_engine *h, const int32_t *sampled) {
    GH_TRY_ST(f64_check(h));
    return f64_step(h, sampled);
}
gh_status gh_f64_run(gh_engine *h, int32_t iters, const int32_t *sample_stream) {
    GH_TRY_ST(f64_check(h));
    for (int32_t t = 0; t < iters; ++t) GH_TRY_ST(f64_step(h, sample_stream && h->S < h->E ? sample_stream + (size_t)t * h->S : nullptr));
    return GH_OK;
}

// per-phase entry points in double (tests): spring forces, neighbour ids, intersection forces for given ids
extern "C" gh_status gh_spring_forces_f64(gh_handle h, double *F) {
    GH_TRY_ST(f64_check(h));
    if (!F) { h->err = "F is NULL"; return GH_ERR_INVALID; }
    gh_f64 *f = h->f64;
    f64_launch_spring(h);
    GH_LAUNCH_CHECK();
    return f64_download(h, f->Fs, F);
}
gh_status gh_f64_knn_midpoints(gh_engine *h, const int32_t *sampled, int32_t *knn) {
    GH_TRY_ST(f64_check(h));
    if (!sampled && h->S < h->E) { h->err = "sampled is NULL"; return GH_ERR_INVALID; }
    GH_TRY_ST(f64_knn(h, sampled));
    GH_HIP(hipMemcpyAsync(knn, h->f64->knn, sizeof(int32_t) * (size_t)h->S * h->k, hipMemcpyDeviceToHost, h->stream));
    GH_HIP(hipStreamSynchronize(h->stream));
    return GH_OK;
}
extern "C" gh_status gh_intersection_forces_f64(gh_handle h, const int32_t *sampled, const int32_t *knn, double *F) {
    GH_TRY_ST(f64_check(h));
    if (!knn || !F || (!sampled && h->S < h->E)) { h->err = "NULL argument"; return GH_ERR_INVALID; }
    gh_f64 *f = h->f64;
    for (int64_t i = 0; i < h->S * h->k; ++i)
        if (knn[i] < 0 || knn[i] >= h->E) { h->err = "neighbour edge id out of range"; return GH_ERR_INVALID; }
    std::vector<int32_t> ids((size_t)h->S);
    for (int64_t i = 0; i < h->S; ++i) ids[(size_t)i] = h->S >= h->E ? (int32_t)i : sampled[i];
    // the engine's stream is non-blocking and a run may still be in flight on it, reading and writing these buffers:
    // everything goes through that stream (the host arrays are done with at the synchronisation of f64_download)
    GH_HIP(hipMemcpyAsync(f->sampled, ids.data(), sizeof(int32_t) * ids.size(), hipMemcpyHostToDevice, h->stream));
    GH_HIP(hipMemcpyAsync(f->knn, knn, sizeof(int32_t) * (size_t)h->S * h->k, hipMemcpyHostToDevice, h->stream));
    GH_HIP(hipMemsetAsync(f->Fi, 0, sizeof(double) * (size_t)h->n * h->D, h->stream));
    f64_intersect_kernel<<<dim3(f64_grid(h->S * h->k)), dim3(256), 0, h->stream>>>(f->pos, h->D, f->edges, f->sampled, f->knn, h->S, h->k,
                                                                                  f->k_inter, f->Fi);
    GH_LAUNCH_CHECK();
    return f64_download(h, f->Fi, F);
}
