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
//   intersection    f64_intersect_kernel: pt.py:638-774 per candidate pair, double atomics into a dense (n, D) array
//   update          f64_sum_kernel / f64_centre_kernel / f64_scale_kernel: new = pos + (Fs + Fi); column means, then
//                   centred sums of squares (two passes, fixed-order reductions), unbiased std + 1e-6, divide.
// Per-iteration cost is dominated by the search: S * E double distances twice.
#include "common.h"
#include "engine.h"

#include <algorithm>
#include <new>
#include <vector>

struct gh_f64 {
    double *pos = nullptr, *nw = nullptr, *Fs = nullptr, *Fi = nullptr, *mid = nullptr, *io = nullptr;
    double *part = nullptr;       // (blocks, D) partial sums of the reductions
    double *colstat = nullptr;    // (2, D): mean, std + 1e-6
    int32_t *rowptr = nullptr, *adj = nullptr, *edges = nullptr, *sampled = nullptr, *knn = nullptr;
    int32_t *fail = nullptr;      // a query whose boundary ties exceeded the pass-2 buffer (never seen; reported)
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
    for (int j = rowptr[x]; j < rowptr[x + 1]; ++j) {
        const int64_t y = adj[j];
        double s = 0.0;
        for (int d = 0; d < D; ++d) { diff[d] = pos[y * D + d] - pos[x * D + d]; s = fma(diff[d], diff[d], s); }
        const double dist = sqrt(s) + 1e-6;                 // pt.py:623
        const double fm = neg_k * (dist - L_min);           // pt.py:626
        for (int d = 0; d < D; ++d) acc[d] = acc[d] + fm * (diff[d] / dist);   // pt.py:629, 633-634
    }
    for (int d = 0; d < D; ++d) F[x * D + d] = acc[d];
}

__global__ void f64_mid_kernel(const double *__restrict__ pos, const int32_t *__restrict__ edges, int64_t E, int D, double *__restrict__ mid) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= E * D) return;
    const int64_t e = t / D;
    const int d = (int)(t % D);
    mid[t] = (pos[(int64_t)edges[2 * e] * D + d] + pos[(int64_t)edges[2 * e + 1] * D + d]) / 2.0;   // pt.py:785
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
__global__ __launch_bounds__(256) void f64_knn_kernel(const double *__restrict__ mid, int64_t E, int D, const int32_t *__restrict__ sampled,
                                                     int K, int32_t *__restrict__ knn, int32_t *__restrict__ fail) {
    __shared__ double q[F64_MAXD];
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
    auto scan = [&](auto visit) __attribute__((always_inline)) {
        for (int64_t e0 = threadIdx.x; e0 < E; e0 += 256 * F64_UNR) {
            double s[F64_UNR];
#pragma unroll
            for (int u = 0; u < F64_UNR; ++u) s[u] = 0.0;
            for (int d = 0; d < D; ++d) {
                double t[F64_UNR];
#pragma unroll
                for (int u = 0; u < F64_UNR; ++u) {
                    const int64_t e = e0 + (int64_t)u * 256;
                    t[u] = e < E ? mid[e * D + d] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < F64_UNR; ++u) { const double df = q[d] - t[u]; s[u] = fma(df, df, s[u]); }
            }
#pragma unroll
            for (int u = 0; u < F64_UNR; ++u) {
                const int64_t e = e0 + (int64_t)u * 256;
                if (e < E) visit(e, s[u]);
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
__global__ __launch_bounds__(256) void f64_sum_kernel(const double *__restrict__ pos, const double *__restrict__ Fs, const double *__restrict__ Fi,
                                                     int64_t n, int D, double *__restrict__ nw, double *__restrict__ part) {
    __shared__ double red[4];
    for (int d = 0; d < D; ++d) {
        double s = 0.0;
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            const double tot = Fs[i * D + d] + Fi[i * D + d];
            const double v = pos[i * D + d] + tot;
            nw[i * D + d] = v;
            s += v;
        }
        s = gh_wave_sum(s);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) part[(int64_t)blockIdx.x * D + d] = ((red[0] + red[1]) + red[2]) + red[3];
    }
}
// mean from the partial sums (fixed order), centre the rows, per-block sums of squares of the centred values (pt.py:802-803).
__global__ __launch_bounds__(256) void f64_centre_kernel(double *__restrict__ nw, int64_t n, int D, const double *__restrict__ part_in, int nparts,
                                                        double *__restrict__ colstat, double *__restrict__ part_out) {
    __shared__ double mean[F64_MAXD];
    __shared__ double red[4];
    if ((int)threadIdx.x < D) {
        double s = 0.0;
        for (int b = 0; b < nparts; ++b) s += part_in[(int64_t)b * D + threadIdx.x];
        mean[threadIdx.x] = s / (double)n;
        if (blockIdx.x == 0) colstat[threadIdx.x] = mean[threadIdx.x];
    }
    __syncthreads();
    for (int d = 0; d < D; ++d) {
        double s = 0.0;
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            const double c = nw[i * D + d] - mean[d];
            nw[i * D + d] = c;
            s += c * c;
        }
        s = gh_wave_sum(s);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) part_out[(int64_t)blockIdx.x * D + d] = ((red[0] + red[1]) + red[2]) + red[3];
    }
}
// unbiased std + 1e-6 (pt.py:803), divide (pt.py:804).
__global__ __launch_bounds__(256) void f64_scale_kernel(const double *__restrict__ nw, int64_t n, int D, const double *__restrict__ part_in, int nparts,
                                                       double *__restrict__ colstat, double *__restrict__ pos) {
    __shared__ double sd[F64_MAXD];
    if ((int)threadIdx.x < D) {
        double s = 0.0;
        for (int b = 0; b < nparts; ++b) s += part_in[(int64_t)b * D + threadIdx.x];
        sd[threadIdx.x] = sqrt(s / (double)(n - 1)) + 1e-6;
        if (blockIdx.x == 0) colstat[D + threadIdx.x] = sd[threadIdx.x];
    }
    __syncthreads();
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n * D; t += (int64_t)gridDim.x * blockDim.x) pos[t] = nw[t] / sd[t % D];
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
    void *ptrs[] = {f->pos, f->nw, f->Fs, f->Fi, f->mid, f->io, f->part, f->colstat, f->rowptr, f->adj, f->edges, f->sampled, f->knn, f->fail};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete f;
    h->f64 = nullptr;
}

gh_status f64_knn(gh_engine *h, const int32_t *host_ids) {
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
    f64_mid_kernel<<<dim3(f64_grid(h->E * h->D)), dim3(256), 0, h->stream>>>(f->pos, f->edges, h->E, h->D, f->mid);
    f64_knn_kernel<<<dim3((unsigned)h->S), dim3(256), 0, h->stream>>>(f->mid, h->E, h->D, f->sampled, h->K, f->knn, f->fail);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

gh_status f64_step(gh_engine *h, const int32_t *host_ids) {
    gh_f64 *f = h->f64;
    const size_t bytes = sizeof(double) * (size_t)h->n * h->D;
    f64_spring_kernel<<<dim3(f64_grid(h->n)), dim3(256), 0, h->stream>>>(f->pos, h->D, f->rowptr, f->adj, h->n, f->L_min, -f->k_attr, f->Fs);
    GH_HIP(hipMemsetAsync(f->Fi, 0, bytes, h->stream));
    if (h->S > 0 && h->k > 0) {
        GH_TRY_ST(f64_knn(h, host_ids));
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
    f->nblocks = (int)std::min<int64_t>(1024, (n + 255) / 256);
    gh_status st;
    if ((st = f64_alloc(h, &f->pos, nD)) || (st = f64_alloc(h, &f->nw, nD)) || (st = f64_alloc(h, &f->Fs, nD)) || (st = f64_alloc(h, &f->Fi, nD)) ||
        (st = f64_alloc(h, &f->io, nD)) || (st = f64_alloc(h, &f->mid, (size_t)E * D)) || (st = f64_alloc(h, &f->part, (size_t)2 * f->nblocks * D)) ||
        (st = f64_alloc(h, &f->colstat, (size_t)2 * D)) || (st = f64_alloc(h, &f->rowptr, (size_t)n + 1)) || (st = f64_alloc(h, &f->adj, adj.size())) ||
        (st = f64_alloc(h, &f->edges, (size_t)std::max<int64_t>(2 * E, 1))) || (st = f64_alloc(h, &f->sampled, (size_t)std::max<int64_t>(h->S, 1))) ||
        (st = f64_alloc(h, &f->knn, (size_t)std::max<int64_t>(h->S * h->k, 1))) || (st = f64_alloc(h, &f->fail, 1)))
        return bail(st);
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
    f64_spring_kernel<<<dim3(f64_grid(h->n)), dim3(256), 0, h->stream>>>(f->pos, h->D, f->rowptr, f->adj, h->n, f->L_min, -f->k_attr, f->Fs);
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
