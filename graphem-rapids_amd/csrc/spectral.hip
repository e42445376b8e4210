// Spectral initialisation support (reference
// _compute_laplacian_embedding, pt.py:337-379): the sparse operator of a Lanczos iteration for the
// smallest eigenpairs of the normalised Laplacian  L = I - D^-1/2 A D^-1/2  of the symmetrised,
// unweighted graph.  The iteration runs on  B = 2I - L  (spectrum in [0, 2], the wanted pairs are
// the LARGEST of B, where Lanczos converges first):
//     y_i = x_i + s_i * sum_{j in adj(i)} s_j x_j      s = deg^-1/2      (deg_i > 0)
//     y_i = 2 x_i                                       (isolated vertex: L_ii = 0 as scipy's
//                                                        csgraph.laplacian(normed=True) defines it)
// fp64 throughout (the reference solves in fp64 and rounds to fp32 once).  All pointers are DEVICE
// pointers (e.g. torch tensors' data_ptr()); the call is asynchronous on the given HIP stream.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/graphem_hip.h"

namespace {

// One wave per GH_ROWS_PER_WAVE rows would need a degree histogram; graphs here have mean degree
// ~10, so 8 lanes per row (8 rows per wave) keeps the x gathers of a row in one instruction and the
// reduction inside a wave (three xor-shuffles).
__global__ __launch_bounds__(256) void spmv_symnorm_kernel(int64_t n, const int64_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices,
                                                           const double *__restrict__ s,
                                                           const double *__restrict__ x, double *__restrict__ y) {
    const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    double acc = 0.0;
    int64_t beg = 0, end = 0;
    if (row < n) { beg = indptr[row]; end = indptr[row + 1]; }
    for (int64_t j = beg + sub; j < end; j += 8) {
        const int32_t c = indices[j];
        acc += s[c] * x[c];
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (row < n && sub == 0) {
        const double xi = x[row];
        y[row] = end > beg ? xi + s[row] * acc : 2.0 * xi;
    }
}

// ---- the dense half of a Lanczos step: classical Gram-Schmidt, twice, against R rows of the basis -------------------
// (round 3; before, ten torch launches and a host synchronisation per step: 1180 steps took 1.25 s at 100 K vertices)
// W = a chunk of GS_CHUNK elements per workgroup, four per thread in registers; every kernel streams the R x chunk block
// of the basis once.  Reductions over the chunks are two-level and in a fixed order (bit-reproducible runs).
#define GS_CHUNK 512
#define GS_PER (GS_CHUNK / 256)   /* elements per thread */
__device__ __forceinline__ double gs_block_sum(double v, double *red) {   // all threads get the sum; red: 4 doubles of LDS
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}
// partial[block][r] = sum over the chunk of V[r][i] * wv[i] for all R rows, GS_GROUP rows at a time: the loads of a group
// are independent (64 in flight per thread) and its block sums share two barriers.  (A row at a time -- load, barrier,
// reduce -- was a chain of R memory latencies per workgroup: ~0.5 ms per Lanczos step at R = 80.)
#define GS_GROUP 16
__device__ __forceinline__ void gs_dots_chunk(const double *__restrict__ V, int64_t n, int R, const double (&wv)[GS_PER], int64_t i0,
                                              double *__restrict__ partial_row /* partial + block * R */, double (*red)[GS_GROUP]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r0 = 0; r0 < R; r0 += GS_GROUP) {
        double acc[GS_GROUP];
#pragma unroll
        for (int g = 0; g < GS_GROUP; ++g) {
            acc[g] = 0.0;
            if (r0 + g < R) {
                const double *row = V + (int64_t)(r0 + g) * n;
#pragma unroll
                for (int u = 0; u < GS_PER; ++u) acc[g] += i0 + u * 256 < n ? row[i0 + u * 256] * wv[u] : 0.0;
            }
        }
#pragma unroll
        for (int g = 0; g < GS_GROUP; ++g) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[g] += __shfl_xor(acc[g], off, 64);
        }
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int g = 0; g < GS_GROUP; ++g) red[wave][g] = acc[g];
        }
        __syncthreads();
        if ((int)threadIdx.x < GS_GROUP && r0 + (int)threadIdx.x < R)
            partial_row[r0 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    }
}
__global__ __launch_bounds__(256) void gs_dots_kernel(const double *__restrict__ V, int64_t n, int R, const double *__restrict__ w,
                                                     double *__restrict__ partial) {
    __shared__ double red[4][GS_GROUP];
    const int64_t i0 = (int64_t)blockIdx.x * GS_CHUNK + threadIdx.x;
    double wv[GS_PER];
#pragma unroll
    for (int u = 0; u < GS_PER; ++u) wv[u] = i0 + u * 256 < n ? w[i0 + u * 256] : 0.0;
    gs_dots_chunk(V, n, R, wv, i0, partial + (int64_t)blockIdx.x * R, red);
}
// h[r] (+)= sum over blocks of partial[block][r], one workgroup per r
__global__ __launch_bounds__(256) void gs_reduce_kernel(const double *__restrict__ partial, int nblocks, int R, double *__restrict__ h, int accumulate) {
    __shared__ double red[4];
    const int r = blockIdx.x;
    double acc = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) acc += partial[(int64_t)b * R + r];
    const double tot = gs_block_sum(acc, red);
    if (threadIdx.x == 0) h[r] = accumulate ? h[r] + tot : tot;
}
// w -= sum_r c[r] V[r]; then either the partial dots of the NEW w with every row (second pass) or its partial squared norm
__global__ __launch_bounds__(256) void gs_update_kernel(const double *__restrict__ V, int64_t n, int R, const double *__restrict__ c,
                                                       double *__restrict__ w, double *__restrict__ partial, int want_dots) {
    __shared__ double red[4];
    __shared__ double cs[256];
    for (int r = threadIdx.x; r < R; r += 256) cs[r] = c[r];
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * GS_CHUNK + threadIdx.x;
    double wv[GS_PER];
#pragma unroll
    for (int u = 0; u < GS_PER; ++u) wv[u] = i0 + u * 256 < n ? w[i0 + u * 256] : 0.0;
    for (int r = 0; r < R; ++r) {
        const double *row = V + (int64_t)r * n;
        const double cr = cs[r];
#pragma unroll
        for (int u = 0; u < GS_PER; ++u) if (i0 + u * 256 < n) wv[u] -= cr * row[i0 + u * 256];
    }
#pragma unroll
    for (int u = 0; u < GS_PER; ++u) if (i0 + u * 256 < n) w[i0 + u * 256] = wv[u];
    if (want_dots) {
        __shared__ double red2[4][GS_GROUP];
        gs_dots_chunk(V, n, R, wv, i0, partial + (int64_t)blockIdx.x * R, red2);
    } else {
        double sq = 0.0;
#pragma unroll
        for (int u = 0; u < GS_PER; ++u) sq += wv[u] * wv[u];
        const double tot = gs_block_sum(sq, red);
        if (threadIdx.x == 0) partial[blockIdx.x] = tot;
    }
}
// beta = sqrt(sum of the partial squared norms) (every workgroup adds them up in the same order), out = w / beta;
// workgroup 0 also files the step's results: beta, the coefficients h1 + h2 of the rows from `first` on -> tcol, max |h|.
__global__ __launch_bounds__(256) void gs_finish_kernel(const double *__restrict__ w, int64_t n, const double *__restrict__ partial, int nblocks,
                                                       double *__restrict__ out, const double *__restrict__ h1, const double *__restrict__ h2,
                                                       int R, int first, double *__restrict__ tcol, int64_t tstride, double *__restrict__ beta_out,
                                                       double *__restrict__ hmax_out) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) acc += partial[b];
    const double beta = sqrt(gs_block_sum(acc, red));
    const int64_t i0 = (int64_t)blockIdx.x * GS_CHUNK + threadIdx.x;
#pragma unroll
    for (int u = 0; u < GS_PER; ++u) if (i0 + u * 256 < n) out[i0 + u * 256] = w[i0 + u * 256] / beta;
    if (blockIdx.x == 0) {
        double hm = 0.0;
        for (int r = first + (int)threadIdx.x; r < R; r += 256) {
            const double hv = h1[r] + h2[r];
            tcol[(int64_t)(r - first) * tstride] = hv;
            hm = fmax(hm, fabs(hv));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) hm = fmax(hm, __shfl_xor(hm, off, 64));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = hm;
        __syncthreads();
        if (threadIdx.x == 0) { *beta_out = beta; *hmax_out = fmax(fmax(red[0], red[1]), fmax(red[2], red[3])); }
    }
}

}  // namespace

static thread_local std::string g_spectral_error;

extern "C" const char *gh_spectral_last_error(void) { return g_spectral_error.c_str(); }

extern "C" gh_status gh_spmv_symnorm(void *hip_stream, int64_t n, const int64_t *indptr, const int32_t *indices,
                                     const double *inv_sqrt_deg, const double *x, double *y) {
    if (n <= 0 || !indptr || !indices || !inv_sqrt_deg || !x || !y) {
        g_spectral_error = "bad argument";
        return GH_ERR_INVALID;
    }
    const int64_t threads = n * 8;
    spmv_symnorm_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream)>>>(
        n, indptr, indices, inv_sqrt_deg, x, y);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_spectral_error = std::string("kernel launch: ") + hipGetErrorString(e);
        return GH_ERR_HIP;
    }
    return GH_OK;
}

// Steps j = k .. m-1 of a (thick-restart) Lanczos sweep, all on the stream, no host synchronisation:
//   w = B V[nl + j]; classical Gram-Schmidt TWICE against rows 0 .. nl + j of V (the nl locked vectors come first and only
//   orthogonalise); Td[0..j][j] = the coefficients of rows nl.. (h1 + h2); beta[j] = |w|; hmax[j] = max |h|;
//   V[nl + j + 1] = w / beta[j].
// V: (nl + m + 1, n) row-major fp64; Td: (m, m) row-major; work: n + 2 * (nl + m + 1) + ceil(n / 512) * (nl + m + 1)
// doubles.  A tiny beta[j] (invariant subspace reached) leaves rows past nl + j + 1 meaningless: the caller reads beta /
// hmax after the sweep and cuts the basis there.
extern "C" gh_status gh_trlan_sweep(void *hip_stream, int64_t n, const int64_t *indptr, const int32_t *indices,
                                    const double *inv_sqrt_deg, double *V, int32_t nl, int32_t m, int32_t k, double *Td,
                                    double *work, double *beta, double *hmax) {
    if (n <= 0 || !indptr || !indices || !inv_sqrt_deg || !V || !Td || !work || !beta || !hmax || nl < 0 || m < 1 || k < 0 || k > m ||
        nl + m + 1 > 256) {
        g_spectral_error = "bad argument";
        return GH_ERR_INVALID;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    const int nb = (int)((n + GS_CHUNK - 1) / GS_CHUNK);
    const int Rmax = nl + m + 1;
    double *w = work, *h1 = w + n, *h2 = h1 + Rmax, *partial = h2 + Rmax;
    for (int j = k; j < m; ++j) {
        const int R = nl + j + 1;
        const gh_status s0 = gh_spmv_symnorm(hip_stream, n, indptr, indices, inv_sqrt_deg, V + (int64_t)(nl + j) * n, w);
        if (s0 != GH_OK) return s0;
        gs_dots_kernel<<<dim3(nb), dim3(256), 0, st>>>(V, n, R, w, partial);
        gs_reduce_kernel<<<dim3(R), dim3(256), 0, st>>>(partial, nb, R, h1, 0);
        gs_update_kernel<<<dim3(nb), dim3(256), 0, st>>>(V, n, R, h1, w, partial, 1);
        gs_reduce_kernel<<<dim3(R), dim3(256), 0, st>>>(partial, nb, R, h2, 0);
        gs_update_kernel<<<dim3(nb), dim3(256), 0, st>>>(V, n, R, h2, w, partial, 0);
        gs_finish_kernel<<<dim3(nb), dim3(256), 0, st>>>(w, n, partial, nb, V + (int64_t)(nl + j + 1) * n, h1, h2, R, nl, Td + j, m, beta + j, hmax + j);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_spectral_error = std::string("kernel launch: ") + hipGetErrorString(e);
        return GH_ERR_HIP;
    }
    return GH_OK;
}
