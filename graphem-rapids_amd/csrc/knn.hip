// KNN of sampled edge midpoints among all edge midpoints: the work of
// _locate_knn_midpoints / _compute_knn_chunked / _compute_knn_torch
// (reference pt.py:381-424, 426-483, 543-593) without the (S, E) distance matrix and
// without materialised midpoints.
//
// Exact brute force.  Distances are squared, in exact-difference form, as an fma chain
// in coordinate order (bit-identical to the oracle's go_d2); ties break on the smaller
// edge id.  Structure (DESIGN.md "KNN"):
//   level 0   one workgroup per query selects the K-th smallest distance over a strided
//             subset of ~2048 reference edges -> an upper bound tau on the true K-th
//             distance (the K-th order statistic of a subset can only be larger);
//   level l   every workgroup takes a tile of reference edges (midpoints recomputed from
//             edges + positions, kept in registers), loops over ALL queries (query
//             coordinates and tau arrive as wave-uniform scalar loads) and appends the few
//             references with dist2 <= tau to that query's candidate list; a per-query
//             workgroup then sorts the list in LDS and either tightens tau (nested
//             subset, next level) or emits the K best keys (last level = all edges);
//   fallback  a query whose list overflowed is redone by the level-0 kernel over all
//             edges, which is exact for any input.
#include "common.h"
#include "engine.h"

#include <math.h>

namespace {

// ---------------------------------------------------------------------------------
// Query midpoints (pt.py:785 for the sampled rows, pt.py:410) and list reset.
__global__ void knn_prepare_kernel(const float *__restrict__ pos, const int32_t *__restrict__ edges,
                                   const int32_t *__restrict__ sampled, int64_t S, int LD,
                                   float *__restrict__ q, int32_t *__restrict__ cnt, int32_t *__restrict__ ovf) {
    const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int64_t e = sampled[s];
    const int64_t u = edges[2 * e], v = edges[2 * e + 1];
    for (int d = 0; d < LD; ++d) q[s * LD + d] = (pos[u * LD + d] + pos[v * LD + d]) / 2.0f;
    cnt[s] = 0;
    ovf[s] = 0;
}

// Bitonic sort of n2 (power of two) keys in LDS by one 256-thread workgroup, ascending.
__device__ void block_sort(uint64_t *buf, int n2) {
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = buf[i], b = buf[ixj];
                    const bool asc = (i & k) == 0;
                    if ((a > b) == asc) { buf[i] = b; buf[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ int next_pow2(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

// ---------------------------------------------------------------------------------
// One workgroup per query: exact K smallest (dist2, id) keys over the reference edges
// e_lo + j*stride, j < M.  Any D (runtime), any K <= GH_SEL_BUF - GH_SEL_CHUNK.
// Running threshold + LDS compaction: keys below the current K-th key are appended to an
// LDS buffer; when the next chunk might not fit, the buffer is sorted and cut to K.
__global__ __launch_bounds__(256) void knn_block_select_kernel(
    const float *__restrict__ pos, int LD, int D, const int32_t *__restrict__ edges, int64_t e_lo, int64_t M,
    int64_t stride, const float *__restrict__ q, int K, const int32_t *__restrict__ only_flagged,
    uint64_t *__restrict__ out_keys /* (S, K) or null */, float *__restrict__ tau_out /* (S) or null */) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint64_t *buf = reinterpret_cast<uint64_t *>(smem_raw);                       // GH_SEL_BUF keys
    float *qs = reinterpret_cast<float *>(smem_raw + sizeof(uint64_t) * GH_SEL_BUF);  // LD floats
    __shared__ int cnt;
    __shared__ uint64_t tau_key;

    const int64_t qi = blockIdx.x;
    if (only_flagged && only_flagged[qi] == 0) return;
    for (int d = threadIdx.x; d < LD; d += blockDim.x) qs[d] = q[qi * LD + d];
    if (threadIdx.x == 0) { cnt = 0; tau_key = GH_KEY_INF; }
    __syncthreads();

    for (int64_t base = 0; base < M; base += GH_SEL_CHUNK) {
        const uint64_t tk = tau_key;
        for (int j = threadIdx.x; j < GH_SEL_CHUNK; j += blockDim.x) {
            const int64_t r = base + j;
            if (r < M) {
                const int64_t e = e_lo + r * stride;
                const int64_t u = edges[2 * e], v = edges[2 * e + 1];
                const float *pu = pos + u * LD, *pv = pos + v * LD;
                float s = 0.0f;
                for (int d = 0; d < D; ++d) {
                    const float m = (pu[d] + pv[d]) / 2.0f;
                    const float t = qs[d] - m;
                    s = fmaf(t, t, s);
                }
                const uint64_t key = gh_key(s, (uint32_t)e);
                if (key < tk) {
                    const int p = atomicAdd(&cnt, 1);
                    buf[p] = key;  // p < GH_SEL_BUF: at most K + GH_SEL_CHUNK entries before a cut
                }
            }
        }
        __syncthreads();
        const int c = cnt;
        const bool last = base + GH_SEL_CHUNK >= M;
        if (c > GH_SEL_BUF - GH_SEL_CHUNK || last) {
            const int n2 = next_pow2(c < 2 ? 2 : c);
            for (int i = c + threadIdx.x; i < n2; i += blockDim.x) buf[i] = GH_KEY_INF;
            __syncthreads();
            block_sort(buf, n2);
            if (threadIdx.x == 0) {
                cnt = c < K ? c : K;
                tau_key = c >= K ? buf[K - 1] : GH_KEY_INF;
            }
            __syncthreads();
        }
    }
    const int c = cnt;
    if (out_keys)
        for (int i = threadIdx.x; i < K; i += blockDim.x) out_keys[qi * K + i] = i < c ? buf[i] : GH_KEY_INF;
    if (tau_out && threadIdx.x == 0) tau_out[qi] = c >= K ? gh_key_d2(buf[K - 1]) : INFINITY;
}

// ---------------------------------------------------------------------------------
// The filtered scan.  256 threads x R reference midpoints in registers; all S queries
// stream past as scalar operands.  Per pair: D sub, 1 mul, D-1 fma, 1 compare.
template <int D, int R>
__global__ __launch_bounds__(256) void knn_scan_kernel(
    const float *__restrict__ pos, const int32_t *__restrict__ edges, int64_t e_lo, int64_t M, int64_t stride,
    const float *__restrict__ q, const float *__restrict__ tau, int S, uint64_t *__restrict__ cand,
    int32_t *__restrict__ cnt) {
    constexpr int LD = D <= 4 ? 4 : D <= 8 ? 8 : 16;
    float m[R][D];
    uint32_t id[R];
    const int64_t tile = (int64_t)blockIdx.x * (256 * R);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t j = tile + r * 256 + threadIdx.x;
        if (j < M) {
            const int64_t e = e_lo + j * stride;
            const int2 uv = reinterpret_cast<const int2 *>(edges)[e];
            float pu[LD], pv[LD];
            gh_load_row<LD>(pos, uv.x, pu);
            gh_load_row<LD>(pos, uv.y, pv);
#pragma unroll
            for (int d = 0; d < D; ++d) m[r][d] = (pu[d] + pv[d]) / 2.0f;
            id[r] = (uint32_t)e;
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) m[r][d] = INFINITY;  // dist2 = inf never passes dist2 <= tau
            id[r] = 0xFFFFFFFFu;
        }
    }
    for (int s = 0; s < S; ++s) {
        float qv[D];
#pragma unroll
        for (int d = 0; d < D; ++d) qv[d] = q[s * LD + d];  // wave-uniform -> scalar loads
        const float t = tau[s];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float t0 = qv[0] - m[r][0];
            float d2 = t0 * t0;  // == fmaf(t0, t0, +0)
#pragma unroll
            for (int d = 1; d < D; ++d) {
                const float td = qv[d] - m[r][d];
                d2 = fmaf(td, td, d2);
            }
            if (d2 <= t) {
                const int p = atomicAdd(&cnt[s], 1);
                if (p < GH_CAND_CAP) cand[(int64_t)s * GH_CAND_CAP + p] = gh_key(d2, id[r]);
            }
        }
    }
}

// One workgroup per query: sort the candidate list; final -> K best keys, else tighten tau.
__global__ __launch_bounds__(256) void knn_select_kernel(uint64_t *__restrict__ cand, int32_t *__restrict__ cnt,
                                                         int K, int final_level, float *__restrict__ tau,
                                                         uint64_t *__restrict__ out_keys,
                                                         int32_t *__restrict__ ovf) {
    __shared__ uint64_t buf[GH_CAND_CAP];
    const int64_t qi = blockIdx.x;
    const int c = cnt[qi];
    __syncthreads();
    if (threadIdx.x == 0) cnt[qi] = 0;
    if (c > GH_CAND_CAP || c < K) {
        // overflow (or an impossible short list): the list is not trustworthy
        if (final_level && threadIdx.x == 0) ovf[qi] = 1;
        return;  // tau keeps its previous (still valid, looser) value
    }
    const int n2 = next_pow2(c < 2 ? 2 : c);
    for (int i = threadIdx.x; i < n2; i += blockDim.x) buf[i] = i < c ? cand[qi * GH_CAND_CAP + i] : GH_KEY_INF;
    __syncthreads();
    block_sort(buf, n2);
    if (final_level) {
        for (int i = threadIdx.x; i < K; i += blockDim.x) out_keys[qi * K + i] = buf[i];
    } else if (threadIdx.x == 0) {
        tau[qi] = gh_key_d2(buf[K - 1]);
    }
}

// Merge the per-rank key lists (world, S, K) -> neighbour ids with column 0 dropped
// (pt.py:421: knn_indices[:, 1:]).  world == 1 degenerates to a copy.
__global__ __launch_bounds__(256) void knn_merge_kernel(const uint64_t *__restrict__ gathered, int world, int64_t S,
                                                        int K, int32_t *__restrict__ knn) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint64_t *buf = reinterpret_cast<uint64_t *>(smem_raw);
    const int64_t qi = blockIdx.x;
    const int total = world * K;
    const int n2 = next_pow2(total < 2 ? 2 : total);
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        uint64_t key = GH_KEY_INF;
        if (i < total) {
            const int w = i / K, c = i % K;
            key = gathered[((int64_t)w * S + qi) * K + c];
        }
        buf[i] = key;
    }
    __syncthreads();
    if (world > 1) block_sort(buf, n2);
    for (int c = 1 + threadIdx.x; c < K; c += blockDim.x) knn[qi * (K - 1) + (c - 1)] = (int32_t)gh_key_id(buf[c]);
}

template <int D, int R>
void launch_scan(gh_engine *h, int64_t M, int64_t stride) {
    const int64_t per = 256 * R;
    const int64_t grid = (M + per - 1) / per;
    knn_scan_kernel<D, R><<<dim3((unsigned)grid), dim3(256), 0, h->stream>>>(
        h->d_pos, h->d_edges, h->part.edge_lo, M, stride, h->d_q, h->d_tau, (int)h->S, h->d_cand, h->d_cnt);
}

}  // namespace

gh_status gh_knn_local(gh_engine *h) {
    const int64_t Mtot = h->part.edge_hi - h->part.edge_lo;
    const int K = h->K;
    const size_t sel_smem = sizeof(uint64_t) * GH_SEL_BUF + sizeof(float) * (size_t)h->LD;
    {
        gh_scope t(h, "knn_prepare");
        const int bs = 256;
        knn_prepare_kernel<<<dim3((unsigned)((h->S + bs - 1) / bs)), dim3(bs), 0, h->stream>>>(
            h->d_pos, h->d_edges, h->d_sampled_cur, h->S, h->LD, h->d_q, h->d_cnt, h->d_ovf);
        GH_LAUNCH_CHECK();
    }
    const bool scan_path = Mtot >= GH_SCAN_MIN_EDGES && h->LD <= 16 && h->D >= 2 && K <= 128 && h->S <= 0x7FFFFFFF;
    if (!scan_path) {
        gh_scope t(h, "knn_block_select");
        knn_block_select_kernel<<<dim3((unsigned)h->S), dim3(256), sel_smem, h->stream>>>(
            h->d_pos, h->LD, h->D, h->d_edges, h->part.edge_lo, Mtot, 1, h->d_q, K, nullptr, h->d_partial, nullptr);
        GH_LAUNCH_CHECK();
        return GH_OK;
    }
    // Level plan: nested strided subsets, ratio r between levels, ~2048 edges at level 0.
    int rmax = 1536 / K;
    if (rmax > 64) rmax = 64;
    if (rmax < 2) rmax = 2;
    const double want = (double)Mtot / 2048.0;
    int L = 1;
    while (pow((double)rmax, L) < want) ++L;
    int64_t r = (int64_t)ceil(pow(want, 1.0 / L));
    if (r < 2) r = 2;
    std::vector<int64_t> strides(L + 1);
    strides[L] = 1;
    for (int l = L - 1; l >= 0; --l) strides[l] = strides[l + 1] * r;
    {
        gh_scope t(h, "knn_level0_select");
        const int64_t M0 = (Mtot + strides[0] - 1) / strides[0];
        knn_block_select_kernel<<<dim3((unsigned)h->S), dim3(256), sel_smem, h->stream>>>(
            h->d_pos, h->LD, h->D, h->d_edges, h->part.edge_lo, M0, strides[0], h->d_q, K, nullptr, nullptr, h->d_tau);
        GH_LAUNCH_CHECK();
    }
    for (int l = 1; l <= L; ++l) {
        const int64_t M = (Mtot + strides[l] - 1) / strides[l];
        {
            gh_scope t(h, l == L ? "knn_scan" : "knn_scan_subset");
            switch (h->D) {
                case 2: launch_scan<2, 4>(h, M, strides[l]); break;
                case 3: launch_scan<3, 4>(h, M, strides[l]); break;
                case 4: launch_scan<4, 4>(h, M, strides[l]); break;
                default:
                    if (h->LD == 8) launch_scan<8, 2>(h, M, strides[l]);
                    else launch_scan<16, 2>(h, M, strides[l]);
            }
            GH_LAUNCH_CHECK();
        }
        {
            gh_scope t(h, "knn_select");
            knn_select_kernel<<<dim3((unsigned)h->S), dim3(256), 0, h->stream>>>(
                h->d_cand, h->d_cnt, K, l == L ? 1 : 0, h->d_tau, h->d_partial, h->d_ovf);
            GH_LAUNCH_CHECK();
        }
    }
    {
        gh_scope t(h, "knn_overflow_fallback");
        knn_block_select_kernel<<<dim3((unsigned)h->S), dim3(256), sel_smem, h->stream>>>(
            h->d_pos, h->LD, h->D, h->d_edges, h->part.edge_lo, Mtot, 1, h->d_q, K, h->d_ovf, h->d_partial, nullptr);
        GH_LAUNCH_CHECK();
    }
    return GH_OK;
}

gh_status gh_knn_merge(gh_engine *h, const uint64_t *gathered, int world) {
    const int total = world * h->K;
    int n2 = 2;
    while (n2 < total) n2 <<= 1;
    if ((size_t)n2 * sizeof(uint64_t) > 60 * 1024) {
        h->err = "world * (n_neighbors + 1) too large for the merge kernel";
        return GH_ERR_INVALID;
    }
    gh_scope t(h, "knn_merge");
    knn_merge_kernel<<<dim3((unsigned)h->S), dim3(256), sizeof(uint64_t) * (size_t)n2, h->stream>>>(
        gathered, world, h->S, h->K, h->d_knn);
    GH_LAUNCH_CHECK();
    return GH_OK;
}
