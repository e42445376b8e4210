// K-smallest extraction of a workgroup / wave, and the per-query intersection phase: shared by the KNN kernels (knn.hip)
// and the GH_DIST_CDIST kernels (cdist.hip).
#pragma once
#include "common.h"
#include "engine.h"
#include "intersect_core.h"
#include <type_traits>

namespace {

// Arguments of the intersection phase, for kernels that finish a query and process its k pairs
// in the same launch (single-rank runs).  pos == nullptr: do not intersect.
struct inter_args {
    const float *pos;
    const int32_t *edges;
    const int32_t *sampled;
    int D, LD, k;
    float k_inter;
    double *acc;
    int32_t *tflag, *touched, *tcount;
    float *scratch;
    int32_t *tq_count = nullptr;    // knn_select_wave_kernel: per-query runs of the touched list (S), (S, 4 k)
    int32_t *tq_touched = nullptr;
    int32_t own_lo = 0, own_hi = 0x7FFFFFFF;   // a row partition accumulates only what lands on its own rows
};

// Whole workgroup; best[] in LDS, visible to all threads (the callers' extraction ends with a barrier).
// DT: the number of components as a compile-time constant (2..16), or 0 = whatever ia.D says.  The kernels that end with
// this are instantiated per dimension: with all fifteen wide instantiations inlined behind a switch they were 28 000
// instructions each, and a workgroup that runs once per launch fetches its path through them from a cold instruction
// cache (see tau_core.h for what that cost the threshold kernel).
template <int DT>
__device__ __forceinline__ void intersect_query(const inter_args &ia, int64_t qi, const uint64_t *best, gh_pair_list *pl) {
    if constexpr (DT >= 2) {
        constexpr int LL = DT <= 4 ? 4 : DT <= 8 ? 8 : 16;
        if (ia.k <= 127 && blockDim.x % (4 * LL) == 0) {   // lanes = (role, coordinate)
            if (ia.tq_count)
                gh_intersect_query_wide<DT, LL>(ia.pos, ia.edges, ia.sampled[qi], best, ia.k, ia.k_inter, ia.acc, ia.tflag,
                                                ia.tq_touched + qi * 4 * ia.k, ia.tcount, pl, ia.tq_count + qi, ia.own_lo, ia.own_hi);
            else
                gh_intersect_query_wide<DT, LL>(ia.pos, ia.edges, ia.sampled[qi], best, ia.k, ia.k_inter, ia.acc, ia.tflag, ia.touched,
                                                ia.tcount, pl, nullptr, ia.own_lo, ia.own_hi);
            return;
        }
    }
    // neighbour c of the query is key column c+1: column 0 is dropped blindly (pt.py:421)
    for (int c = threadIdx.x; c < ia.k; c += blockDim.x) {
        const int32_t j = (int32_t)gh_key_id(best[c + 1]);
        if constexpr (DT >= 2)
            gh_intersect_pair_t<DT, (DT <= 4 ? 4 : DT <= 8 ? 8 : 16)>(ia.pos, ia.edges, ia.sampled[qi], j, ia.k_inter, ia.acc, ia.tflag,
                                                                     ia.touched, ia.tcount, ia.own_lo, ia.own_hi);
        else
            gh_intersect_pair(ia.pos, ia.D, ia.LD, ia.edges, ia.sampled[qi], j, ia.k_inter, ia.acc, ia.tflag, ia.touched, ia.tcount,
                              ia.scratch + (qi * ia.k + c) * ia.LD, ia.own_lo, ia.own_hi);
    }
}
// host side: run X<DT> for the engine's dimension (0 past 16)
#define GH_DISPATCH_DIM(Dval, X)                                                      \
    switch (Dval) {                                                                   \
        case 2: X(2); break;  case 3: X(3); break;  case 4: X(4); break;  case 5: X(5); break;    \
        case 6: X(6); break;  case 7: X(7); break;  case 8: X(8); break;  case 9: X(9); break;    \
        case 10: X(10); break; case 11: X(11); break; case 12: X(12); break; case 13: X(13); break; \
        case 14: X(14); break; case 15: X(15); break; case 16: X(16); break;                        \
        default: X(0); break;                                                         \
    }

// Bitonic sort of n2 (power of two) keys in LDS by one 256-thread workgroup, ascending.
// Only used when K > GH_EXTRACT_MAX_K (latency-bound: ~20 us per 1024 keys).
__device__ __forceinline__ void block_sort(uint64_t *buf, int n2) {
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

// Wave-wide minimum of 64-bit keys on the DPP datapath (quad swaps, half-row and row mirrors:
// VALU-rate, no LDS crossbar round trips as with __shfl_xor), then the four row minima through
// v_readlane.  Every lane returns the minimum.
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v) {
    int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t min_u64(uint64_t a, uint64_t b) { return b < a ? b : a; }
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
    v = min_u64(v, dpp_u64<0xB1>(v));   // quad_perm [1,0,3,2]
    v = min_u64(v, dpp_u64<0x4E>(v));   // quad_perm [2,3,0,1]
    v = min_u64(v, dpp_u64<0x141>(v));  // row_half_mirror
    v = min_u64(v, dpp_u64<0x140>(v));  // row_mirror: every lane holds its row's minimum
    return min_u64(min_u64(readlane_u64(v, 0), readlane_u64(v, 16)),
                   min_u64(readlane_u64(v, 32), readlane_u64(v, 48)));
}

#define GH_EXTRACT_MAX_K 128

// K smallest of the keys an NT-thread workgroup holds in registers (NPT per thread, unused
// slots = GH_KEY_INF), written ascending to out[0..K) in LDS.  Each of the NT/64 waves extracts
// the K smallest of ITS keys on its own -- K rounds of a wave-wide minimum, no barrier; keys are
// unique (the id is part of the key), so the owner of a round's minimum retires it by equality
// -- and the survivors are ranked by counting.  wsc: (NT/64) * GH_EXTRACT_MAX_K keys of LDS scratch.
template <int NPT, int NT = 256>
__device__ __forceinline__ void block_extract_smallest(uint64_t (&keys)[NPT], int K, uint64_t *out, uint64_t *wsc) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = 0; r < K; ++r) {
        uint64_t m = keys[0];
#pragma unroll
        for (int j = 1; j < NPT; ++j) m = min_u64(m, keys[j]);
        m = wave_min_u64(m);
        if (lane == 0) wsc[w * K + r] = m;
        if (m == GH_KEY_INF) {  // wave-uniform: this wave has run dry
            for (int rr = r + 1 + lane; rr < K; rr += 64) wsc[w * K + rr] = GH_KEY_INF;
            break;
        }
#pragma unroll
        for (int j = 0; j < NPT; ++j) keys[j] = keys[j] == m ? GH_KEY_INF : keys[j];
    }
    __syncthreads();
    const int n4 = (NT / 64) * K;
    for (int t = threadIdx.x; t < n4; t += NT) {
        const uint64_t key = wsc[t];
        int rank = 0;
        for (int j = 0; j < n4; ++j) {
            const uint64_t o = wsc[j];
            rank += (o < key || (o == key && j < t)) ? 1 : 0;  // equal keys are GH_KEY_INF fillers only
        }
        if (rank < K) out[rank] = key;
    }
    __syncthreads();
}



// K smallest of the c keys in src (LDS or global), by the smallest per-thread register count
// that holds them: the cost of a round is proportional to the keys each thread rescans.
// K smallest of c <= 2 * NT keys by counting: every key's rank among all of them (keys are unique), one pass over an
// LDS copy.  The wave-minimum rounds above cost ~100 instructions per extracted key and run K times in a row: at
// K = 33 (n_neighbors = 32) that was 8 us of a select launch; counting costs c/2 LDS reads whatever K is.
template <int NT = 256>
__device__ __forceinline__ void block_rank_smallest(const uint64_t *src, int c, int K, uint64_t *out, uint64_t *stage /* >= 2 * NT keys */) {
    uint64_t mine[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = j * NT + threadIdx.x;
        mine[j] = i < c ? src[i] : GH_KEY_INF;
        stage[i] = mine[j];
    }
    for (int i = c + threadIdx.x; i < K; i += NT) out[i] = GH_KEY_INF;   // fewer than K keys: the tail
    __syncthreads();
    int rank[2] = {0, 0};
    const int c2 = (c + 1) & ~1;
    for (int i = 0; i < c2; i += 2) {
        const uint64_t a = stage[i], b = stage[i + 1];   // broadcast read of 16 bytes (slots past c hold GH_KEY_INF)
#pragma unroll
        for (int j = 0; j < 2; ++j) rank[j] += (a < mine[j] ? 1 : 0) + (b < mine[j] ? 1 : 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (mine[j] != GH_KEY_INF && rank[j] < K) out[rank[j]] = mine[j];
    __syncthreads();
}

// (forced inline, like the others here: once the kernels were instantiated per dimension the compiler stopped inlining
// this one, and the call -- 246 VGPRs for the callee's worst case -- cost the k = 32 selection 4.6 us)
template <int MAXNPT, int NT = 256>
__device__ __forceinline__ void block_extract_adaptive(const uint64_t *src, int c, int K, uint64_t *out, uint64_t *red) {
    if (K > 16 && c <= 2 * NT && 2 * NT <= (NT / 64) * GH_EXTRACT_MAX_K) {   // red holds (NT/64) * GH_EXTRACT_MAX_K keys
        block_rank_smallest<NT>(src, c, K, out, red);
        return;
    }
    auto run = [&](auto npt_tag) {
        constexpr int NPT = decltype(npt_tag)::value;
        uint64_t keys[NPT];
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int i = j * NT + threadIdx.x;
            keys[j] = i < c ? src[i] : GH_KEY_INF;
        }
        __syncthreads();  // src may alias out's neighbourhood in LDS; everyone has loaded
        block_extract_smallest<NPT, NT>(keys, K, out, red);
    };
    if (c <= NT || MAXNPT <= 1) run(std::integral_constant<int, 1>{});
    else if (c <= 4 * NT || MAXNPT <= 4) run(std::integral_constant<int, (MAXNPT < 4 ? MAXNPT : 4)>{});
    else if (c <= 8 * NT || MAXNPT <= 8) run(std::integral_constant<int, (MAXNPT < 8 ? MAXNPT : 8)>{});
    else if (c <= 16 * NT || MAXNPT <= 16) run(std::integral_constant<int, (MAXNPT < 16 ? MAXNPT : 16)>{});
    else run(std::integral_constant<int, MAXNPT>{});
}

// K smallest of a candidate list of up to 2 * GH_LIST_HALF keys in memory: the register extraction takes GH_LIST_HALF keys
// (32 per thread); a longer list is taken in two halves -- ONE call site in a loop, so the common path's code does not
// grow -- and the two results are merged by rank counting.  (Raising the per-thread key count instead made the select
// kernel use scratch: +2.7 us per launch at 1 M vertices.)  best, best2: K keys of LDS each; red: 4 * GH_EXTRACT_MAX_K.
#define GH_LIST_HALF 8192
template <int NT = 256>
__device__ __forceinline__ void block_extract_list(const uint64_t *list, int c, int K, uint64_t *best, uint64_t *best2, uint64_t *red) {
    const int nh = c > GH_LIST_HALF ? 2 : 1;
    for (int hh = 0; hh < nh; ++hh)
        block_extract_adaptive<GH_LIST_HALF / NT, NT>(list + hh * GH_LIST_HALF, min(GH_LIST_HALF, c - hh * GH_LIST_HALF), K, hh == 0 ? best : best2, red);
    if (nh == 2) {
        for (int i = threadIdx.x; i < 2 * K; i += NT) red[i] = i < K ? best[i] : best2[i - K];
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * K; i += NT) {
            const uint64_t key = red[i];
            int rank = 0;
            for (int j = 0; j < 2 * K; ++j) rank += (red[j] < key || (red[j] == key && j < i)) ? 1 : 0;   // equal keys: GH_KEY_INF fillers only
            if (rank < K) best[rank] = key;
        }
        __syncthreads();
    }
}

// Column `col` of the fused kernel's per-workgroup sums ([entry][workgroup]) -> stats[col], in a fixed order; one
// 256-thread workgroup, dred: 4 doubles of LDS.  (Rides in the selection launches of single-rank fused steps: it depends on
// the fused kernel only, so it runs beside the selection instead of in front of stats_fix_kernel's corrections.)
__device__ __forceinline__ void gh_reduce_stats_column(const double *__restrict__ blockstats, int nblocks, int col,
                                                       double *__restrict__ stats, double *dred) {
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    const double *c = blockstats + (int64_t)col * nblocks;
    int b = threadIdx.x;
    for (; b + 3 * 256 < nblocks; b += 4 * 256) {
#pragma unroll
        for (int u = 0; u < 4; ++u) s4[u] += c[b + u * 256];
    }
    for (int u = 0; b < nblocks; b += 256, ++u) s4[u] += c[b];
    const double a = gh_wave_sum((s4[0] + s4[1]) + (s4[2] + s4[3]));
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) stats[col] = ((dred[0] + dred[1]) + dred[2]) + dred[3];
}

// Arguments of the intersection phase of a launch that also finishes queries (on = false: none).
inline inter_args make_inter_args(gh_engine *h, bool on) {
    inter_args ia{};
    if (on) {
        ia = inter_args{h->d_pos, h->d_edges, h->d_sampled_cur, h->D, h->LD, h->k, h->prm.k_inter,
                        h->d_acc, h->d_tflag, h->d_touched, h->d_tcount, h->d_iscratch};
        if (h->rows != h->n) { ia.own_lo = (int32_t)h->part.row_lo; ia.own_hi = (int32_t)h->part.row_hi; }
    }
    return ia;
}

}  // namespace
