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
#include "scan_core.h"
#include "setup_core.h"
#include "tau_core.h"
#include "intersect_core.h"
#include "select_core.h"

#include <math.h>
#include <type_traits>
#include <stdlib.h>

namespace {


// ---------------------------------------------------------------------------------
// One launch sets a KNN search up from the CURRENT positions (setup_core.h): sample ids, query
// records, list reset and -- on the scan path, a.tiles > 0 -- the group minima of the threshold
// subset.  (In a run the same work rides in the previous iteration's normalise launch instead.)
__global__ __launch_bounds__(256) void knn_setup_kernel(const float *__restrict__ pos, gh_setup_args a,
                                                       int32_t *__restrict__ tcount, int32_t *__restrict__ qexact) {
    __shared__ __align__(16) unsigned char lds[GH_SETUP_LDS_BYTES];
    if (blockIdx.x == 0 && threadIdx.x == 0) { *tcount = 0; qexact[0] = 0; }  // the previous iteration's normalise kernel has consumed it
    const int LD = a.LD;
    auto getp = [=](int64_t v, int d) { return pos[v * LD + d]; };
    if (a.tiles > 0) gh_setup_block_any(a, (int)blockIdx.x, getp, lds);
    else gh_setup_item(a, blockIdx.x * (int64_t)blockDim.x + threadIdx.x, getp);
}


// ---------------------------------------------------------------------------------
// One workgroup per query: exact K smallest (dist2, id) keys over the reference edges
// e_lo + j*stride, j < M.  Any D (runtime).  Chunks of 2048 references: every thread
// computes 8 keys into registers; keys that do not beat the current K-th key are dropped
// at once; a chunk with a survivor re-extracts the best K from (survivors + previous best).
// K <= GH_EXTRACT_MAX_K.
struct search_args {  // what the exact per-query search reads
    const float *mid;      // midpoint rows (row j <-> reference j), or null: gather the endpoints from pos
    const float *pos;
    const int32_t *edges;
    const int32_t *eids;   // ids of the references, or null: e_lo + j*stride
    int LD, D;
    int64_t e_lo, M, mem_stride, stride;
    const float *qt;       // query records (QS floats each, coordinates first)
    int QS;
};

// best[0..K) <- the K smallest keys of query qi; qs: LD floats of LDS, red: extraction scratch.
__device__ __forceinline__ void block_search_query(const search_args &a, int64_t qi, int K, float *qs, uint64_t *best, uint64_t *red) {
    const float *__restrict__ mid = a.mid, *__restrict__ pos = a.pos, *__restrict__ qt = a.qt;
    const int32_t *__restrict__ edges = a.edges, *__restrict__ eids = a.eids;
    const int LD = a.LD, D = a.D, QS = a.QS;
    const int64_t e_lo = a.e_lo, M = a.M, mem_stride = a.mem_stride, stride = a.stride;
    constexpr int NPT = 8;
    for (int d = threadIdx.x; d < LD; d += blockDim.x) qs[d] = d < D ? qt[qi * QS + d] : 0.0f;
    for (int i = threadIdx.x; i < K; i += blockDim.x) best[i] = GH_KEY_INF;
    __syncthreads();

    for (int64_t base = 0; base < M; base += 256 * NPT) {
        const uint64_t tk = best[K - 1];
        uint64_t keys[NPT + 1];
        int any = 0;
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int64_t r = base + j * 256 + threadIdx.x;
            uint64_t key = GH_KEY_INF;
            if (r < M) {
                const int64_t e = eids ? (int64_t)eids[r * stride] : e_lo + r * stride;
                float s = 0.0f;
                if (mid) {
                    const float *mr = mid + (r * mem_stride) * LD;
                    for (int d = 0; d < D; ++d) {
                        const float t = qs[d] - mr[d];
                        s = fmaf(t, t, s);
                    }
                } else {  // no midpoint array (fused spring+scan path): gather the endpoints
                    const float *pu = pos + (int64_t)edges[2 * e] * LD, *pv = pos + (int64_t)edges[2 * e + 1] * LD;
                    for (int d = 0; d < D; ++d) {
                        const float t = qs[d] - (pu[d] + pv[d]) / 2.0f;
                        s = fmaf(t, t, s);
                    }
                }
                key = gh_key(s, (uint32_t)e);
                if (key < tk) any = 1; else key = GH_KEY_INF;
            }
            keys[j] = key;
        }
        if (__syncthreads_or(any)) {
            keys[NPT] = threadIdx.x < K ? best[threadIdx.x] : GH_KEY_INF;  // previous best joins the pool
            __syncthreads();
            block_extract_smallest<NPT + 1>(keys, K, best, red);
        }
    }
}

template <int DT>
__global__ __launch_bounds__(256) void knn_block_select_kernel(search_args a, int K,
                                                               const int32_t *__restrict__ only_flagged,
                                                               uint64_t *__restrict__ out_keys /* (S, K) or null */,
                                                               float *__restrict__ tau_out /* qt + tau offset, or null */,
                                                               inter_args ia) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float *qs = reinterpret_cast<float *>(smem_raw);  // LD floats
    __shared__ uint64_t best[GH_EXTRACT_MAX_K];
    __shared__ uint64_t red[4 * GH_EXTRACT_MAX_K];
    const int64_t qi = blockIdx.x;
    if (only_flagged && only_flagged[qi] == 0) return;
    block_search_query(a, qi, K, qs, best, red);
    if (out_keys)
        for (int i = threadIdx.x; i < K; i += blockDim.x) out_keys[qi * K + i] = best[i];
    if (tau_out && threadIdx.x == 0) tau_out[qi * a.QS] = best[K - 1] != GH_KEY_INF ? gh_key_d2(best[K - 1]) : INFINITY;
    __shared__ gh_pair_list pairs;
    if (ia.pos) intersect_query<DT>(ia, qi, best, &pairs);
}

// The same selection for large K (> GH_EXTRACT_MAX_K): running threshold + LDS compaction +
// bitonic sort.  K <= GH_SEL_BUF - GH_SEL_CHUNK.
__global__ __launch_bounds__(256) void knn_block_select_sort_kernel(
    const float *__restrict__ mid, const float *__restrict__ pos, const int32_t *__restrict__ edges,
    const int32_t *__restrict__ eids /* ids of the scanned edges, or null: e_lo + j*stride */, int LD, int D,
    int64_t e_lo, int64_t M, int64_t mem_stride, int64_t stride, const float *__restrict__ qt, int QS, int K, const int32_t *__restrict__ only_flagged,
    uint64_t *__restrict__ out_keys, float *__restrict__ tau_out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint64_t *buf = reinterpret_cast<uint64_t *>(smem_raw);                           // GH_SEL_BUF keys
    float *qs = reinterpret_cast<float *>(smem_raw + sizeof(uint64_t) * GH_SEL_BUF);  // LD floats
    __shared__ int cnt;
    __shared__ uint64_t tau_key;

    const int64_t qi = blockIdx.x;
    if (only_flagged && only_flagged[qi] == 0) return;
    for (int d = threadIdx.x; d < LD; d += blockDim.x) qs[d] = d < D ? qt[qi * QS + d] : 0.0f;
    if (threadIdx.x == 0) { cnt = 0; tau_key = GH_KEY_INF; }
    __syncthreads();

    for (int64_t base = 0; base < M; base += GH_SEL_CHUNK) {
        const uint64_t tk = tau_key;
        for (int j = threadIdx.x; j < GH_SEL_CHUNK; j += blockDim.x) {
            const int64_t r = base + j;
            if (r < M) {
                const int64_t e = eids ? (int64_t)eids[r * stride] : e_lo + r * stride;
                float s = 0.0f;
                if (mid) {
                    const float *mr = mid + (r * mem_stride) * LD;
                    for (int d = 0; d < D; ++d) {
                        const float t = qs[d] - mr[d];
                        s = fmaf(t, t, s);
                    }
                } else {  // no midpoint array (fused spring+scan path): gather the endpoints
                    const float *pu = pos + (int64_t)edges[2 * e] * LD, *pv = pos + (int64_t)edges[2 * e + 1] * LD;
                    for (int d = 0; d < D; ++d) {
                        const float t = qs[d] - (pu[d] + pv[d]) / 2.0f;
                        s = fmaf(t, t, s);
                    }
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
    if (tau_out && threadIdx.x == 0) tau_out[qi * QS] = c >= K ? gh_key_d2(buf[K - 1]) : INFINITY;
}

// ---------------------------------------------------------------------------------
// The filtered scan as a stand-alone kernel: references are read from the midpoint array
// (coalesced when stride == 1).  blockIdx.y selects the query group.  The inner loop lives in
// scan_core.h.
#define GH_SCAN_HITBUF 1024

template <int D, int R>
__global__ __launch_bounds__(256) void knn_scan_kernel(
    const float *__restrict__ mid, int64_t e_lo, const int32_t *__restrict__ eids, int64_t M, int64_t mem_stride, int64_t stride,
    const float *__restrict__ qt, const float *__restrict__ qscan, int S, int qgroup,
    uint64_t *__restrict__ cand, int32_t *__restrict__ cnt, int cdist /* GH_DIST_CDIST: keys carry ATen's cdist value */) {
    static_assert(R % 2 == 0, "references are processed in packed pairs");
    constexpr int LD = D <= 4 ? 4 : D <= 8 ? 8 : 16;
    constexpr int QS = D <= 3 ? 4 : LD + 4;
    __shared__ float4 qsh[(GH_SCAN_QGROUP + 1) * (QS / 4)];
    __shared__ float taush[GH_SCAN_QGROUP];
    __shared__ uint64_t hkey[GH_SCAN_HITBUF];
    __shared__ int hq[GH_SCAN_HITBUF];
    __shared__ int hcount;

    if (threadIdx.x == 0) hcount = 0;
    const int s_lo = blockIdx.y * qgroup;
    const int nq = min(S - s_lo, qgroup);
    gh_stage_queries<QS, (D <= 3 ? 3 : LD)>(qscan, qt, s_lo, nq, qsh, taush);
    gh_f2 m[R / 2][D], c0[R / 2];
    uint32_t id[R];
    const int64_t tile = (int64_t)blockIdx.x * (256 * R);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t j = tile + r * 256 + threadIdx.x;
        float mv[LD];
        if (j < M) {
            gh_load_row<LD>(mid, j * mem_stride, mv);
            id[r] = eids ? (uint32_t)eids[j * stride] : (uint32_t)(e_lo + j * stride);
        } else {
#pragma unroll
            for (int d = 0; d < LD; ++d) mv[d] = 0.0f;  // padding slot: c0 = +inf never passes the filter
            id[r] = 0xFFFFFFFFu;
        }
        const float c = gh_ref_c0<D>(mv, j < M);
        if (r & 1) c0[r / 2].y = c; else c0[r / 2].x = c;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (r & 1) m[r / 2][d].y = mv[d];
            else m[r / 2][d].x = mv[d];
        }
    }
    __syncthreads();
    gh_scan_queries<D, R, GH_SCAN_HITBUF>(m, c0, id, qsh, nq, s_lo, taush, hkey, hq, &hcount, cand, cnt, nullptr, cdist);
    __syncthreads();
    gh_flush_hits<GH_SCAN_HITBUF>(hkey, hq, &hcount, cand, cnt);
}

// Thresholds as a launch of their own (tau_core.h): one wave per query; one small kernel per register count and row form
// (the code runs once per CU from a cold instruction cache: what is not executed must not be in the way).
template <int NV, int FORM>
__global__ __launch_bounds__(64) void knn_tau_kernel(gh_tau_args a) { gh_tau_query_any<FORM, NV>(a, blockIdx.x, threadIdx.x); }

// One workgroup per query: K smallest of the candidate list; final -> K best keys (and the
// intersection phase of the query when ia is set), else tighten tau.  A final list that overflowed
// is not trustworthy: that query is searched exactly over all own edges right here (fb), which is
// rare (thousands of edges within tau: tied distances) and slow, but needs no further launch.
template <int DT>
__global__ __launch_bounds__(256) void knn_select_kernel(uint64_t *__restrict__ cand, int32_t *__restrict__ cnt,
                                                         int K, int final_level, float *__restrict__ tau, int QS,
                                                         uint64_t *__restrict__ out_keys,
                                                         int32_t *__restrict__ ovf, int32_t *__restrict__ dbg_cnt,
                                                         search_args fb, inter_args ia, int S,
                                                         const double *__restrict__ blockstats, int nblocks,
                                                         double *__restrict__ stats, int32_t *__restrict__ only = nullptr) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float *qs = reinterpret_cast<float *>(smem_raw);  // LD floats (exact search only)
    __shared__ uint64_t best[GH_EXTRACT_MAX_K];
    __shared__ uint64_t best2[GH_EXTRACT_MAX_K];
    __shared__ uint64_t red[4 * GH_EXTRACT_MAX_K];
    static_assert(GH_CAND_CAP <= 2 * GH_LIST_HALF, "block_extract_list takes two halves");
    const int64_t qi = blockIdx.x;
    if (qi >= S) {
        // Workgroups past the queries (single-rank fused steps): column qi - S of the fused kernel's per-workgroup
        // sums -> stats, in a fixed order.  It depends on the fused kernel only, so it runs beside the selection
        // instead of in front of the touched-row corrections of stats_fix_kernel (forces.hip).
        gh_reduce_stats_column(blockstats, nblocks, (int)(qi - S), stats, reinterpret_cast<double *>(red));
        return;
    }
    if (only && !only[qi]) return;   // second launch behind knn_select_wave_kernel: the queries it left
    // the first 1024 list slots are fetched before the list's length is known (the list is allocated whole: slots past the
    // end hold stale keys, masked below): one memory round trip instead of two in front of the extraction
    uint64_t pre[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pre[j] = cand[qi * GH_CAND_CAP + j * 256 + threadIdx.x];
    const int c = cnt[qi * GH_CNT_STRIDE];
    __syncthreads();
    if (threadIdx.x == 0) { cnt[qi * GH_CNT_STRIDE] = 0; dbg_cnt[qi] = c; if (only) only[qi] = 0; }
    if (__builtin_expect(c > GH_CAND_CAP || c < K, 0)) {   // (cold: laid out behind the usual path)
        if (!final_level) return;  // tau keeps its previous (still valid, looser) value
        if (threadIdx.x == 0) ovf[qi] = 1;
        block_search_query(fb, qi, K, qs, best, red);
    } else if (c <= 1024 && !(K > 16 && c <= 512)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pre[j] = j * 256 + (int)threadIdx.x < c ? pre[j] : GH_KEY_INF;
        block_extract_smallest<4>(pre, K, best, red);
    } else {
        block_extract_list(cand + qi * GH_CAND_CAP, c, K, best, best2, red);
    }
    if (final_level) {
        for (int i = threadIdx.x; i < K; i += blockDim.x) out_keys[qi * K + i] = best[i];
        __shared__ gh_pair_list pairs;
        if (ia.pos) intersect_query<DT>(ia, qi, best, &pairs);
    } else if (threadIdx.x == 0) {
        tau[qi * QS] = gh_key_d2(best[K - 1]);
    }
}

// The same selection (final level) with ONE WAVE per query, for thousands of queries on the unfused paths: a query's
// selection is a chain of dependent memory round trips (list length, keys, the pairs' edges, their rows, atomics) during
// which a 256-thread workgroup mostly waits; with a wave per query four times as many queries are in flight per CU
// (16384 queries, 1 M vertices: 320 -> 99 us together with the per-query runs of the touched list).  Takes lists of K .. 1024 keys; anything else (overflow, too few
// candidates, longer lists) is flagged in `redo` with its counter untouched and done by knn_select_kernel right behind.
template <int DT>
__global__ __launch_bounds__(64) void knn_select_wave_kernel(const uint64_t *__restrict__ cand, int32_t *__restrict__ cnt, int K,
                                                            uint64_t *__restrict__ out_keys, int32_t *__restrict__ dbg_cnt,
                                                            int32_t *__restrict__ redo, inter_args ia, int S) {
    constexpr int NT = 64;
    __shared__ uint64_t best[GH_EXTRACT_MAX_K];
    __shared__ uint64_t red[2 * NT];   // K <= 128 keys of one wave's extraction, or the 2 * NT keys the rank form stages
    __shared__ gh_pair_list pairs;
    const int64_t qi = blockIdx.x;
    if (qi >= S) return;
    uint64_t pre[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pre[j] = cand[qi * GH_CAND_CAP + j * NT + threadIdx.x];
    const int c = cnt[qi * GH_CNT_STRIDE];
    if (c > 16 * NT || c < K) {
        if (threadIdx.x == 0) redo[qi] = 1;
        return;
    }
    __syncthreads();
    if (threadIdx.x == 0) { cnt[qi * GH_CNT_STRIDE] = 0; dbg_cnt[qi] = c; }
    if (c <= 4 * NT && !(K > 16 && c <= 2 * NT)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pre[j] = j * NT + (int)threadIdx.x < c ? pre[j] : GH_KEY_INF;
        block_extract_smallest<4, NT>(pre, K, best, red);
    } else {
        block_extract_adaptive<16, NT>(cand + qi * GH_CAND_CAP, c, K, best, red);
    }
    for (int i = threadIdx.x; i < K; i += NT) out_keys[qi * K + i] = best[i];
    if (ia.pos) intersect_query<DT>(ia, qi, best, &pairs);
}

// The per-query runs knn_select_wave_kernel's intersection phase left -> the touched list: one workgroup adds the S run
// lengths up, reserves the whole stretch with ONE atomic on the shared counter and hands every query its start; the copy
// follows.  (A reservation per query was 16 K returning atomics on one address: 180 us of a 206 us launch.)
__global__ __launch_bounds__(1024) void knn_touched_prefix_kernel(const int32_t *__restrict__ tq_count, int S, int32_t *__restrict__ tq_base,
                                                                  int32_t *__restrict__ tcount) {
    // query q = round * 1024 + t: consecutive threads read consecutive counts; a wave scans its 64 values with shuffles,
    // the 16 wave totals are scanned by every thread from LDS, the running offset carries over the rounds (the first form
    // -- a strided chunk per thread, a twenty-barrier scan -- took 26 us at 16 K queries)
    __shared__ int wtot[16];
    __shared__ int base0;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    int total = 0;
    for (int r0 = 0; r0 < S; r0 += 1024) total += r0 + t < S ? tq_count[r0 + t] : 0;   // (first pass: the grand total)
    {
        int v = total;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) wtot[w] = v;
    }
    __syncthreads();
    if (t == 0) {
        int sum = 0;
        for (int i = 0; i < 16; ++i) sum += wtot[i];
        base0 = sum > 0 ? atomicAdd(tcount, sum) : 0;
    }
    __syncthreads();
    int carry = base0;
    for (int r0 = 0; r0 < S; r0 += 1024) {
        const int q = r0 + t;
        const int c = q < S ? tq_count[q] : 0;
        int incl = c;   // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        __syncthreads();   // wtot of the previous round has been read by everyone
        if (lane == 63) wtot[w] = incl;
        __syncthreads();
        int before = 0, round_total = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { before += i < w ? wtot[i] : 0; round_total += wtot[i]; }
        if (q < S) tq_base[q] = carry + before + incl - c;
        carry += round_total;
    }
}
__global__ __launch_bounds__(256) void knn_touched_copy_kernel(int32_t *__restrict__ tq_count, const int32_t *__restrict__ tq_base,
                                                              const int32_t *__restrict__ tq_touched, int S, int k4,
                                                              int32_t *__restrict__ touched) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t q = i / k4;
    const int slot = (int)(i % k4);
    if (q >= S) return;
    if (slot < tq_count[q]) touched[tq_base[q] + slot] = tq_touched[q * k4 + slot];
}

// Merge the per-rank key lists (world, S, K) into the K globally best keys per query (S, K).
// Only launched for world > 1; the intersection phase reads keys and drops column 0 itself
// (pt.py:421: knn_indices[:, 1:]).
template <int DT>
__global__ __launch_bounds__(256) void knn_merge_kernel(const uint64_t *__restrict__ gathered, int world, int64_t S,
                                                        int K, uint64_t *__restrict__ merged, inter_args ia) {
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
    block_sort(buf, n2);
    for (int c = threadIdx.x; c < K; c += blockDim.x) merged[qi * K + c] = buf[c];
    __shared__ gh_pair_list pairs;
    if (ia.pos) intersect_query<DT>(ia, qi, buf, &pairs);  // the k candidate pairs of this query, same launch (pt.py:638-774)
}

template <int D, int R>
void launch_scan(gh_engine *h, const float *mid, int64_t M, int64_t mem_stride, int64_t id_stride) {
    const int64_t per = 256 * R;
    const int64_t tiles = (M + per - 1) / per;
    // enough workgroups to fill 256 CUs: split the queries when there are few tiles
    int groups = (int)((1024 + tiles - 1) / tiles);
    const int min_groups = (int)((h->S + GH_SCAN_QGROUP - 1) / GH_SCAN_QGROUP);
    if (groups < min_groups) groups = min_groups;
    if (groups > (int)h->S) groups = (int)h->S;
    int qgroup = (int)((h->S + groups - 1) / groups);
    if (qgroup > GH_SCAN_QGROUP) qgroup = GH_SCAN_QGROUP;
    groups = (int)((h->S + qgroup - 1) / qgroup);
    knn_scan_kernel<D, R><<<dim3((unsigned)tiles, (unsigned)groups), dim3(256), 0, h->stream>>>(
        mid, h->part.edge_lo, h->d_own_eids, M, mem_stride, id_stride, h->d_q, h->d_qscan, (int)h->S, qgroup, h->d_cand, h->d_cnt,
        h->cdist ? 1 : 0);
}

template <int R>
void launch_scan_d(gh_engine *h, const float *mid, int64_t M, int64_t mem_stride, int64_t id_stride) {
    switch (h->D) {
        case 2: launch_scan<2, R>(h, mid, M, mem_stride, id_stride); break;
        case 3: launch_scan<3, R>(h, mid, M, mem_stride, id_stride); break;
        case 4: launch_scan<4, R>(h, mid, M, mem_stride, id_stride); break;
        default:
            if (h->LD == 8) launch_scan<8, (R > 4 ? 4 : R)>(h, mid, M, mem_stride, id_stride);
            else launch_scan<16, (R > 2 ? 2 : R)>(h, mid, M, mem_stride, id_stride);
    }
}

// mid == nullptr: gather the endpoints from positions instead (slow; the exact fallback only).
search_args make_search_args(gh_engine *h, const float *mid, int64_t M, int64_t mem_stride, int64_t id_stride) {
    return search_args{mid, h->d_pos, h->d_edges, h->d_own_eids, h->LD, h->D, h->part.edge_lo, M, mem_stride, id_stride,
                       h->d_q, gh_qs(h->D, h->LD)};
}

// with_intersect only takes effect in the extraction kernel (K <= GH_EXTRACT_MAX_K).
void launch_block_select(gh_engine *h, const float *mid, int64_t M, int64_t mem_stride, int64_t id_stride,
                         const int32_t *only_flagged, uint64_t *out_keys, bool write_tau, bool with_intersect) {
    const int QS = gh_qs(h->D, h->LD);
    float *tau_out = write_tau ? h->d_q + gh_qtau(h->D, h->LD) : nullptr;
    if (h->K <= GH_EXTRACT_MAX_K) {
#define GH_BSEL(DD)                                                                                             \
    knn_block_select_kernel<DD><<<dim3((unsigned)h->S), dim3(256), sizeof(float) * (size_t)h->LD, h->stream>>>(       \
        make_search_args(h, mid, M, mem_stride, id_stride), h->K, only_flagged, out_keys, tau_out,                     \
        make_inter_args(h, with_intersect))
        if (with_intersect) { GH_DISPATCH_DIM(h->D, GH_BSEL) } else { GH_BSEL(0); }   // (no intersection phase: nothing depends on the dimension at compile time)
#undef GH_BSEL
    } else {
        const size_t smem = sizeof(uint64_t) * GH_SEL_BUF + sizeof(float) * (size_t)h->LD;
        knn_block_select_sort_kernel<<<dim3((unsigned)h->S), dim3(256), smem, h->stream>>>(
            mid, h->d_pos, h->d_edges, h->d_own_eids, h->LD, h->D, h->part.edge_lo, M,
            mem_stride, id_stride, h->d_q, QS, h->K, only_flagged, out_keys, tau_out);
    }
}

// Stride of the threshold subset.  The set-up evaluates M/stride subset edges against every query (inside the
// normalise launch: ~0.1 us per 1000 rows at 256 queries beyond what the streaming part hides), the final pass then
// meets ~K*stride candidates per query, each a divergent exact re-check + LDS append in the scan and a key in the
// selection.  Measured optimum (rr1m, M = 4M, S*K = 2816): 64 (178 us per iteration; 40: 182, 100: 181); M = 400K: 20.
// That is sqrt(c * M / (S*K)) with c = 3 for the MFMA form of the scan (its hits cost less than half as much as the
// packed-VALU form's, whose balance sits lower).  Bounds: the list holds GH_CAND_CAP candidates (mean K*stride kept
// <= 4096), a workgroup parks its hits in LDS (mean min(S,256)*K*stride*tile/M per query group kept near 300 for a buffer of >= 512), and the
// subset must keep well over K groups of GH_THR_GSIZE rows.
int64_t subset_stride(int64_t Mtot, int K, int64_t S, int tile, bool mfma) {
    // beyond 256 queries the threshold kernel's workgroups no longer run all at once and its cost grows
    // with S like the hits do, so the balance point stops moving
    // (the MFMA form's hits cost less than half as much -- ~0.1 us per unit of stride -- so its balance sits higher)
    int64_t r = (int64_t)sqrt((mfma ? 3.0 : 2.0) * (double)Mtot / ((double)(S < 256 ? S : 256) * K));
    const int64_t by_list = 4096 / K;
    if (r > by_list) r = by_list;
    if (r > 256) r = 256;
    // (per query GROUP: the fused kernels empty the buffer between groups once it is a quarter full)
    const int64_t by_hits = (int64_t)(300.0 * (double)Mtot / ((double)(S < 256 ? S : 256) * K * tile));
    if (r > by_hits) r = by_hits;
    if (r < 2) r = 2;
    while (r > 2 && Mtot / r < 8 * (int64_t)K * GH_THR_GSIZE) r /= 2;  // at least 8 K groups: tau stays close to the subset's K-th smallest
    return r;
}

// fb_mid: midpoint rows for the exact search of overflowed queries, or null (gather the endpoints).
gh_status launch_select(gh_engine *h, bool final_level, bool with_intersect, const float *fb_mid) {
    // the column sums of the fused kernel's workgroup partials ride along (stats_fix_kernel then skips them)
    // (not with thousands of queries: there the wave-per-query form below is worth more than the ride, and stats_fix_kernel
    // adds the column sums up itself)
    const bool reduce = final_level && h->new0_ready && h->rows > 0 && h->LD <= 16 && h->S < 2048;
    gh_scope t(h, with_intersect ? "knn_select_intersect" : "knn_select");
    // thousands of queries, no column sums riding along: a wave per query first, the workgroup form for what it leaves
    const bool wave = final_level && !reduce && h->S >= 2048;
    bool wave_tq = false;
    if (wave) {
#define GH_SELW(DD)                                                                                                      \
    knn_select_wave_kernel<DD><<<dim3((unsigned)h->S), dim3(64), 0, h->stream>>>(h->d_cand, h->d_cnt, h->K, h->d_partial,   \
                                                                                 h->d_dbg_cnt + (size_t)h->S, h->d_sel_redo, \
                                                                                 iaw, (int)h->S)
        inter_args iaw = make_inter_args(h, with_intersect);
        wave_tq = with_intersect && h->k <= 127 && h->D >= 2 && h->LD <= 16;   // the lanes-per-coordinate form of the phase (intersect_query)
        if (wave_tq) { iaw.tq_count = h->d_tq_count; iaw.tq_touched = h->d_tq_touched; }
        if (with_intersect) { GH_DISPATCH_DIM(h->D, GH_SELW) } else { GH_SELW(0); }
#undef GH_SELW
    }
#define GH_SEL(DD)                                                                                                                          \
    knn_select_kernel<DD><<<dim3((unsigned)h->S + (reduce ? 2u * (unsigned)h->LD : 0u)), dim3(256), sizeof(float) * (size_t)h->LD, h->stream>>>( \
        h->d_cand, h->d_cnt, h->K, final_level ? 1 : 0, h->d_q + gh_qtau(h->D, h->LD), gh_qs(h->D, h->LD),                                     \
        h->d_partial, h->d_ovf, h->d_dbg_cnt + (size_t)(final_level ? 1 : 0) * h->S,                                                           \
        make_search_args(h, fb_mid, h->own_count, 1, 1), make_inter_args(h, with_intersect), (int)h->S,                                        \
        h->d_blockstats, h->n_vblocks, h->d_stats, wave ? h->d_sel_redo : nullptr)
    if (with_intersect) { GH_DISPATCH_DIM(h->D, GH_SEL) } else { GH_SEL(0); }
#undef GH_SEL
    if (wave_tq) {
        const int k4 = 4 * h->k;
        knn_touched_prefix_kernel<<<dim3(1), dim3(1024), 0, h->stream>>>(h->d_tq_count, (int)h->S, h->d_tq_base, h->d_tcount);
        knn_touched_copy_kernel<<<dim3((unsigned)(((int64_t)h->S * k4 + 255) / 256)), dim3(256), 0, h->stream>>>(
            h->d_tq_count, h->d_tq_base, h->d_tq_touched, (int)h->S, k4, h->d_touched);
        GH_HIP(hipMemsetAsync(h->d_tq_count, 0, sizeof(int32_t) * (size_t)h->S, h->stream));
    }
    GH_LAUNCH_CHECK();
    h->stats_reduced = reduce;
    return GH_OK;
}

}  // namespace

// Edges this rank searches: the range [edge_lo, edge_hi), or the list d_own_eids (hashed ownership).
static int64_t own_edges(const gh_engine *h) { return h->own_count; }

bool gh_knn_scan_path(const gh_engine *h) {
    const int64_t Mtot = own_edges(h);
    return Mtot >= GH_SCAN_MIN_EDGES && h->LD <= 16 && h->D >= 2 && h->Ksel <= GH_EXTRACT_MAX_K && h->S <= 0x7FFFFFFF;
}

// Sample ids (if still pending), query records, list reset and -- on the scan path -- the compact
// threshold subset: one launch.
gh_setup_args gh_make_setup_args(gh_engine *h, int mode, int32_t *sampled, uint64_t iter) {
    if (h->thr_stride == 0) {  // fixed at the first use: d_gmin is sized from it
        const int64_t Mtot = own_edges(h);
        const bool scan = gh_knn_scan_path(h) && !gh_grid_path(h) && !gh_ivf_path(h);   // the grid / IVF searches take their thresholds from their own structures
        h->thr_stride = scan ? subset_stride(Mtot, h->Ksel, h->S, gh_fused_tile(h), gh_fused_uses_mfma(h)) : 1;
        h->thr_M1 = scan ? (Mtot + h->thr_stride - 1) / h->thr_stride : 0;
    }
    const int tiles = (int)((h->thr_M1 + GH_THR_TILE - 1) / GH_THR_TILE);
    // replayed iterations: the number of the iteration being set up = device counter + (iter - h->iter - 1): the counter
    // was moved to "this iteration + 1" by stats_fix_kernel before the normalise launch that carries the set-up
    const bool dev = h->graph_capturing;
    return gh_setup_args{h->d_edges, sampled, mode, h->E, h->prm.seed, dev ? iter - h->iter - 1 : iter, h->S, h->D, h->LD, h->d_q, h->d_cnt,
                         h->d_ovf, h->part.edge_lo, h->d_own_eids, reinterpret_cast<const int2 *>(h->d_sub_uv), h->thr_M1, h->thr_stride, h->d_gmin,
                         (int64_t)tiles * GH_THR_GROUPS, tiles, h->tau_embedded ? h->d_tau_flag : nullptr, dev ? h->d_iter : nullptr};
}

// Workgroups of 256 threads a set-up takes (knn_setup_kernel, or the head of a normalise launch).
unsigned gh_setup_blocks(const gh_setup_args &a) {
    return a.tiles > 0 ? (unsigned)a.tiles : (unsigned)((a.S + 255) / 256);
}
int64_t gh_gmin_floats(const gh_engine *h) {
    const gh_setup_args a = gh_make_setup_args(const_cast<gh_engine *>(h), 0, nullptr, 0);
    if (gh_ivf_path(h)) return h->S * 256 + 4;   // GH_IVF_GROUPS minima per query, written by ivf_probe_kernel
    if (a.tiles == 0) return 4;
    return h->S * a.Gpad + 4;
}

gh_status gh_knn_prepare(gh_engine *h) {
    const int mode = h->sample_pending ? h->sample_mode : 0;
    h->sample_pending = false;
    // the previous normalise launch may already have done this iteration's set-up (forces.hip)
    const bool done = h->presetup_valid && h->presetup_mode == mode && h->presetup_iter == h->iter &&
                      (mode != 0 || h->presetup_ids == h->d_sampled_cur);
    h->presetup_valid = false;
    h->tcount_reset_pending = done;  // the stand-alone kernel resets d_tcount; else the threshold kernel does
    if (done) return GH_OK;
    const gh_setup_args a = gh_make_setup_args(h, mode, h->d_sampled_cur, h->iter);
    gh_scope t(h, "knn_setup");
    knn_setup_kernel<<<dim3(gh_setup_blocks(a)), dim3(256), 0, h->stream>>>(h->d_pos, a, h->d_tcount, h->d_qexact);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

// Arguments of the threshold computation (tau_core.h) for this iteration.
gh_tau_args gh_make_tau_args(gh_engine *h) {
    const gh_setup_args a = gh_make_setup_args(h, 0, h->d_sampled_cur, h->iter);
    gh_tau_args t{};
    t.gmin = reinterpret_cast<const uint32_t *>(h->d_gmin);
    t.Gpad = a.Gpad;
    t.D = h->D;
    t.QS = gh_qs(h->D, h->LD);
    t.QT = gh_qtau(h->D, h->LD);
    t.K = h->Ksel;   // GH_DIST_CDIST: a bound for K + 1 neighbours (cdist.hip)
    t.S = (int)h->S;
    t.qt = h->d_q;
    t.qscan = h->d_qscan;
    t.qA = reinterpret_cast<_Float16 *>(h->d_qA);
    t.qA_kb = gh_fused_mfma_kb(h);   // -1: no MFMA form in use
    t.qexact = h->d_qexact;
    t.tcount_reset = h->tcount_reset_pending ? h->d_tcount : nullptr;
    return t;
}

// tau of every query from the group minima the set-up left in d_gmin.  Needs gh_knn_scan_path(h).
gh_status gh_knn_thresholds(gh_engine *h, int64_t groups) {
    gh_scope t(h, "knn_tau");
    gh_tau_args a = gh_make_tau_args(h);
    if (groups > 0) a.Gpad = groups;
    const int nv = gh_tau_nv_host(a.Gpad);
#define GH_TAU_FORM(NVv)                                                                            \
    if (a.qA_kb == 0) knn_tau_kernel<NVv, 0><<<dim3((unsigned)h->S), dim3(64), 0, h->stream>>>(a);  \
    else knn_tau_kernel<NVv, -1><<<dim3((unsigned)h->S), dim3(64), 0, h->stream>>>(a);
    if (nv == 8) { GH_TAU_FORM(8) } else if (nv == 16) { GH_TAU_FORM(16) } else { GH_TAU_FORM(32) }
#undef GH_TAU_FORM
    GH_LAUNCH_CHECK();
    return GH_OK;
}

// K best keys of every query from its final candidate list; queries whose list overflowed are
// redone exactly over all own edges in the same launch (from d_mid when have_mid, else by
// gathering endpoints).
// fuse_intersect (single-rank steps): the same launches also run the intersection phase of each
// query they finish (h->intersect_done tells the caller).
gh_status gh_knn_finish(gh_engine *h, bool have_mid, bool fuse_intersect) {
    if (h->cdist) return gh_knn_finish_cdist(h, false, fuse_intersect);   // the reference's cdist + topk rows (cdist.hip)
    const bool fuse = fuse_intersect && h->K <= GH_EXTRACT_MAX_K;
    GH_TRY_ST(launch_select(h, true, fuse, have_mid ? h->d_mid : nullptr));
    h->intersect_done = fuse;
    return GH_OK;
}

// Unfused search over the materialised midpoints d_mid -> d_partial.
gh_status gh_knn_local(gh_engine *h, bool fuse_intersect) {
    const int64_t Mtot = own_edges(h);
    GH_TRY_ST(gh_knn_prepare(h));
    if (!gh_knn_scan_path(h)) {
        if (h->cdist) return gh_knn_finish_cdist(h, true, false);   // every query against all edges (cdist.hip)
        const bool fuse = fuse_intersect && h->K <= GH_EXTRACT_MAX_K;
        gh_scope t(h, "knn_block_select");
        launch_block_select(h, h->d_mid, Mtot, 1, 1, nullptr, h->d_partial, false, fuse);
        GH_LAUNCH_CHECK();
        h->intersect_done = fuse;
        return GH_OK;
    }
    GH_TRY_ST(gh_knn_thresholds(h));
    {
        gh_scope t(h, "knn_scan");
        launch_scan_d<8>(h, h->d_mid, Mtot, 1, 1);
        GH_LAUNCH_CHECK();
    }
    return gh_knn_finish(h, true, fuse_intersect);
}

gh_status gh_knn_merge(gh_engine *h, const uint64_t *gathered, int world) {
    if (h->cd_part) return gh_knn_merge_cdist(h, gathered, world);   // (S, K + 2) records per rank: the rows are decided now (cdist.hip)
    if (world == 1) {  // nothing to merge: the intersection phase reads this rank's keys directly
        h->d_keys_cur = gathered;
        return GH_OK;
    }
    const int total = world * h->K;
    int n2 = 2;
    while (n2 < total) n2 <<= 1;
    if ((size_t)n2 * sizeof(uint64_t) > 60 * 1024) {
        h->err = "world * (n_neighbors + 1) too large for the merge kernel";
        return GH_ERR_INVALID;
    }
    gh_scope t(h, "knn_merge_intersect");
#define GH_MERGE(DD)                                                                                      \
    knn_merge_kernel<DD><<<dim3((unsigned)h->S), dim3(256), sizeof(uint64_t) * (size_t)n2, h->stream>>>(      \
        gathered, world, h->S, h->K, h->d_merged, make_inter_args(h, !h->intersect_done))
    if (!h->intersect_done) { GH_DISPATCH_DIM(h->D, GH_MERGE) } else { GH_MERGE(0); }
#undef GH_MERGE
    GH_LAUNCH_CHECK();
    h->d_keys_cur = h->d_merged;
    h->intersect_done = true;
    return GH_OK;
}

// F4 (influence.py:28-37): the K vertices with the largest radial distance.  Key = (~bits(radial) <<
// 32) | vertex: non-negative floats order like their bit patterns, so the K smallest keys are the K
// largest distances, ties on the smaller vertex id.  Level 0: workgroup b takes vertices b*256+t,
// stride gridDim*256, keeps its K best as it goes (chunks of 8 keys per thread) -> part[b][K];
// level 1 (one workgroup): K best of all parts.
__global__ __launch_bounds__(256) void radial_topk_kernel(const float *__restrict__ pos, const int32_t *__restrict__ order,
                                                         int64_t n, int D, int LD, int K, uint64_t *__restrict__ part) {
    __shared__ uint64_t best[GH_EXTRACT_MAX_K];
    __shared__ uint64_t red[4 * GH_EXTRACT_MAX_K];
    constexpr int NPT = 8;
    for (int i = threadIdx.x; i < K; i += 256) best[i] = GH_KEY_INF;
    __syncthreads();
    const int64_t step = (int64_t)gridDim.x * 256 * NPT;
    for (int64_t base = (int64_t)blockIdx.x * 256 * NPT; base < n; base += step) {
        const uint64_t tk = best[K - 1];
        uint64_t keys[NPT + 1];
        int any = 0;
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int64_t v = base + j * 256 + threadIdx.x;
            uint64_t key = GH_KEY_INF;
            if (v < n) {
                const float *row = pos + (order ? (int64_t)order[v] : v) * LD;
                float s = 0.0f;
                for (int d = 0; d < D; ++d) s = s + row[d] * row[d];  // numpy: multiply, then add, in order
                const float r = sqrtf(s);
                key = ((uint64_t)(~__float_as_uint(r)) << 32) | (uint32_t)v;
                if (key < tk) any = 1; else key = GH_KEY_INF;
            }
            keys[j] = key;
        }
        if (__syncthreads_or(any)) {
            keys[NPT] = threadIdx.x < K ? best[threadIdx.x] : GH_KEY_INF;
            __syncthreads();
            block_extract_smallest<NPT + 1>(keys, K, best, red);
        }
    }
    for (int i = threadIdx.x; i < K; i += 256) part[(int64_t)blockIdx.x * K + i] = best[i];
}

__global__ __launch_bounds__(256) void radial_topk_merge_kernel(const uint64_t *__restrict__ part, int nparts, int K,
                                                               int32_t *__restrict__ ids) {
    __shared__ uint64_t best[GH_EXTRACT_MAX_K];
    __shared__ uint64_t red[4 * GH_EXTRACT_MAX_K];
    constexpr int NPT = 8;
    for (int i = threadIdx.x; i < K; i += 256) best[i] = GH_KEY_INF;
    __syncthreads();
    const int total = nparts * K;
    for (int base = 0; base < total; base += 256 * NPT) {
        uint64_t keys[NPT + 1];
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int i = base + j * 256 + threadIdx.x;
            keys[j] = i < total ? part[i] : GH_KEY_INF;
        }
        keys[NPT] = threadIdx.x < K ? best[threadIdx.x] : GH_KEY_INF;
        __syncthreads();
        block_extract_smallest<NPT + 1>(keys, K, best, red);
    }
    for (int i = threadIdx.x; i < K; i += 256) ids[i] = (int32_t)(uint32_t)best[i];
}

gh_status gh_radial_topk_device(gh_engine *h, int K, uint64_t *d_part, int nparts, int32_t *d_ids) {
    radial_topk_kernel<<<dim3((unsigned)nparts), dim3(256), 0, h->stream>>>(h->d_pos, h->d_order, h->n, h->D, h->LD, K, d_part);
    radial_topk_merge_kernel<<<dim3(1), dim3(256), 0, h->stream>>>(d_part, nparts, K, d_ids);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

// Plain point-set KNN (the reference's _compute_knn_chunked / _compute_knn_torch, pt.py:426-483,
// 543-593, as a library call): K smallest (dist2, id) keys of every query row among the
// reference rows, by the per-query kernels above with the point arrays standing in for the
// midpoint array (row stride D) and the query records (QS = D).
gh_status gh_knn_points_device(hipStream_t stream, const float *d_q, int64_t nq, const float *d_ref, int64_t nref,
                               int D, int K, uint64_t *d_keys, std::string *err) {
    if (nq == 0) return GH_OK;
    inter_args none{};
    if (K <= GH_EXTRACT_MAX_K) {
        knn_block_select_kernel<0><<<dim3((unsigned)nq), dim3(256), sizeof(float) * (size_t)D, stream>>>(
            search_args{d_ref, nullptr, nullptr, nullptr, D, D, 0, nref, 1, 1, d_q, D}, K, nullptr, d_keys, nullptr, none);
    } else {
        const size_t smem = sizeof(uint64_t) * GH_SEL_BUF + sizeof(float) * (size_t)D;
        if (smem > 64 * 1024) { *err = "dimension too large for the point KNN kernel"; return GH_ERR_INVALID; }
        knn_block_select_sort_kernel<<<dim3((unsigned)nq), dim3(256), smem, stream>>>(
            d_ref, nullptr, nullptr, nullptr, D, D, 0, nref, 1, 1, d_q, D, K, nullptr, d_keys, nullptr);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { *err = std::string("kernel launch: ") + hipGetErrorString(e); return GH_ERR_HIP; }
    return GH_OK;
}

#include "grid_core.h"
