// GH_DIST_CDIST: neighbour rows as the reference's PyTorch-CPU backend gets them from torch.cdist + torch.topk
// (pt.py:580-583), value for value and tie for tie.
//
// What ATen computes (PyTorch 2.10, ATen/native/Distance.cpp _euclidean_dist, taken for p = 2 when either side has more
// than 25 rows; ATen/native/TopKImpl.h topk_impl_loop) -- tests/test_oracle_reference_fullsize.py pins
// oracle/aten_cdist_topk.cpp, the CPU statement of the same, against the reference's own ids at 100 K and 1 M vertices:
//     |x|^2 = sum over d of fl(x_d * x_d), added left to right;
//     acc = fma(-2 q_d, m_d, acc) for d = 0 .. D-1 from 0;  acc = fl(acc + |q|^2);  acc = fl(acc + |m|^2);
//     value = sqrt(max(acc, 0));
//     topk = std::partial_sort of (value, index) pairs in index order with a comparator on the value alone (K * 64 <= E):
//     equal values come out in the order the heap leaves them, which depends on every element that ever entered it.
// The fp32 quantum of acc is ulp(|q|^2 + |m|^2): near neighbours whose exact distances differ by less swap or tie, and
// the sampled edge's own value is not 0, so column 0 -- dropped blindly, pt.py:417-421 -- need not be the edge itself.
//
// How the engine gets there without an (S, E) matrix:
//   1. the filtered scan runs unchanged with a threshold for K + 1 neighbours: the candidate list of a query holds every
//      edge whose EXACT squared distance (fma chain) is <= tau;
//   2. a candidate is parked with the formula's value in its key (the scan has both midpoints at hand: scan_core.h
//      gh_aten_cdist; round 3 re-valued the lists in the selection launch: two dependent row gathers); knn_select_cdist_kernel
//      extracts the K + 1 smallest (value, id) keys and PROVES the list complete for cdist's ranking: an edge outside
//      has exact distance > tau, hence acc > tau (1 - (7D + 15) u) - 3 (3D + 6) u |q|^2, u = 2^-24 (derivation at
//      cdist_lower_bound); if the (K+1)-th smallest value clears that bound and the K + 1 values are pairwise
//      different, the row is decided: ascending values, no heap order involved.  Such a row's k candidate pairs go
//      through the intersection phase in the same launch (single-rank steps);
//   3. the other rows -- a tie among the K + 1 smallest values (about one row in a hundred at a million vertices), or a
//      list that could not be proven complete -- are LISTED and get partial_sort's heap replayed.  Element i enters the
//      heap iff value_i < (K-th smallest of values 0 .. i-1), a bound that only falls with i.  Round 4: that bound is
//      used twice so that the replay never sees more than a sliver of the E values (round 3 valued all E edges for every
//      listed row -- 86 us of gathers at a million vertices -- and one wave walked them for 188 us):
//        a. the candidate list already holds every edge with a safely small value; by the id P where K of them (clearing
//           the bound of (2)) have been seen -- about E K / |list| = E / stride, rounded up to one of 1024 equal id
//           ranges -- the heap's maximum is <= the largest of those K values, so every element that enters from P on
//           IS in the list.  The select workgroup of a listed row finds P from a histogram of the list's ids and puts the
//           list's tail (ids >= P) in id order (range start + rank inside the range), in registers / LDS;
//        b. cdist_prefix_kernel values only the ids below P (chip-wide; minima per 64 ids on the side);
//        c. cdist_replay_kernel (one workgroup per listed row; wave 0 replays, all four waves list and fetch) walks the
//           prefix -- skipping 64-id chunks whose minimum cannot enter, the live ones staged through LDS 128 at a time as
//           value keys, the first 128 chunks fetched ahead of the heap build -- then the sorted tail, pops the heap into
//           ascending order and runs the row's intersection phase.  The replay is ONE wave's instruction stream, ~150
//           entering elements per row at a million vertices, so the step is written for its instruction count: K <= 16
//           keys live in pinned scalar register pairs and an element enters through a generated block of ~13 scalar
//           instructions (cdist_heap_asm.h); K <= 64 one element per lane, an element enters in one data-parallel step
//           (children through ds_bpermute, the sift path through v_readlane); beyond, in LDS.
//      A row whose list overflowed or could not be proven complete takes P = E (no tail): the round-3 full pass.
//   EVERY row of a graph too small for the scan takes (b) + (c) with P = E.
//   Rows of graphs with K * 64 > E (tiny ones) are ranked by std::nth_element + std::sort in ATen: cdist_nth_kernel replays
//   libstdc++'s introselect and introsort on their values (E <= 8000; beyond, equal values are ordered by id and a row with a
//   tie is counted in gh_knn_cdist_stats).
//   Row partitions: a rank stops at (2) with its K + 1 best keys and a proof flag; after the all-gather
//   knn_merge_cdist_kernel decides the rows without a tie and lists the others, which every rank replays -- over the ids
//   below a bound P taken from the gathered keys alone (about E / world) plus the gathered keys behind it; P = E where a
//   rank could not prove its keys.
#include "common.h"
#include "engine.h"
#include "scan_core.h"
#include "select_core.h"
#include "cdist_heap_asm.h"

#include <algorithm>
#include <math.h>

namespace {

#define GH_CD_TILE 4096    /* chunk minima staged in LDS at a time (cdist_replay_kernel) */
#define GH_CD_BATCH 128    /* live chunks fetched together */
#define GH_CD_ALT 8192     /* a listed row's id-keyed copy of its candidate list starts here (lists of <= 8192 keys) */
#define GH_CD_NB 1024      /* buckets of the tail sort */
#define GH_CD_TAIL_LDS 1024 /* tail keys the replay keeps in LDS */

struct cdist_args {
    const float *pos;          // (n, LD) rows
    const int32_t *edges;      // (E, 2)
    int D, LD;
    int64_t E;
    const float *qt;           // query records (scan_core.h gh_qs): coordinates, then tau
    int QS, QT;
    int mm_form;               // ATen's matmul form (S > 25 or E > 25), else its direct kernel: sqrt of the exact-difference sum
};

// The listed rows: slot t < hdr[0] is query rare[1 + t]; ids [0, P[t]) are valued by cdist_prefix_kernel, the ct[t]
// candidates from P[t] on sit sorted by id at cand[q * GH_CAND_CAP + GH_CD_ALT ..] as (id << 32 | value bits).
struct cdist_rows {
    int32_t *rare;   // [1 .. S] (slot t at rare[1 + t])
    int32_t *P;      // (S)
    int32_t *ct;     // (S)
    // Counters of this search: [0] listed rows, [1] rows with a tie that ATen's nth_element path decides, [2] max P over
    // the slots.  Two sets, used alternately: a search zeroes the OTHER set (cdist_prefix_kernel) for the next one, so no
    // memset sits in the stream (each was a 4 us fill kernel plus its launch gap; two per iteration).
    int32_t *hdr, *hdr_next;
};

// |x|^2 as x.pow(2).sum(-1) rounds it: every square on its own, added left to right.
__device__ __forceinline__ float cdist_norm(const float *x, int D) {
    float s = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float sq = x[d] * x[d];
        s = s + sq;
    }
    return s;
}

// The value of pair (query q with |q|^2 = qn, edge e).  LDT: the row stride at compile time (4, 8, 16: vector row
// loads) or 0 (any stride, scalar loads).  Padding coordinates are 0 on both sides and change nothing
// (fma(-0, 0, acc) == acc, s + 0 == s), the loops stop at D all the same.
template <int LDT>
__device__ __forceinline__ float cdist_pair(const cdist_args &a, const float *q, float qn, int64_t e) {
    const int2 uv = reinterpret_cast<const int2 *>(a.edges)[e];
    float s = 0.0f, acc = 0.0f;
    if constexpr (LDT > 0) {
        float pu[LDT], pv[LDT];
        gh_load_row<LDT>(a.pos, uv.x, pu);
        gh_load_row<LDT>(a.pos, uv.y, pv);
#pragma unroll
        for (int d = 0; d < LDT; ++d) {
            if (d < a.D) {
                const float m = (pu[d] + pv[d]) / 2.0f;   // pt.py:785
                if (a.mm_form) {
                    const float sq = m * m;
                    s = s + sq;
                    acc = fmaf(q[d] * -2.0f, m, acc);
                } else {
                    const float t = q[d] - m;
                    acc = fmaf(t, t, acc);
                }
            }
        }
    } else {
        const float *pu = a.pos + (int64_t)uv.x * a.LD, *pv = a.pos + (int64_t)uv.y * a.LD;
        for (int d = 0; d < a.D; ++d) {
            const float m = (pu[d] + pv[d]) / 2.0f;
            if (a.mm_form) {
                const float sq = m * m;
                s = s + sq;
                acc = fmaf(q[d] * -2.0f, m, acc);
            } else {
                const float t = q[d] - m;
                acc = fmaf(t, t, acc);
            }
        }
    }
    if (a.mm_form) {
        acc = acc + qn;   // the matmul's last two terms: fma(|q|^2, 1, acc), fma(1, |m|^2, acc)
        acc = acc + s;
    }
    return sqrtf(fmaxf(acc, 0.0f)) + 0.0f;   // clamp_min(0).sqrt(); + 0: a -0 must not read as the largest key
}

// Lower bound of acc for every edge whose exact-chain squared distance exceeds tau.  With u = 2^-24, reals starred:
//   computed norms: |qn - qn*| <= D u qn*, same for the midpoint; the D + 2 roundings of the accumulation are each
//   <= u times a partial result <= 2 (qn* + mn*):   acc >= d2* - (3D + 4) u (qn* + mn*)   -- c = 3D + 6 below;
//   the exact chain: d2_fl <= d2* (1 + (D + 3) u), so d2_fl > tau gives d2* > tau (1 - (D + 3) u);
//   mn* <= 2 qn* + 2 d2*:   acc >= d2* (1 - 2 c u) - 3 c u qn*  >  tau (1 - (2c + D + 3) u) - 3 c u qn*,
// and qn* <= qn (1 + 2 D u).  Evaluated in double with a further percent of slack.
__device__ __forceinline__ double cdist_lower_bound(float tau, float qn, int D) {
    const double u = 5.9604644775390625e-08, c = 3.0 * D + 6.0;
    return (double)tau * (1.0 - 1.01 * (2.0 * c + D + 3.0) * u) - 3.05 * c * u * (double)qn;
}
// A value w whose square clears the bound: every edge OUTSIDE the candidate list has acc > bound >= w^2 (1 + 1e-6), hence a
// value more than four ulps above w.
__device__ __forceinline__ bool cdist_clears(float w, double bound) { return (double)w * (double)w * (1.0 + 1e-6) <= bound; }

// One workgroup per query: candidate list (exact d2 <= tau) -> the K smallest cdist keys when the list proves enough
// (then also the query's intersection phase, ia.pos != null); else the row is listed for the replay, with its prefix
// length P and its tail sorted by id.  Workgroups past the S queries reduce the fused kernel's column sums (nblocks > 0).
template <int DT>
__global__ __launch_bounds__(256) void knn_select_cdist_kernel(uint64_t *__restrict__ cand, int32_t *__restrict__ cnt, int K,
                                                               cdist_args a, uint64_t *__restrict__ out_keys,
                                                               int32_t *__restrict__ ovf, int32_t *__restrict__ dbg_cnt,
                                                               cdist_rows rr, inter_args ia, int S,
                                                               const double *__restrict__ blockstats, int nblocks,
                                                               double *__restrict__ stats,
                                                               int part_mode /* a row partition: only this rank's K + 1 best keys and
                                                               whether they are provably its K + 1 best -> out_keys (S, K + 2); ties
                                                               and listing are decided after the ranks' keys are merged */,
                                                               int force_ties /* the loop: list a row only for a tie that can change what
                                                               the intersection phase reads (below) */) {
    constexpr int LDT = DT <= 4 ? 4 : DT <= 8 ? 8 : 16;
    __shared__ uint64_t best[GH_EXTRACT_MAX_K];
    __shared__ uint64_t best2[GH_EXTRACT_MAX_K];
    __shared__ uint64_t red[4 * GH_EXTRACT_MAX_K];
    __shared__ float qs[16];
    __shared__ gh_pair_list pairs;
    __shared__ int bc[GH_CD_NB], bf[GH_CD_NB];
    __shared__ uint64_t tl[1024];
    __shared__ int wsum[4], s_bk, s_tail0;
    const int Ks = K + 1;
    const int64_t qi = blockIdx.x;
    if (qi >= S) {
        gh_reduce_stats_column(blockstats, nblocks, (int)(qi - S), stats, reinterpret_cast<double *>(red));
        return;
    }
    uint64_t *list = cand + qi * GH_CAND_CAP;
    // the first 1024 list slots are fetched before the list's length is known (as knn_select_kernel does)
    uint64_t pre[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pre[j] = list[j * 256 + threadIdx.x];
    const int c = cnt[qi * GH_CNT_STRIDE];
    if (threadIdx.x < 16) qs[threadIdx.x] = (int)threadIdx.x < a.D ? a.qt[qi * a.QS + threadIdx.x] : 0.0f;
    __syncthreads();
    if (threadIdx.x == 0) { cnt[qi * GH_CNT_STRIDE] = 0; dbg_cnt[qi] = c; }
    const float qn = cdist_norm(qs, a.D);
    const double bound = cdist_lower_bound(a.qt[qi * a.QS + a.QT], qn, a.D);
    int reason = (c > GH_CAND_CAP || c < Ks) ? 2 : 0;   // the list overflowed, or tau was not a bound for K + 1 edges
    const bool in_regs = c <= 1024 && Ks <= 16;          // the usual case: re-valued keys never leave the registers
    uint64_t keep[4] = {GH_KEY_INF, GH_KEY_INF, GH_KEY_INF, GH_KEY_INF};
    if (!reason) {
        // (the scan parked every candidate with ATen's cdist value in its key already: scan_core.h gh_aten_cdist)
        if (in_regs) {
            uint64_t keys[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j * 256 + (int)threadIdx.x < c) keep[j] = pre[j];
                keys[j] = keep[j];
            }
            block_extract_smallest<4>(keys, Ks, best, red);
        } else {
            block_extract_list(list, c, Ks, best, best2, red);
        }
        // A tie among the K + 1 smallest values leaves the ORDER of the equal ones to partial_sort's heap.  The intersection
        // phase reads the row as a set of pairs -- column 0 dropped (pt.py:417-421), the other k ids each paired with the sampled
        // edge (pt.py:668-699) -- so inside the loop (force_ties) only two ties can change a force: values 0 and 1 equal (WHICH
        // id is dropped) and values K - 1 and K equal (WHICH id is a member).  A tie strictly inside ranks 1 .. K - 1 permutes
        // columns of the same set: decided here, equal values in id order.  The per-phase call (gh_knn_midpoints) lists every tie
        // and returns the reference's rows column for column.
        int bad = 0;
        for (int i = threadIdx.x; i + 1 < Ks; i += 256)
            if (!force_ties || i == 0 || i == K - 1) bad |= (uint32_t)(best[i] >> 32) == (uint32_t)(best[i + 1] >> 32) ? 1 : 0;
        if (threadIdx.x == 0 && !cdist_clears(gh_key_d2(best[Ks - 1]), bound)) bad |= 2;   // the (K+1)-th smallest VALUE (a distance, not squared)
        // 1: a tie among the K + 1 smallest values, 2: the list is not provably complete (3: both)
        reason = (__syncthreads_or(bad & 1) ? 1 : 0) | (__syncthreads_or(bad & 2) ? 2 : 0);
    }
    if (part_mode) {
        // (reason bit 1 = a tie: for the merged list to judge; the keys are this rank's K + 1 smallest either way)
        const bool have = c <= GH_CAND_CAP && c >= Ks;
        for (int i = threadIdx.x; i < Ks; i += 256) out_keys[qi * (Ks + 1) + i] = have ? best[i] : GH_KEY_INF;
        if (threadIdx.x == 0) { out_keys[qi * (Ks + 1) + Ks] = have && !(reason & 2) ? 1ull : 0ull; ovf[qi] = reason; }
        return;
    }
    if (!reason) {
        for (int i = threadIdx.x; i < K; i += 256) out_keys[qi * K + i] = best[i];
        if (ia.pos) intersect_query<DT>(ia, qi, best, &pairs);
        return;
    }
    // ---- a listed row ----
    // Histogram of the candidates' ids over GH_CD_NB equal id ranges (low half of a counter: all of them, high half: those
    // that do not clear the bound); P = the end of the first range by which K clearing candidates have been seen; the
    // candidates of the later ranges are the tail, put in id order: start of the range + rank inside it (by counting).
    // Lists of up to 1024 keys (the usual case) never leave the registers / LDS until the sorted tail is stored.
    int64_t P = a.E;
    int ct = 0;
    if (reason == 1 && c <= GH_CD_ALT) {
        uint64_t *alt = list + GH_CD_ALT;
        const uint32_t W = (uint32_t)((a.E + GH_CD_NB - 1) / GH_CD_NB);
        for (int i = threadIdx.x; i < GH_CD_NB; i += 256) bc[i] = 0;
        if (threadIdx.x == 0) s_bk = -1;
        __syncthreads();
        auto file = [&](uint64_t key) {   // -> (id << 32 | value bits), counted in its range
            const float v = gh_key_d2(key);
            const uint32_t id = gh_key_id(key);
            atomicAdd(&bc[id / W], cdist_clears(v, bound) ? 1 : 0x10001);
            return ((uint64_t)id << 32) | __float_as_uint(v);
        };
        uint64_t mine[4] = {0ull, 0ull, 0ull, 0ull};
        if (in_regs) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j * 256 + (int)threadIdx.x < c) mine[j] = file(keep[j]);
        } else {
            for (int i = threadIdx.x; i < c; i += 256) alt[i] = file(list[i]);
        }
        __syncthreads();
        int a4[4], s = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { a4[u] = bc[4 * threadIdx.x + u]; s += a4[u]; }   // (both halves at once: neither passes 8192)
        int incl = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if ((int)(threadIdx.x & 63) >= off) incl += o;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        int run = incl - s;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += wsum[w];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int excl = run;
            run += a4[u];
            if ((excl & 0xFFFF) - (excl >> 16) < K && (run & 0xFFFF) - (run >> 16) >= K) { s_bk = 4 * threadIdx.x + u; s_tail0 = run & 0xFFFF; }
            bc[4 * threadIdx.x + u] = excl & 0xFFFF;
            bf[4 * threadIdx.x + u] = excl & 0xFFFF;
        }
        __syncthreads();
        const int bk = s_bk;
        if (bk >= 0) {
            const int tail0 = s_tail0;
            P = std::min<int64_t>(a.E, (int64_t)(bk + 1) * W);
            ct = c - tail0;
            if (in_regs) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int b = (int)((uint32_t)(mine[j] >> 32) / W);
                    if (j * 256 + (int)threadIdx.x < c && b > bk) tl[atomicAdd(&bf[b], 1) - tail0] = mine[j];
                }
                __syncthreads();
                for (int p = threadIdx.x; p < ct; p += 256) {
                    const uint64_t key = tl[p];
                    const int b = (int)((uint32_t)(key >> 32) / W);
                    const int s0 = bc[b] - tail0, s1 = (b + 1 < GH_CD_NB ? bc[b + 1] : c) - tail0;
                    int r = s0;
                    for (int q = s0; q < s1; ++q) r += tl[q] < key ? 1 : 0;
                    alt[r] = key;
                }
            } else {
                for (int i = threadIdx.x; i < c; i += 256) {
                    const uint64_t key = alt[i];
                    const int b = (int)((uint32_t)(key >> 32) / W);
                    if (b > bk) list[atomicAdd(&bf[b], 1) - tail0] = key;
                }
                __syncthreads();
                for (int p = threadIdx.x; p < ct; p += 256) {
                    const uint64_t key = list[p];
                    const int b = (int)((uint32_t)(key >> 32) / W);
                    const int s0 = bc[b] - tail0, s1 = (b + 1 < GH_CD_NB ? bc[b + 1] : c) - tail0;
                    int r = s0;
                    for (int q = s0; q < s1; ++q) r += list[q] < key ? 1 : 0;
                    alt[r] = key;
                }
            }
        }
    }
    if (threadIdx.x == 0) {
        const int slot = atomicAdd(&rr.hdr[0], 1);
        rr.rare[1 + slot] = (int32_t)qi;
        rr.P[slot] = (int32_t)P;
        rr.ct[slot] = ct;
        atomicMax(&rr.hdr[2], (int32_t)P);
        ovf[qi] = reason;
    }
}

// Row partitions: the ranks' K + 1 best cdist keys (and whether each rank could prove them its K + 1 best) -> the global
// rows.  A query whose merged K + 1 smallest values are pairwise different, every rank's list proven, is decided: ascending
// values (its k candidate pairs go through the intersection phase here).  Any other query is LISTED and partial_sort's heap
// is replayed -- on every rank alike, every rank holding all positions and the whole edge list, so no further collective
// is needed and all ranks get the same rows; over which ids, see the listed branch below.
template <int DT>
__global__ __launch_bounds__(256) void knn_merge_cdist_kernel(const uint64_t *__restrict__ gathered /* (world, S, K + 2) */, int world,
                                                              int64_t S, int K, int64_t E, uint64_t *__restrict__ merged, cdist_rows rr,
                                                              inter_args ia, uint64_t *__restrict__ cand /* listed rows: their tails */,
                                                              int bound /* a second n2 keys of LDS are there for the id order */) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint64_t *buf = reinterpret_cast<uint64_t *>(smem_raw);
    __shared__ gh_pair_list pairs;
    __shared__ uint32_t s_tstar;
    const int Ks = K + 1;
    const int64_t qi = blockIdx.x;
    const int total = world * Ks;
    const int n2 = next_pow2(total < 2 ? 2 : total);
    int bad = 0, unproven = 0;
    if (threadIdx.x == 0) s_tstar = 0xFFFFFFFFu;
    for (int i = threadIdx.x; i < n2; i += 256) {
        uint64_t key = GH_KEY_INF;
        if (i < total) {
            const int w = i / Ks, c = i % Ks;
            key = gathered[((int64_t)w * S + qi) * (Ks + 1) + c];
        }
        buf[i] = key;
    }
    for (int w = threadIdx.x; w < world; w += 256) unproven |= gathered[((int64_t)w * S + qi) * (Ks + 1) + Ks] == 1ull ? 0 : 1;
    __syncthreads();
    // t* = the smallest of the ranks' (K + 1)-th best values: an edge outside the gathered keys is no better than its own
    // rank's (K + 1)-th best, hence not below t* (a rank with fewer keys holds all its edges: its slot is the padding key)
    for (int w = threadIdx.x; w < world; w += 256) atomicMin(&s_tstar, (uint32_t)(buf[w * Ks + Ks - 1] >> 32));
    unproven = __syncthreads_or(unproven);
    const uint32_t tstar = s_tstar;
    block_sort(buf, n2);
    for (int i = threadIdx.x; i + 1 < Ks; i += 256) bad |= (uint32_t)(buf[i] >> 32) == (uint32_t)(buf[i + 1] >> 32) ? 1 : 0;
    if (__syncthreads_or(bad) || unproven) {
        // LISTED.  With every rank's keys proven, partial_sort's heap need not see all E values (the bound of the single-rank
        // replay, from the gathered keys alone): let C = the gathered keys with value <= t* (at least K + 1 of them) and P =
        // one past the K-th smallest ID in C.  Once the ids below P are through, the heap holds K values <= t*, so whatever
        // enters later has a value < t* -- below every rank's (K + 1)-th best, hence among the gathered keys.  The replay
        // therefore takes the ids below P in full and, behind them, the keys of C with id >= P in id order (the tail).  With
        // the ranks' best keys alike in value C holds about world * K keys and P is about E / world.
        int32_t P = (int32_t)E, ct = 0;
        if (!unproven && bound) {
            uint64_t *byid = buf + n2;
            int c = 0;   // (buf is sorted by (value, id): C is its first c keys)
            for (int i0 = 0; i0 < n2; i0 += 256) {
                const int i = i0 + (int)threadIdx.x;
                const uint64_t key = i < n2 ? buf[i] : GH_KEY_INF;
                const bool in = i < total && (uint32_t)(key >> 32) <= tstar;
                if (i < n2) byid[i] = in ? (key << 32) | (key >> 32) : GH_KEY_INF;   // (id << 32 | value bits): the tail's format
                c += __syncthreads_count(in ? 1 : 0);
            }
            block_sort(byid, n2);
            if (c >= K) {
                P = (int32_t)min((int64_t)(byid[K - 1] >> 32) + 1, E);
                ct = c - K;
                uint64_t *tl = cand + qi * GH_CAND_CAP + GH_CD_ALT;
                for (int i = threadIdx.x; i < ct; i += 256) tl[i] = byid[K + i];
            }
        }
        if (threadIdx.x == 0) {
            const int slot = atomicAdd(&rr.hdr[0], 1);
            rr.rare[1 + slot] = (int32_t)qi;
            rr.P[slot] = P;
            rr.ct[slot] = ct;
            atomicMax(&rr.hdr[2], P);
        }
        return;
    }
    for (int c = threadIdx.x; c < K; c += 256) merged[qi * K + c] = buf[c];
    if (ia.pos) intersect_query<DT>(ia, qi, buf, &pairs);
}

// ---- values of the listed rows' prefixes -----------------------------------------------------------------------------
// Workgroup x takes the ids [x * CH, (x + 1) * CH), CH = 256 * NPT (then x + gridDim.x, ... while below the longest
// prefix), gathers their midpoints once (registers) and loops over the rows of this round that reach that far
// (t = r0 + blockIdx.y, step gridDim.y): vbuf[slot][e] = value, cmin[slot][c] = the minimum over ids [64 c, 64 (c + 1)) --
// one wave's 64 consecutive ids.
template <int LDT, int NPT>
__global__ __launch_bounds__(256) void cdist_prefix_kernel(cdist_args a, cdist_rows rr, int all_rows, int r0, int R,
                                                           float *__restrict__ vbuf, int64_t vstride, float *__restrict__ cmin,
                                                           int nchunks) {
    constexpr int CH = 256 * NPT;
    constexpr int LM = LDT > 0 ? LDT : 1;
    constexpr int QW = LDT > 0 ? 16 : 1;
    // the records of this workgroup's rows (query coordinates, prefix length) come in first, all at once, before the
    // number of listed rows is known (slots exist for every t < S): per row they were two dependent round trips
    __shared__ float qsh[256][QW];
    __shared__ int64_t psh[256];
    __shared__ int64_t qish[256];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 4) rr.hdr_next[threadIdx.x] = 0;   // the next search's counters
    const int nrare = all_rows ? all_rows : rr.hdr[0];
    const int64_t Pmax = all_rows ? a.E : (int64_t)rr.hdr[2];
    const int t_end = min(nrare, r0 + R);
    if (r0 + (int)blockIdx.y >= t_end || (int64_t)blockIdx.x * CH >= Pmax) return;
    const int lane = threadIdx.x & 63;
    float m[NPT][LM], mn[NPT];
    auto midpoints = [&](int64_t e0) __attribute__((always_inline)) {   // midpoints of this thread's ids from e0 on, and their norms
        if constexpr (LDT > 0) {
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                const int64_t e = e0 + j * 256 + threadIdx.x;
                const int2 uv = e < a.E ? reinterpret_cast<const int2 *>(a.edges)[e] : make_int2(0, 0);
                float pu[LDT], pv[LDT];
                gh_load_row<LDT>(a.pos, uv.x, pu);
                gh_load_row<LDT>(a.pos, uv.y, pv);
                float s = 0.0f;
#pragma unroll
                for (int d = 0; d < LDT; ++d) {
                    m[j][d] = d < a.D ? (pu[d] + pv[d]) / 2.0f : 0.0f;
                    if (d < a.D) {
                        const float sq = m[j][d] * m[j][d];
                        s = s + sq;
                    }
                }
                mn[j] = s;
            }
        }
    };
    // the records of this workgroup's rows (query coordinates, prefix length) come in together, once (per row they were
    // two dependent round trips), while the first gathers are in flight
    const int gy = (int)gridDim.y;
    const int nrow = (t_end - r0 - (int)blockIdx.y + gy - 1) / gy;   // rows r0 + blockIdx.y + i * gy below t_end (<= 256: R <= 256 * gy)
    for (int i = threadIdx.x; i < nrow * QW; i += 256) {
        const int row = i / QW, d = i % QW;
        const int t = r0 + (int)blockIdx.y + row * gy;
        const int64_t qi = all_rows ? t : rr.rare[1 + t];
        if constexpr (LDT > 0) qsh[row][d] = d < a.D ? a.qt[qi * a.QS + d] : 0.0f;
        if (d == 0) { psh[row] = all_rows ? a.E : (int64_t)rr.P[t]; qish[row] = qi; }
    }
    midpoints((int64_t)blockIdx.x * CH);
    __syncthreads();
    for (int64_t e0 = (int64_t)blockIdx.x * CH; e0 < Pmax; e0 += (int64_t)gridDim.x * CH) {
        if (e0 != (int64_t)blockIdx.x * CH) midpoints(e0);
        for (int row = 0, t = r0 + (int)blockIdx.y; t < t_end; ++row, t += gy) {
            if (psh[row] <= e0) continue;   // (uniform over the workgroup)
            const int slot = t - r0;
            float vv[NPT];
            if constexpr (LDT > 0) {
                float q[LDT];
#pragma unroll
                for (int d = 0; d < LDT; ++d) q[d] = d < 16 ? qsh[row][d] : 0.0f;
                const float qn = cdist_norm(q, a.D);
#pragma unroll
                for (int j = 0; j < NPT; ++j) {
                    const int64_t e = e0 + j * 256 + threadIdx.x;
                    float acc = 0.0f;
#pragma unroll
                    for (int d = 0; d < LDT; ++d) {
                        if (d < a.D) {
                            if (a.mm_form) acc = fmaf(q[d] * -2.0f, m[j][d], acc);
                            else { const float df = q[d] - m[j][d]; acc = fmaf(df, df, acc); }
                        }
                    }
                    if (a.mm_form) { acc = acc + qn; acc = acc + mn[j]; }
                    vv[j] = e < a.E ? sqrtf(fmaxf(acc, 0.0f)) + 0.0f : INFINITY;
                }
            } else {   // any dimension: rows read from memory per pair
                const float *q = a.qt + qish[row] * a.QS;
                const float qn = cdist_norm(q, a.D);
#pragma unroll
                for (int j = 0; j < NPT; ++j) {
                    const int64_t e = e0 + j * 256 + threadIdx.x;
                    vv[j] = e < a.E ? cdist_pair<0>(a, q, qn, e) : INFINITY;
                }
            }
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                const int64_t e = e0 + j * 256 + threadIdx.x;
                vbuf[(int64_t)slot * vstride + e] = vv[j];
                float mv = vv[j];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) mv = fminf(mv, __shfl_xor(mv, off, 64));
                if (lane == 0) cmin[(int64_t)slot * nchunks + (e >> 6)] = mv;
            }
        }
    }
}

// Values are compared as UNSIGNED INTEGERS: a cdist value is +0 or larger (sqrt of a clamped sum, + 0), so the bit patterns
// order like the numbers; NaN gets one pattern above +inf.  That is exactly the comparator of ATen's topk on values ("x before
// y" iff x < y or only y is NaN) at one instruction per comparison, and it makes a heap element one 64-bit key
// (value bits << 32 | id) -- the layout of gh_key.
// (the predicate straight into a lane mask: HIP's __ballot goes through an integer compare of the widened predicate)
__device__ __forceinline__ unsigned long long gh_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t cdist_vkey(float v) { return v != v ? 0x7FC00000u : __float_as_uint(v); }
__device__ __forceinline__ uint32_t cdist_kv(uint64_t key) { return (uint32_t)(key >> 32); }

// Max-heap with the tie behaviour of libstdc++'s __adjust_heap, which std::partial_sort is built on: the hole at `top`
// sinks to the bottom along the larger child (the RIGHT one unless it is smaller than the left), then the new element climbs
// while its parent is smaller.  Three homes for the heap:
//   cdist_heap_scalar  K <= 16 (n_neighbors <= 15): the keys in SCALAR registers, every index a register name: a generated
//                   decision tree (cdist_heap_asm.h) that walks DOWN only -- the path of larger children is
//                   non-increasing, so the new element lands at the first child below it, which is where the climb
//                   would have brought it back to --, 12.6 scalar instructions and 2.2 taken branches per element
//                   on average.  One wave issues an instruction every ~4 cycles and pays ~28 for a taken branch, so
//                   instructions and branches ARE the time: ~0.07 us per element; the lane-parallel form below takes
//                   ~75 instructions (0.24 us).
//   cdist_heap_par  K <= 64: element i in lane i's registers; the whole adjustment is ONE data-parallel step.  Every
//                   lane looks at its two children (ds_bpermute) and knows which one the hole would move to; the path is
//                   followed through v_readlane (a few scalar steps); along the path the old values w_1 >= w_2 >= ... (heap
//                   order) end up as: w_(t+1) moved up into position t while it is not below the new value, the new element at
//                   the first position whose child is below it, the rest untouched -- the same array the sequential form
//                   leaves, since the climb undoes the sink below the new element's final position.  Also builds the heap
//                   (make_heap) and pops it (sort_heap) for the scalar form: those run once per row.
//   cdist_heap_lds  any K: keys in LDS, lane 0 walks (the sequential form as written in libstdc++).
struct cdist_heap_par {
    uint64_t key;
    int lane;
    __device__ __forceinline__ uint64_t at(int i) const { return readlane_u64(key, i); }
    __device__ __forceinline__ uint32_t val(int i) const { return (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), i); }
    __device__ __forceinline__ void set(int i, uint64_t x) { if (lane == i) key = x; }
    __device__ __forceinline__ void adjust(int top, int len, uint64_t x) {
        const int c1 = 2 * lane + 1, c2 = 2 * lane + 2;
        const int lo = (int)(uint32_t)key, hi = (int)(uint32_t)(key >> 32);
        const uint32_t h1 = (uint32_t)__builtin_amdgcn_ds_bpermute((c1 & 63) << 2, hi);
        const uint32_t h2 = (uint32_t)__builtin_amdgcn_ds_bpermute((c2 & 63) << 2, hi);
        const uint32_t l1 = (uint32_t)__builtin_amdgcn_ds_bpermute((c1 & 63) << 2, lo);
        const uint32_t l2 = (uint32_t)__builtin_amdgcn_ds_bpermute((c2 & 63) << 2, lo);
        const bool right = c2 < len && !(h2 < h1);
        const int c = right ? c2 : c1;          // where a hole at this lane moves to (lanes with c1 >= len: nowhere)
        const uint32_t ch = right ? h2 : h1, cl = right ? l2 : l1;
        unsigned long long path = 0ull;
        for (int p = top;;) {                   // (scalar: p comes from v_readlane)
            path |= 1ull << p;
            if (2 * p + 1 >= len) break;
            p = __builtin_amdgcn_readlane(c, p);
        }
        const uint32_t xv = cdist_kv(x);
        const bool on = (path >> lane) & 1ull;
        const bool up = on && c1 < len && !(ch < xv);                          // my child on the path moves up into me
        const bool mine = on && !up && (lane == top || !((uint32_t)hi < xv));  // the new element stops here
        if (up) key = ((uint64_t)ch << 32) | cl;
        else if (mine) key = x;
    }
    __device__ __forceinline__ void make_heap(int K) {
        for (int parent = (K - 2) / 2; parent >= 0; --parent) adjust(parent, K, at(parent));
    }
    __device__ __forceinline__ void sort_heap(int K) {   // pops the heap into ascending order
        for (int last = K - 1; last > 0; --last) {
            const uint64_t x = at(last);
            set(last, at(0));
            adjust(0, last, x);
        }
    }
};

#define GH_CD_KS 16
// (cdist_heap_asm.h, generated by tools/gen_heap_asm.py: element i pinned to the scalar register pair s[40 + 2 i : 41 + 2 i],
// id in the low half and value key in the high one, so that a move is one s_mov_b64 and a comparison one s_cmp_lt_u32 of
// the high halves.  The same tree written as C++ templates over two 16-element arrays compiled to a block that copied the
// heap between register sets at every merge: 526 cycles per entering element against ~25 instructions of real work.)
struct cdist_heap_scalar {
    uint64_t h[GH_CD_KS];
    __device__ __forceinline__ void load(const cdist_heap_par &hp) {
#pragma unroll
        for (int i = 0; i < GH_CD_KS; ++i) h[i] = readlane_u64(hp.key, i);
    }
    // (through LDS: a chain of selects on the lane id is turned into an indexed read of h[], which would put the heap in
    // scratch memory)
    __device__ __forceinline__ void store(cdist_heap_par &hp, uint64_t *lds /* GH_CD_KS */) const {
        if (hp.lane == 0) {
#pragma unroll
            for (int i = 0; i < GH_CD_KS; ++i) lds[i] = h[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wave: its LDS operations complete in order)
        hp.key = lds[hp.lane & (GH_CD_KS - 1)];
    }
    // std::__pop_heap's __adjust_heap(first, 0, K, x): the maximum leaves, x = (xv << 32 | id) sinks in from the root
    __device__ __forceinline__ void replace_max(int K, uint32_t xv, uint32_t xid) {
        const uint64_t x = ((uint64_t)xv << 32) | xid;
        asm(GH_CD_HEAP_REPLACE_ASM : GH_CD_HEAP_OPERANDS(h) : [x] "s"(x), [xv] "s"(xv), [len] "s"(K) : "scc");
    }
    __device__ __forceinline__ uint32_t top() const { return (uint32_t)(h[0] >> 32); }
};
struct cdist_heap_lds {
    uint64_t *hk;
    int lane;
    __device__ __forceinline__ uint64_t at(int i) const { return hk[i]; }
    __device__ __forceinline__ uint32_t val(int i) const { return cdist_kv(hk[i]); }
    // one wave works on the heap: its LDS operations complete in program order; the compiler must not reorder them
    __device__ __forceinline__ void sync() const { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    __device__ __forceinline__ void set(int i, uint64_t x) { if (lane == 0) hk[i] = x; sync(); }
    __device__ __forceinline__ void adjust(int hole, int len, uint64_t x) {
        const int top = hole;
        int child = hole;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            if (val(child) < val(child - 1)) --child;
            set(hole, at(child));
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            set(hole, at(child - 1));
            hole = child - 1;
        }
        int parent = (hole - 1) / 2;
        while (hole > top && val(parent) < cdist_kv(x)) {
            set(hole, at(parent));
            hole = parent;
            parent = (hole - 1) / 2;
        }
        set(hole, x);
    }
    __device__ __forceinline__ void make_heap(int K) {
        for (int parent = (K - 2) / 2; parent >= 0; --parent) adjust(parent, K, at(parent));
    }
    __device__ __forceinline__ void sort_heap(int K) {
        for (int last = K - 1; last > 0; --last) {
            const uint64_t x = at(last);
            set(last, at(0));
            adjust(0, last, x);
        }
    }
};

// One workgroup per listed row: std::partial_sort(first, first + K, last) over the (value, index) pairs in index order.
// The first K pairs are heapified; then pair i enters (replacing the maximum) iff value_i < maximum -- so a chunk whose
// minimum is not below the current maximum holds nothing that would enter and is skipped -- over the valued prefix
// [0, P), then over the id-sorted tail of the candidate list; finally the heap is popped into ascending order.  Wave 0 owns
// the heap (its steps are uniform over the wave; the lanes pre-test 64 values at a time); all four waves fetch.
// HEAP: 0 = scalar registers (K <= 16), 1 = lanes (K <= 64), 2 = LDS.
// nth_form (K * 64 > E): ATen ranks with std::nth_element + std::sort instead; the K smallest values are the same,
// equal values are put in id order here and the row is counted in hdr[1] when any tie is present.
// INTER: the row's k candidate pairs go through the intersection phase (pt.py:638-774) right here.
template <int DT, int HEAP, bool INTER>
__global__ __launch_bounds__(256) void cdist_replay_kernel(cdist_rows rr, int all_rows, int r0, int R, int64_t E, int K,
                                                          const float *__restrict__ vbuf, int64_t vstride,
                                                          const float *__restrict__ cmin, int nchunks,
                                                          const uint64_t *__restrict__ cand, uint64_t *__restrict__ out_keys,
                                                          int nth_arg, inter_args ia,
                                                          unsigned long long *__restrict__ stamps /* diagnostic (GRAPHEM_HIP_STAMPS), or null */,
                                                          int S_slots /* slot records that exist (S) */, int stamp_slots /* slots the stamp buffer holds */) {
    extern __shared__ __align__(16) unsigned char smem_raw[];   // the LDS heap (HEAP == 2)
    __shared__ float cml[GH_CD_TILE];
    __shared__ uint32_t stage[GH_CD_BATCH * 64];   // value KEYS of the fetched chunks; 0xFFFFFFFF where the id is out of [K, P)
    __shared__ int live_ids[GH_CD_BATCH];
    __shared__ int s_nlive, s_pnext, s_cnt[4];
    __shared__ uint32_t s_hmax;
    __shared__ uint64_t tail_lds[GH_CD_TAIL_LDS];
    // (the tiny-graph bookkeeping stays out of the scalar-register form, whose loop is counted in instructions: the host
    // sends such rows to the lane form)
    const bool nth_form = HEAP != 0 && nth_arg != 0;
    const int t = r0 + (int)blockIdx.x;
    if (t >= r0 + R || t >= S_slots) return;   // (the grid always has cd_R workgroups: the last round's reach past the S slot records)
    // (slot records are read before the count is known: they exist for every t < S)
    const int nrare = all_rows ? all_rows : rr.hdr[0];
    const int64_t qi = all_rows ? t : rr.rare[1 + t];
    const int64_t P = all_rows ? E : (int64_t)rr.P[t];
    const int ct = all_rows ? 0 : rr.ct[t];
    if (t >= nrare) return;
    const float *v = vbuf + (int64_t)(t - r0) * vstride;
    const float *cm = cmin + (int64_t)(t - r0) * nchunks;
    const uint64_t *tl = cand + qi * GH_CAND_CAP + GH_CD_ALT;
    const int lane = threadIdx.x & 63;
    // (a scalar condition: what wave 0 keeps across the kernel -- the heap above all -- then lives in scalar registers)
    const bool wave0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0;
    // diagnostic: 16 records per slot -- wall clock (100 MHz) at start, heap built, prefix done, tail done, sorted, end;
    // then batches, elements entered, chunks processed, P, tail length
    // (the buffer is the fused launch's, shared with tools/stamp_probe.py: (n_vblocks + GH_STAMP_EXTRA) * 8 words -- a slot
    // beyond it records nothing)
    if (stamps) stamps = t < stamp_slots ? stamps + (int64_t)t * 32 : nullptr;
    int n_batches = 0, n_entered = 0, n_chunks = 0;
#define GH_CD_STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[k] = wall_clock64(); } while (0)
    GH_CD_STAMP(0);
    // ---- everything the first stretch needs is requested at once, ahead of the heap build: the K values of the heap, the
    // chunk minima of the first tile, the tail of the candidate list, and the FIRST BATCH of the valued prefix -- the chunks
    // right behind the heap's own values, which are live under the first maximum whatever it is, so they are fetched
    // without being listed (listing + fetching them behind the heap build was 3.5 us of every row)
    const int nch = (int)((P + 63) >> 6);
    const int P32 = (int)P;
    const int c_first = K >> 6;
    const int tile_first = c_first / GH_CD_TILE * GH_CD_TILE;
    int pre = max(0, min(min(nch, tile_first + GH_CD_TILE) - c_first, GH_CD_BATCH));
    constexpr int PT = GH_CD_BATCH * 64 / 256;   // values per thread of a full batch
    using wide_t = typename std::conditional<HEAP == 2, cdist_heap_lds, cdist_heap_par>::type;
    wide_t hw;
    cdist_heap_scalar hs;
    hw.lane = lane;
    uint32_t hmax = 0xFFFFFFFFu;
    uint32_t eq_out = 0xFFFFFFFFu;   // nth_form: a value left outside the heap while equal to its maximum (boundary tie if it stays so)
    {
        float h0 = 0.0f;
        if constexpr (HEAP != 2) h0 = wave0 && lane < K ? v[lane] : 0.0f;
        float tmc[GH_CD_TILE / 256], tps[PT];
        const int nt0 = min(GH_CD_TILE, nch - tile_first);
#pragma unroll
        for (int u = 0; u < GH_CD_TILE / 256; ++u) tmc[u] = u * 256 + (int)threadIdx.x < nt0 ? cm[tile_first + u * 256 + threadIdx.x] : INFINITY;
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int i = u * 256 + (int)threadIdx.x;
            tps[u] = i < pre * 64 ? v[((int64_t)(c_first + (i >> 6)) << 6) + (i & 63)] : 0.0f;
        }
        if (!wave0)
            for (int i = (int)threadIdx.x - 64; i < min(ct, GH_CD_TAIL_LDS); i += 192) tail_lds[i] = tl[i];
        // the heap: built in the lane form (or in LDS); the scalar form takes it over for the replay
        if (wave0) {
            if constexpr (HEAP == 2) {
                hw.hk = reinterpret_cast<uint64_t *>(smem_raw);
                for (int i = lane; i < K; i += 64) hw.hk[i] = ((uint64_t)cdist_vkey(v[i]) << 32) | (uint32_t)i;
                hw.sync();
            } else {
                hw.key = ((uint64_t)(lane < K ? cdist_vkey(h0) : 0xFFFFFFFFu) << 32) | (uint32_t)lane;
            }
            if (K >= 2) hw.make_heap(K);
            if constexpr (HEAP == 0) hs.load(hw);
            hmax = hw.val(0);
            if (lane == 0) s_hmax = hmax;
        }
#pragma unroll
        for (int u = 0; u < GH_CD_TILE / 256; ++u) cml[u * 256 + threadIdx.x] = tmc[u];
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int id = ((c_first + u * 4 + (int)(threadIdx.x >> 6)) << 6) + lane;   // (i >> 6 = u * 4 + wave, i & 63 = lane)
            stage[u * 256 + threadIdx.x] = id >= K && id < P32 ? cdist_vkey(tps[u]) : 0xFFFFFFFFu;
        }
        if ((int)threadIdx.x < pre) live_ids[threadIdx.x] = c_first - tile_first + (int)threadIdx.x;
    }
    GH_CD_STAMP(1);
    // "may hold an element that enters": minimum below the maximum (nth_form: or equal to it, for the boundary-tie count)
    auto live_key = [&](uint32_t mk) { return nth_form ? !(hmax < mk) : mk < hmax; };
    // 64 pairs (lane l: value key x, id xid, `in` = to be considered): in lane (= index) order, whatever still beats the maximum
    // enters.  The loop is counted in instructions (one wave issues one every ~4 cycles, a taken branch costs ~28): lanes
    // out of consideration get the key 0xFFFFFFFF, above every value's, so that the lanes still in the running are ONE
    // v_cmp against the maximum and one s_and with the lanes not yet through (the maximum only falls: a lane that failed
    // once fails for good, so only the lanes that entered have to be struck out).
    auto process = [&](uint32_t x, int32_t xid, bool in) __attribute__((always_inline)) {
        if (nth_form && gh_ballot(in && x == hmax)) eq_out = hmax;
        const uint32_t xk = in ? x : 0xFFFFFFFFu;
        unsigned long long todo = ~0ull;
        unsigned long long mask = gh_ballot(xk < hmax);
        while (mask) {   // every lane of the mask holds a value below the maximum of this moment
            const int l = __builtin_ctzll(mask);
            const uint32_t ev = (uint32_t)__builtin_amdgcn_readlane((int)xk, l), eid = (uint32_t)__builtin_amdgcn_readlane(xid, l);
            // std::__pop_heap(first, middle, i): the maximum leaves, the new element sinks in from the root
            const uint32_t old = hmax;
            if constexpr (HEAP == 0) { hs.replace_max(K, ev, eid); hmax = hs.top(); }
            else { hw.adjust(0, K, ((uint64_t)ev << 32) | eid); hmax = hw.val(0); }
            ++n_entered;
            if (nth_form && old == hmax) eq_out = hmax;   // one of several equal maxima was popped: it now waits outside
            todo &= ~(1ull << l);
            mask = gh_ballot(xk < hmax) & todo;
            if (nth_form && (gh_ballot(in && x == hmax) & (l == 63 ? 0ull : ~((2ull << l) - 1ull)))) eq_out = hmax;
        }
    };
    // ---- the valued prefix: chunk minima through LDS, GH_CD_TILE at a time; the chunks that are live under the maximum of the
    // moment are listed (up to GH_CD_BATCH), fetched by all four waves into LDS -- every thread's loads issued together: a
    // load per loop trip was a memory round trip per trip, most of this kernel's first version -- and processed in order,
    // each re-tested (64 minima per ballot) against the maximum of its moment
    for (int tile0 = tile_first; tile0 < nch; tile0 += GH_CD_TILE) {
        const int nt = min(GH_CD_TILE, nch - tile0);
        if (tile0 != tile_first) {
            __syncthreads();
            float tmp[GH_CD_TILE / 256];
#pragma unroll
            for (int u = 0; u < GH_CD_TILE / 256; ++u) tmp[u] = u * 256 + (int)threadIdx.x < nt ? cm[tile0 + u * 256 + threadIdx.x] : INFINITY;
#pragma unroll
            for (int u = 0; u < GH_CD_TILE / 256; ++u) cml[u * 256 + threadIdx.x] = tmp[u];
        }
        __syncthreads();
        int p = max(0, c_first - tile0);   // next chunk of the tile to look at (wave 0)
        for (;;) {
            int nl;
            if (pre) {   // the batch fetched ahead (first tile only)
                nl = pre;
                p += pre;
                pre = 0;
                if (stamps && threadIdx.x == 0 && n_batches < 4) stamps[16 + 4 * n_batches] = wall_clock64();
                ++n_batches;
            } else {
                // the chunks that are live under the maximum of the moment (published by wave 0), listed in order by all four
                // waves: each counts the live chunks of its quarter of [p, nt), then writes them behind the quarters before
                // it; the GH_CD_BATCH-th one listed says where the next batch starts
                {
                    const uint32_t hm = s_hmax;
                    const int wv = (int)(threadIdx.x >> 6);
                    const int g0 = p >> 6, per = (((nt + 63) >> 6) - g0 + 3) >> 2;
                    const int ga = g0 + wv * per, gb = min((nt + 63) >> 6, ga + per);
                    auto group = [&](int g) {
                        const int c = (g << 6) + lane;
                        const uint32_t mk = c >= p && c < nt ? cdist_vkey(cml[c]) : 0xFFFFFFFFu;
                        return gh_ballot(nth_form ? !(hm < mk) : mk < hm);
                    };
                    int cnt = 0;
                    for (int g = ga; g < gb; ++g) cnt += __builtin_popcountll(group(g));
                    if (lane == 0) s_cnt[wv] = cnt;
                    __syncthreads();
                    int off = 0, tot = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) { const int cw = s_cnt[w]; off += w < wv ? cw : 0; tot += cw; }
                    for (int g = ga; g < gb && off < GH_CD_BATCH; ++g) {
                        const unsigned long long m = group(g);
                        const int rank = off + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                        if (((m >> lane) & 1ull) && rank < GH_CD_BATCH) {
                            live_ids[rank] = (g << 6) + lane;
                            if (rank == GH_CD_BATCH - 1) s_pnext = (g << 6) + lane + 1;
                        }
                        off += __builtin_popcountll(m);
                    }
                    if (threadIdx.x == 0) {
                        s_nlive = min(tot, GH_CD_BATCH);
                        if (tot < GH_CD_BATCH) s_pnext = nt;
                    }
                }
                __syncthreads();
                nl = s_nlive;
                p = s_pnext;
                if (nl == 0) break;
                if (stamps && threadIdx.x == 0 && n_batches < 4) stamps[16 + 4 * n_batches] = wall_clock64();
                ++n_batches;
                {
                    float tmp[PT];
                    int tid0[PT];
#pragma unroll
                    for (int u = 0; u < PT; ++u) {
                        const int i = u * 256 + (int)threadIdx.x;
                        tid0[u] = i < nl * 64 ? ((tile0 + live_ids[i >> 6]) << 6) + (i & 63) : -1;
                        tmp[u] = tid0[u] >= 0 ? v[tid0[u]] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < PT; ++u) stage[u * 256 + threadIdx.x] = tid0[u] >= K && tid0[u] < P32 ? cdist_vkey(tmp[u]) : 0xFFFFFFFFu;
                }
                __syncthreads();
            }
            if (stamps && threadIdx.x == 0 && n_batches <= 4) stamps[16 + 4 * (n_batches - 1) + 1] = wall_clock64();
            if (wave0) {
                // (the ids and minima of 64 listed chunks in registers; the values of the next live chunk are requested from
                // LDS before the current one is processed: two dependent LDS round trips per chunk were a third of its cost)
                for (int rb = 0; rb < nl; rb += 64) {
                    const int my_c = rb + lane < nl ? live_ids[rb + lane] : 0;
                    const uint32_t mk = rb + lane < nl ? cdist_vkey(cml[my_c]) : 0xFFFFFFFFu;   // (never live)
                    unsigned long long m = gh_ballot(live_key(mk));
                    int r = m ? __builtin_ctzll(m) : 0;
                    uint32_t xn = stage[(rb + r) * 64 + lane];
                    while (m) {
                        ++n_chunks;
                        const uint32_t x = xn;
                        const int base = (tile0 + __builtin_amdgcn_readlane(my_c, r)) << 6;   // (edge ids are 32-bit)
                        const unsigned long long later = r == 63 ? 0ull : ~((2ull << r) - 1ull);
                        const int r2 = (m & later) ? __builtin_ctzll(m & later) : r;   // the next one as things stand
                        xn = stage[(rb + r2) * 64 + lane];
                        process(x, base + lane, true);
                        m = gh_ballot(live_key(mk)) & later;
                        if (!m) break;
                        r = __builtin_ctzll(m);
                        if (r != r2) xn = stage[(rb + r) * 64 + lane];   // the maximum fell below that chunk's minimum meanwhile
                    }
                }
            }
            if (wave0 && lane == 0) s_hmax = hmax;
            if (stamps && threadIdx.x == 0 && n_batches <= 4) { stamps[16 + 4 * (n_batches - 1) + 2] = wall_clock64(); stamps[16 + 4 * (n_batches - 1) + 3] = nl; }
            __syncthreads();
        }
    }
    GH_CD_STAMP(2);
    // ---- the tail of the candidate list, sorted by id: (id << 32 | value bits)
    if (wave0 && ct > 0) {
        uint64_t nxt = lane < ct ? tail_lds[lane] : 0ull;   // (ct > GH_CD_TAIL_LDS: the rest straight from memory)
        for (int b0 = 0; b0 < ct; b0 += 64) {
            const uint64_t tk = nxt;
            const int i = b0 + 64 + lane;
            nxt = i >= ct ? 0ull : i < GH_CD_TAIL_LDS ? tail_lds[i] : tl[i];
            process(cdist_vkey(__uint_as_float((uint32_t)tk)), (int32_t)(tk >> 32), b0 + lane < ct);
        }
    }
    GH_CD_STAMP(3);
    __shared__ uint64_t best[GH_EXTRACT_MAX_K];
    static_assert(GH_EXTRACT_MAX_K >= GH_CD_KS, "best[] carries the scalar heap back to the lanes");
    if (wave0) {
        if constexpr (HEAP == 0) hs.store(hw, best);
        hw.sort_heap(K);   // std::sort_heap: ascending
        if (nth_form) {
            bool tie = eq_out == hw.val(K - 1);
            for (int i = 0; i + 1 < K; ++i) tie = tie || hw.val(i) == hw.val(i + 1);
            if (tie) {
                if (lane == 0) atomicAdd(&rr.hdr[1], 1);
                for (int i = 1; i < K; ++i) {   // equal values in id order (insertion sort within runs)
                    const uint64_t x = hw.at(i);
                    int j = i;
                    while (j > 0 && hw.val(j - 1) == cdist_kv(x) && (uint32_t)hw.at(j - 1) > (uint32_t)x) { hw.set(j, hw.at(j - 1)); --j; }
                    hw.set(j, x);
                }
            }
        }
        if constexpr (HEAP == 2) {
            for (int i = lane; i < K; i += 64) out_keys[qi * K + i] = hw.hk[i];
        } else {
            if (lane < K) {
                out_keys[qi * K + lane] = hw.key;
                if (INTER) best[lane] = hw.key;
            }
        }
    }
    GH_CD_STAMP(4);
    if constexpr (INTER) {
        static_assert(HEAP != 2, "the fused intersection phase reads the keys of a register heap");
        __shared__ gh_pair_list pairs;
        __syncthreads();
        if (ia.pos) intersect_query<DT>(ia, qi, best, &pairs);
    }
    GH_CD_STAMP(5);
    if (stamps && threadIdx.x == 0) {
        stamps[6] = n_batches; stamps[7] = n_entered; stamps[8] = n_chunks; stamps[9] = (unsigned long long)P; stamps[10] = ct;
    }
#undef GH_CD_STAMP
}

// ---- tiny graphs: K * 64 > E -------------------------------------------------------------------------------------------
// There ATen's topk (ATen/native/TopKImpl.h topk_impl_loop) ranks with std::nth_element(first, first + K - 1, last) followed by
// std::sort(first, first + K - 1): equal values come out in the order libstdc++'s introselect and introsort leave them.  One
// wave per row replays both on the row's (value, index) pairs in LDS -- the algorithms as written in bits/stl_algo.h
// (__introselect, __unguarded_partition_pivot, __move_median_to_first, __unguarded_partition, __insertion_sort,
// __introsort_loop, __final_insertion_sort), control flow uniform over the wave, lane 0 storing, all lanes reading; the two
// scans of a partition step look at 64 elements per ballot.  E <= GH_CD_NTH_MAX (64 KB of LDS); beyond that -- K > 125 on a
// graph of a few thousand edges -- the heap replay above ranks the row and counts it when a tie is present.
// A depth limit that runs out (libstdc++ then switches to heap select / heap sort; adversarial inputs only) is not followed:
// the partitioning simply goes on and the row is counted in hdr[1].
#define GH_CD_NTH_MAX 8000
struct cdist_nth {
    uint64_t *a;   // LDS: (value key << 32 | index)
    int n, lane;
    bool gave_up;
    __device__ __forceinline__ void fence() const { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    __device__ __forceinline__ uint64_t ld(int i) const { return a[i]; }
    __device__ __forceinline__ uint32_t v(int i) const { return (uint32_t)(a[i] >> 32); }
    __device__ __forceinline__ void st(int i, uint64_t x) { if (lane == 0) a[i] = x; fence(); }
    __device__ __forceinline__ void swap(int i, int j) { const uint64_t x = ld(i), y = ld(j); st(i, y); st(j, x); }
    __device__ __forceinline__ static int lg(int x) { return 31 - __builtin_clz(x); }
    // std::__move_median_to_first(result, a, b, c)
    __device__ __forceinline__ void median_to_first(int result, int ia, int ib, int ic) {
        const uint32_t va = v(ia), vb = v(ib), vc = v(ic);
        if (va < vb) {
            if (vb < vc) swap(result, ib);
            else if (va < vc) swap(result, ic);
            else swap(result, ia);
        } else if (va < vc) swap(result, ia);
        else if (vb < vc) swap(result, ic);
        else swap(result, ib);
    }
    // std::__unguarded_partition(first, last, pivot): the scans stop at elements not below / not above the pivot, which
    // exist inside [first - 1, last] by construction (the median sits at first - 1, an element not below it at or before last)
    __device__ __forceinline__ int partition(int first, int last, int pivot) {
        const uint32_t pv = v(pivot);
        for (;;) {
            for (;;) {   // while (comp(first, pivot)) ++first;
                const int i = first + lane;
                const unsigned long long m = __ballot(i >= n || !(v(i < n ? i : n - 1) < pv));
                if (m) { first += __builtin_ctzll(m); break; }
                first += 64;
            }
            --last;
            for (;;) {   // while (comp(pivot, last)) --last;
                const int i = last - lane;
                const unsigned long long m = __ballot(i < 0 || !(pv < v(i >= 0 ? i : 0)));
                if (m) { last -= __builtin_ctzll(m); break; }
                last -= 64;
            }
            if (!(first < last)) return first;
            swap(first, last);
            ++first;
        }
    }
    __device__ __forceinline__ int partition_pivot(int first, int last) {   // std::__unguarded_partition_pivot
        const int mid = first + (last - first) / 2;
        median_to_first(first, first + 1, mid, last - 1);
        return partition(first + 1, last, first);
    }
    __device__ __forceinline__ void linear_insert(int last) {   // std::__unguarded_linear_insert
        const uint64_t val = ld(last);
        int next = last - 1;
        while ((uint32_t)(val >> 32) < v(next)) { st(last, ld(next)); last = next; --next; }
        st(last, val);
    }
    __device__ __forceinline__ void insertion_sort(int first, int last) {   // std::__insertion_sort
        if (first == last) return;
        for (int i = first + 1; i != last; ++i) {
            if (v(i) < v(first)) {
                const uint64_t val = ld(i);
                for (int j = i; j > first; --j) st(j, ld(j - 1));   // std::move_backward(first, i, i + 1)
                st(first, val);
            } else {
                linear_insert(i);
            }
        }
    }
    __device__ __forceinline__ void nth_element(int nth) {   // std::nth_element(0, nth, n): __introselect
        if (n == 0 || nth == n) return;
        int first = 0, last = n, depth = lg(n) * 2;
        while (last - first > 3) {
            if (depth == 0) gave_up = true; else --depth;
            const int cut = partition_pivot(first, last);
            if (cut <= nth) first = cut; else last = cut;
        }
        insertion_sort(first, last);
    }
    __device__ __forceinline__ void sort(int cnt) {   // std::sort(0, cnt): __introsort_loop + __final_insertion_sort
        if (cnt <= 0) return;
        // the recursion of __introsort_loop as an explicit stack: the two sides of a cut are disjoint ranges, sorted
        // independently of each other, so the order they are taken in does not matter
        int sf[48], sl[48], sd[48], sp = 0;
        sf[0] = 0; sl[0] = cnt; sd[0] = lg(cnt) * 2; sp = 1;
        while (sp > 0) {
            --sp;
            int first = sf[sp], last = sl[sp], depth = sd[sp];
            while (last - first > 16) {
                if (depth == 0) gave_up = true; else --depth;
                const int cut = partition_pivot(first, last);
                if (sp < 48) { sf[sp] = cut; sl[sp] = last; sd[sp] = depth; ++sp; } else gave_up = true;
                last = cut;
            }
        }
        if (cnt > 16) {
            insertion_sort(0, 16);
            for (int i = 16; i != cnt; ++i) linear_insert(i);   // std::__unguarded_insertion_sort
        } else {
            insertion_sort(0, cnt);
        }
    }
};
__global__ __launch_bounds__(64) void cdist_nth_kernel(int S, int r0, int R, int64_t E, int K, const float *__restrict__ vbuf, int64_t vstride,
                                                       uint64_t *__restrict__ out_keys, int32_t *__restrict__ hdr) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int t = r0 + (int)blockIdx.x;
    if (t >= min(S, r0 + R)) return;
    const float *v = vbuf + (int64_t)(t - r0) * vstride;
    cdist_nth q;
    q.a = reinterpret_cast<uint64_t *>(smem_raw);
    q.n = (int)E;
    q.lane = threadIdx.x;
    q.gave_up = false;
    for (int i = threadIdx.x; i < q.n; i += 64) q.a[i] = ((uint64_t)cdist_vkey(v[i]) << 32) | (uint32_t)i;
    q.fence();
    q.nth_element(K - 1);
    q.sort(K - 1);
    for (int i = threadIdx.x; i < K; i += 64) out_keys[(int64_t)t * K + i] = q.a[i];
    if (q.gave_up && threadIdx.x == 0) atomicAdd(&hdr[1], 1);
}

cdist_args make_cdist_args(gh_engine *h) {
    return cdist_args{h->d_pos, h->d_edges, h->D, h->LD, h->E, h->d_q, gh_qs(h->D, h->LD), gh_qtau(h->D, h->LD),
                      (h->S > 25 || h->E > 25) ? 1 : 0};
}

template <typename T>
gh_status cdist_dev_alloc(gh_engine *h, T **p, size_t count) {
    if (hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) {
        *p = nullptr;
        h->err = "hipMalloc failed (GH_DIST_CDIST buffers: up to min(sample_size, 2^30 / E) rows of E floats; sample_size = " +
                 std::to_string(h->S) + ", E = " + std::to_string(h->E) + ")";
        return GH_ERR_NOMEM;
    }
    return GH_OK;
}

int64_t cdist_vstride(const gh_engine *h) { return (h->E + 1023) / 1024 * 1024; }   // whole workgroups of cdist_prefix_kernel

}  // namespace

// Buffers of the replay: values of up to cd_R listed rows (at most 4 GiB if every one of them needed all E values; a
// listed row usually writes E / stride of its slot; more listed rows than cd_R take further rounds of the two kernels --
// every round is launched, an empty one returns at once).
gh_status gh_cdist_alloc(gh_engine *h) {
    if (!h->cdist || h->S == 0 || h->k == 0) return GH_OK;
    const int64_t vstride = cdist_vstride(h);
    h->cd_nchunks = (int)(vstride / 64);
    int64_t R = ((int64_t)1 << 30) / std::max<int64_t>(vstride, 1);
    if (R < 16) R = 16;
    if (R > h->S) R = h->S;
    h->cd_R = (int)R;
    GH_TRY_ST(cdist_dev_alloc(h, &h->d_rare, (size_t)h->S + 1));
    GH_TRY_ST(cdist_dev_alloc(h, &h->d_cd_rows, 2 * (size_t)h->S));
    GH_TRY_ST(cdist_dev_alloc(h, &h->d_cd_vbuf, (size_t)(R * vstride)));
    GH_TRY_ST(cdist_dev_alloc(h, &h->d_cd_cmin, (size_t)(R * h->cd_nchunks)));
    GH_TRY_ST(cdist_dev_alloc(h, &h->d_cd_stat, 8));
    GH_HIP(hipMemsetAsync(h->d_rare, 0, sizeof(int32_t) * ((size_t)h->S + 1), h->stream));
    GH_HIP(hipMemsetAsync(h->d_cd_rows, 0, sizeof(int32_t) * 2 * (size_t)h->S, h->stream));
    GH_HIP(hipMemsetAsync(h->d_cd_stat, 0, sizeof(int32_t) * 8, h->stream));
    return GH_OK;
}

// The replay of the listed rows (or of all S queries: all_rows), cd_R rows per round: values of their prefixes, then one
// workgroup per row -> out_keys (S, K).  fuse: the rows' intersection phase rides in the replay launch.
static gh_status cdist_replay_rounds(gh_engine *h, const cdist_args &a, const cdist_rows &rr, bool all_rows, bool fuse, uint64_t *out_keys) {
    const int64_t vstride = cdist_vstride(h);
    const int nth_form = (int64_t)h->K * 64 > h->E ? 1 : 0;
    const int all = all_rows ? (int)h->S : 0;
    // ids per thread of cdist_prefix_kernel: four where the prefix is all edges or E / world of them (more gathers in flight per
    // thread); one where it is the sliver a single engine's candidate list leaves (~E / stride ids: spread over as many
    // workgroups as there are, each thread's chain one id long)
    const int npt = h->E >= (1 << 16) && (all_rows || h->cd_part) ? 4 : 1;
    const unsigned nwg = (unsigned)((h->E + 256 * npt - 1) / (256 * npt));
    const unsigned gx = std::min(nwg, 1024u);
    // rows of a round are spread over gridDim.y so that the chip is filled, and so that a workgroup takes at most 256 of them
    const unsigned ys = (unsigned)std::max({1, std::min(h->cd_R, 1024 / (int)std::max(gx, 1u)), (h->cd_R + 255) / 256});
    for (int r0 = 0; r0 < (int)h->S; r0 += h->cd_R) {
        {
            gh_scope t(h, "cdist_prefix");
            const dim3 grid(gx, ys);
#define GH_CPRE(LL, NP) cdist_prefix_kernel<LL, NP><<<grid, dim3(256), 0, h->stream>>>(a, rr, all, r0, h->cd_R, h->d_cd_vbuf, vstride, \
                                                                                       h->d_cd_cmin, h->cd_nchunks)
            if (npt == 4) {
                if (h->LD == 4) GH_CPRE(4, 4); else if (h->LD == 8) GH_CPRE(8, 4); else if (h->LD == 16) GH_CPRE(16, 4); else GH_CPRE(0, 4);
            } else {
                if (h->LD == 4) GH_CPRE(4, 1); else if (h->LD == 8) GH_CPRE(8, 1); else if (h->LD == 16) GH_CPRE(16, 1); else GH_CPRE(0, 1);
            }
#undef GH_CPRE
            GH_LAUNCH_CHECK();
        }
        if (nth_form && all_rows && h->E <= GH_CD_NTH_MAX) {   // tiny graphs: ATen's nth_element + sort, replayed
            gh_scope t(h, "cdist_nth");
            cdist_nth_kernel<<<dim3((unsigned)h->cd_R), dim3(64), sizeof(uint64_t) * (size_t)h->E, h->stream>>>(
                (int)h->S, r0, h->cd_R, h->E, h->K, h->d_cd_vbuf, vstride, out_keys, rr.hdr);
            GH_LAUNCH_CHECK();
            continue;
        }
        gh_scope t(h, fuse ? "cdist_replay_intersect" : "cdist_replay");
        const size_t smem = h->K <= 64 ? 0 : sizeof(uint64_t) * (size_t)h->K;
        const inter_args ia = make_inter_args(h, fuse);
        const bool scalar_heap = h->K <= GH_CD_KS && !nth_form;
#define GH_CREP(DD, HEAPv, INTv) cdist_replay_kernel<DD, HEAPv, INTv><<<dim3((unsigned)h->cd_R), dim3(256), smem, h->stream>>>( \
        rr, all, r0, h->cd_R, h->E, h->K, h->d_cd_vbuf, vstride, h->d_cd_cmin, h->cd_nchunks, h->d_cand, out_keys, nth_form, ia, h->d_stamps, \
        (int)h->S, (int)(((int64_t)h->n_vblocks + GH_STAMP_EXTRA) * 8 / 32))
#define GH_CREP_D(DD) case DD: if (scalar_heap) GH_CREP(DD, 0, true); else GH_CREP(DD, 1, true); break;
        if (fuse) {
            switch (h->D) {
                GH_CREP_D(2) GH_CREP_D(3) GH_CREP_D(4) GH_CREP_D(5) GH_CREP_D(6) GH_CREP_D(7) GH_CREP_D(8) GH_CREP_D(9)
                GH_CREP_D(10) GH_CREP_D(11) GH_CREP_D(12) GH_CREP_D(13) GH_CREP_D(14) GH_CREP_D(15)
                default: if (scalar_heap) GH_CREP(16, 0, true); else GH_CREP(16, 1, true); break;
            }
        } else if (scalar_heap) {
            GH_CREP(0, 0, false);
        } else if (h->K <= 64) {
            GH_CREP(0, 1, false);
        } else {
            GH_CREP(0, 2, false);
        }
#undef GH_CREP_D
#undef GH_CREP
        GH_LAUNCH_CHECK();
    }
    return GH_OK;
}

// all_rows: the graph is too small for the filtered scan -- every query is replayed over all edges.  Otherwise the
// candidate lists of the scan are in place.  -> d_partial (S, K): the reference's rows, column 0 included.
// fuse_intersect (single-rank steps, K <= 64): the same launches run the intersection phase of every query they finish
// and reduce the fused kernel's column sums (h->intersect_done / h->stats_reduced tell the caller).
// A row-partitioned engine (h->cd_part) stops after its own candidates: -> d_partial (S, K + 2): this rank's K + 1 best cdist
// keys of every query and 1 where they are provably its K + 1 best; gh_knn_merge_cdist finishes after the ranks' all-gather.
gh_status gh_knn_finish_cdist(gh_engine *h, bool all_rows, bool fuse_intersect) {
    const cdist_args a = make_cdist_args(h);
    if (h->cd_part && all_rows) {   // too few own edges for the scan: nothing proven, every row is replayed after the merge
        GH_HIP(hipMemsetAsync(h->d_partial, 0xFF, sizeof(uint64_t) * (size_t)h->S * (h->K + 2), h->stream));
        h->intersect_done = false;
        h->stats_reduced = false;
        return GH_OK;
    }
    const int set = h->cd_part ? h->cd_set ^ 1 : h->cd_set;   // (a partitioned engine's search takes its counter set at the merge)
    if (!h->cd_part) h->cd_set ^= 1;
    const cdist_rows rr{h->d_rare, h->d_cd_rows, h->d_cd_rows + h->S, h->d_cd_stat + 4 * set, h->d_cd_stat + 4 * (set ^ 1)};
    const bool fuse = !all_rows && !h->cd_part && fuse_intersect && h->K <= 64 && h->D >= 2 && h->D <= 16;
    bool reduce = false;
    if (!all_rows) {
        if (h->D < 2 || h->D > 16) { h->err = "GH_DIST_CDIST: the candidate selection needs 2..16 components"; return GH_ERR_RUNTIME; }
        // the column sums of the fused kernel's workgroup partials ride along (stats_fix_kernel then skips them)
        reduce = h->new0_ready && h->rows > 0 && h->LD <= 16 && h->S < 2048;
        gh_scope t(h, fuse ? "knn_select_cdist_intersect" : "knn_select_cdist");
        const inter_args ia = make_inter_args(h, fuse);
        const int part_mode = h->cd_part ? 1 : 0;
#define GH_CSEL(DD)                                                                                                                \
    knn_select_cdist_kernel<DD><<<dim3((unsigned)h->S + (reduce ? 2u * (unsigned)h->LD : 0u)), dim3(256), 0, h->stream>>>(          \
        h->d_cand, h->d_cnt, h->K, a, h->d_partial, h->d_ovf, h->d_dbg_cnt + h->S, rr, ia, (int)h->S, h->d_blockstats, h->n_vblocks, \
        h->d_stats, part_mode, (fuse && !h->cd_all_ties) ? 1 : 0)
        switch (h->D) {
            case 2: GH_CSEL(2); break;   case 3: GH_CSEL(3); break;   case 4: GH_CSEL(4); break;   case 5: GH_CSEL(5); break;
            case 6: GH_CSEL(6); break;   case 7: GH_CSEL(7); break;   case 8: GH_CSEL(8); break;   case 9: GH_CSEL(9); break;
            case 10: GH_CSEL(10); break; case 11: GH_CSEL(11); break; case 12: GH_CSEL(12); break; case 13: GH_CSEL(13); break;
            case 14: GH_CSEL(14); break; case 15: GH_CSEL(15); break; default: GH_CSEL(16); break;
        }
#undef GH_CSEL
        GH_LAUNCH_CHECK();
    }
    if (!h->cd_part) GH_TRY_ST(cdist_replay_rounds(h, a, rr, all_rows, fuse, h->d_partial));
    h->intersect_done = fuse;
    h->stats_reduced = reduce;
    return GH_OK;
}

// Row partitions, after the all-gather of the ranks' keys: gathered (world, S, K + 2) -> d_merged (S, K), the reference's rows
// (knn_merge_cdist_kernel; the listed rows replayed over all edges), with the intersection phase of every query when
// K <= 64 and 2..16 components (h->intersect_done).
gh_status gh_knn_merge_cdist(gh_engine *h, const uint64_t *gathered, int world) {
    const cdist_args a = make_cdist_args(h);
    const int set = h->cd_set;
    h->cd_set ^= 1;
    const cdist_rows rr{h->d_rare, h->d_cd_rows, h->d_cd_rows + h->S, h->d_cd_stat + 4 * set, h->d_cd_stat + 4 * (set ^ 1)};
    const bool fuse = !h->intersect_done && h->K <= 64 && h->D >= 2 && h->D <= 16;
    const int total = world * (h->K + 1);
    int n2 = 2;
    while (n2 < total) n2 <<= 1;
    if ((size_t)n2 * sizeof(uint64_t) > 48 * 1024) {
        h->err = "world * (n_neighbors + 2) too large for the merge kernel";
        return GH_ERR_INVALID;
    }
    // the listed rows' prefix bound needs the keys a second time, in id order (else: P = E); ids are 32-bit
    const int bound = (size_t)2 * n2 * sizeof(uint64_t) <= 40 * 1024 && h->E < ((int64_t)1 << 31) ? 1 : 0;
    {
        gh_scope t(h, fuse ? "knn_merge_cdist_intersect" : "knn_merge_cdist");
        const inter_args ia = make_inter_args(h, fuse);
#define GH_CMRG(DD) knn_merge_cdist_kernel<DD><<<dim3((unsigned)h->S), dim3(256), sizeof(uint64_t) * (size_t)n2 * (bound ? 2 : 1), h->stream>>>( \
        gathered, world, h->S, h->K, h->E, h->d_merged, rr, ia, h->d_cand, bound)
        if (fuse) { GH_DISPATCH_DIM(h->D, GH_CMRG) } else { GH_CMRG(0); }
#undef GH_CMRG
        GH_LAUNCH_CHECK();
    }
    GH_TRY_ST(cdist_replay_rounds(h, a, rr, false, fuse, h->d_merged));
    h->d_keys_cur = h->d_merged;
    if (fuse) h->intersect_done = true;
    return GH_OK;
}
