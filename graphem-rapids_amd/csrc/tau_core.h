#pragma once
// Thresholds of the filtered scan: tau of one query = the K-th smallest of its G group minima (setup_core.h), an upper
// bound of its K-th smallest distance over ALL own edges, and from it the query's pre-filter records (scan_core.h).
// One WAVE per query.  Runs either as its own launch (knn_tau_kernel, knn.hip) or at the head of the fused spring+scan
// launch (fused.hip): the first workgroups of that grid compute the thresholds while the others are in their spring
// phase, which does not need them, and publish them through a counter; see gh_tau_publish / gh_tau_wait.
#include "common.h"
#include "scan_core.h"
#include "setup_core.h"

struct gh_tau_args {
    const uint32_t *gmin;     // (S, Gpad) group minima as float bits
    int64_t Gpad;
    int D, QS, QT, K, S;
    float *qt, *qscan;        // query records (coordinates, tau) and pre-filter records (-2q, t)
    _Float16 *qA;             // operand rows of the MFMA forms of the filter, or null
    int qA_kb;                // host side: which row form knn_tau_kernel writes (-1 none, 0 split form of D <= 3)
    int32_t *qexact;          // [0] = count, then the queries outside the f16 range of that filter
    int32_t *tcount_reset;    // the touched-list counter to reset (set-up ran inside the previous normalise launch), or null
    // -- inside the fused launch only
    unsigned *flag;           // queries published so far by THIS launch (zeroed by the iteration's set-up, setup_core.h)
    unsigned target;          // value of *flag once this launch's S queries are out (= S)
    int nblocks;              // workgroups at the head of the grid that compute thresholds (0: a launch of their own did)
    int32_t *wait_failed;     // set when a consumer gave up waiting (cannot happen while workgroups start in index order)
    // -- rides here because every fused kernel takes this struct
    int cdist;                // GH_DIST_CDIST: the parked candidates carry ATen's cdist value in their keys (scan_core.h gh_aten_cdist)
};

// One wave, one query: the minima sit NV per lane in
// registers (more than 64 * NV groups: folded by min, which only makes groups coarser), K rounds of a
// wave-wide minimum retire the smallest value each (equal values retire together: the bound can only
// get looser).  Then the pre-filter records of the scan (scan_core.h).
template <int NV>
__device__ __forceinline__ uint32_t gh_tau_kth(const gh_tau_args &a, int64_t qi, int lane, unsigned long long *st8 = nullptr) {
    const int K = a.K;
    const int64_t G = a.Gpad;
    const uint32_t *row = a.gmin + qi * a.Gpad;
    uint32_t v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = 0x7F800000u;
    for (int64_t base = 0; base < G; base += 64 * NV) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int64_t i = base + j * 64 + lane;
            const uint32_t x = i < G ? row[i] : 0x7F800000u;
            v[j] = min(v[j], x);
        }
    }
    // K-th smallest of the group minima AS A MULTISET: a round takes the smallest remaining value and all its copies at
    // once, counting them.  (Retiring the copies without counting made tau the K-th DISTINCT value: on a collapsed layout
    // -- a hub that has flown off holds the variance, thousands of midpoints coincide to the last bit -- that let
    // 12 000 candidates per query through instead of 300 and sent 8 queries per iteration to the exact fallback.)
    uint32_t kth = 0x7F800000u;
    int need = K;
    if (st8 && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st8[1] = wall_clock64(); }
    if (K >= 24) {
        // many neighbours: the same order statistic bit by bit from the top -- 31 rounds of NV compares and ballots whatever
        // K is, against K rounds of a wave minimum (k = 32: 6.7 -> 2.5 us); values are non-negative float bits
        uint32_t prefix = 0;
        for (int bit = 30; bit >= 0; --bit) {
            const uint32_t t = prefix | (1u << bit);
            int below = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j) below += __popcll(__ballot(v[j] < t));
            if (below < K) prefix = t;   // fewer than K values under t: the K-th smallest has this bit
        }
        return prefix;
    }
    while (need > 0) {
        uint32_t m = v[0];
#pragma unroll
        for (int j = 1; j < NV; ++j) m = min(m, v[j]);
        m = gh_row_min_u32(m);
        m = min(min((uint32_t)__builtin_amdgcn_readlane((int)m, 0), (uint32_t)__builtin_amdgcn_readlane((int)m, 16)),
                min((uint32_t)__builtin_amdgcn_readlane((int)m, 32), (uint32_t)__builtin_amdgcn_readlane((int)m, 48)));
        kth = m;
        if (m == 0x7F800000u) break;  // fewer than K occupied groups: tau = inf, the candidate lists overflow, exact fallback
        int mine = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            mine += v[j] == m ? 1 : 0;
            v[j] = v[j] == m ? 0x7F800000u : v[j];
        }
        for (int t = 1;; ++t) {   // copies over the wave: one ballot per multiplicity level (almost always a single one)
            const unsigned long long b = __ballot(mine >= t);
            if (!b) break;
            need -= __popcll(b);
        }
    }
    return kth;
}

// The records of one query from its threshold: tau into the query record, the pre-filter record (-2q, t) of the
// packed-VALU scan and, FORM == 0, the operand row of the split-f16 MFMA form (D <= 3; the wide form builds its rows
// when it stages the queries, fused.hip).
// FORM is a compile-time constant on purpose: this code runs once per CU from a cold instruction cache -- with the three
// row forms and the three register counts of gh_tau_kth inlined into one another (9 copies, 3200 instructions, taken
// branches over all but one) knn_tau_kernel went from 7.9 to 34 us and the producers inside the fused launch from 11 to 27.
template <int FORM>
__device__ __forceinline__ void gh_tau_records(const gh_tau_args &a, int64_t qi, int lane, float qcoord, uint32_t kth,
                                               unsigned long long *st8 = nullptr) {
    const int D = a.D, QS = a.QS, QT = a.QT;
    const float tau = __uint_as_float(kth);
    if (st8 && lane == 0) st8[2] = wall_clock64();
    float qs[16];
    float qn = 0.0f;
#pragma unroll
    for (int d = 0; d < 16; ++d) {   // every lane gets all coordinates (coordinates past D are 0: fma(0, 0, s) == s)
        qs[d] = __shfl(qcoord, d, 64);
        qn = fmaf(qs[d], qs[d], qn);
    }
    // scan record of the pre-filter (scan_core.h): (-2q, t),  t = tau - |q|^2 + eps*(2|q|^2 + tau)
    // Inside the fused launch every result leaves as an agent-scope atomic store (global_store ... sc1: written through,
    // no copy left in this XCD's L2) -- see gh_tau_produce.
    const bool coh = a.nblocks > 0;
    auto st = [coh](void *p, uint32_t v) {
        if (coh) __hip_atomic_store(static_cast<uint32_t *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *static_cast<uint32_t *>(p) = v;
    };
    if (lane < QS && lane != QT) st(&a.qscan[qi * QS + lane], __float_as_uint(lane < D ? -2.0f * qcoord : 0.0f));
    if (lane == 0) {
        st(&a.qt[qi * QS + QT], __float_as_uint(tau));
        const float eps = gh_filter_eps(D);
        // + 1e-30: a must-pass value is then strictly negative even when every magnitude is 0 (the
        // MFMA form of the filter tests sign bits)
        st(&a.qscan[qi * QS + QT], __float_as_uint(fmaf(eps, fmaf(2.0f, qn, tau), tau - qn) + 1e-30f));
    }
    if constexpr (FORM == 0) {  // operand row of the split-f16 MFMA form (scan_core.h): lane k < 8 stores elements 2k, 2k+1
        _Float16 rowh[16];
        const bool ok = gh_mf_query_row(qs, D, tau, rowh);
        if (lane == 0 && !ok) st(&a.qexact[1 + atomicAdd(&a.qexact[0], 1)], (uint32_t)qi);
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t pk = (uint32_t)__builtin_bit_cast(uint16_t, rowh[2 * k]) | ((uint32_t)__builtin_bit_cast(uint16_t, rowh[2 * k + 1]) << 16);
            mine = lane == k ? pk : mine;
        }
        if (lane < 8) st(reinterpret_cast<uint32_t *>(a.qA + qi * 16) + lane, mine);
    }
}

// One query by one wave.  Registers per lane for the group minima: the selection rounds rescan them all
// (a few groups past 64 * NV fold onto the first lanes by min: coarser groups, a valid and barely looser bound).
__device__ __forceinline__ int gh_tau_nv(int64_t Gpad) { return Gpad <= 64 * 8 + 64 ? 8 : Gpad <= 64 * 16 + 256 ? 16 : 32; }
__host__ inline int gh_tau_nv_host(int64_t Gpad) { return Gpad <= 64 * 8 + 64 ? 8 : Gpad <= 64 * 16 + 256 ? 16 : 32; }
template <int FORM, int NVFIX = 0 /* 0: chosen at run time */>
__device__ __forceinline__ void gh_tau_query_any(const gh_tau_args &a, int64_t qi, int lane, unsigned long long *st8 = nullptr) {
    // set-up done inside the previous normalise launch: the touched-list counter is reset here instead
    if (a.tcount_reset && qi == 0 && lane == 0) *a.tcount_reset = 0;
    // lane d holds coordinate d (issued with the loads of gh_tau_kth; always the sc1 form -- past the L1: inside the fused
    // launch no CU may hold a copy of a record line from before its threshold was stored, and with a plain-load
    // alternative the compiler waits for this load before it issues the others)
    const float qcoord = lane < a.D ? gh_ld_f32(&a.qt[qi * a.QS + lane], true) : 0.0f;
    uint32_t kth;
    if constexpr (NVFIX != 0) {
        kth = gh_tau_kth<NVFIX>(a, qi, lane, st8);
    } else {
        const int nv = gh_tau_nv(a.Gpad);
        if (nv == 8) kth = gh_tau_kth<8>(a, qi, lane, st8);
        else if (nv == 16) kth = gh_tau_kth<16>(a, qi, lane, st8);
        else kth = gh_tau_kth<32>(a, qi, lane, st8);
    }
    gh_tau_records<FORM>(a, qi, lane, qcoord, kth, st8);
}

// ---- thresholds inside the fused launch ---------------------------------------------------------------------------
// Producer: workgroup b < a.nblocks, wave w takes query b * (NT / 64) + w; every result is stored write-through (sc1)
// and the wave waits for its stores before it moves the counter.  (An agent-scope release instead -- buffer_wbl2 -- has
// to write back whatever the spring phases of the OTHER workgroups have dirtied in this XCD's L2 in the meantime: the
// consumers then waited 25-35 us for their thresholds.)
// Consumer: after its spring phase a workgroup waits until the counter says all S queries of THIS launch are out
// (normally they long are: ~6 us against a spring phase of ~8), then reads every threshold-dependent value with
// sc1 loads (they go past the CU's L1, the one cache that can hold a stale copy of released data; 16 bytes wide they cost
// what a plain load costs).  No acquire fence on the consumer side -- an L1 invalidate per workgroup, ~1.7-7 us each by
// the MI355X guide's price list, on CUs whose other workgroups are gathering; tools/micro/grid_barrier.hip "flag" runs
// this protocol with stale copies planted (0 stale values in 200 launches x 8192 consumers).  The wait cannot deadlock while workgroups are started in index
// order (the producers are the first of the grid and wait for nobody); it is bounded all the same: a consumer that
// gives up sets *wait_failed, which the host reports at the next synchronisation instead of results.
template <int NT, int FORM>
__device__ __forceinline__ void gh_tau_produce(const gh_tau_args &a, unsigned long long *st8 = nullptr /* diagnostic: this workgroup's 8 stamps */) {
    const int lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (qi >= a.S) return;
    if (threadIdx.x >= 64) st8 = nullptr;
    if (st8 && lane == 0) st8[0] = wall_clock64();
    gh_tau_query_any<FORM>(a, qi, lane, st8);
    if (st8 && lane == 0) st8[3] = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have left before the counter moves
    if (st8 && lane == 0) st8[4] = wall_clock64();
    if (lane == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (st8) { st8[6] = old; st8[5] = wall_clock64(); }
    }
}
__host__ __device__ inline int gh_tau_blocks(int S, int NT) { return (S + NT / 64 - 1) / (NT / 64); }

__device__ __forceinline__ void gh_tau_wait(const gh_tau_args &a) {  // one thread; follow with a barrier
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(a.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.target) < 0) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > (1u << 20)) { *a.wait_failed = 1; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // orders the loads below after the poll; no cache invalidate
}

