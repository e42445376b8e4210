#pragma once
// The inner loop of the filtered KNN scan, shared by the stand-alone scan kernel (knn.hip) and
// the fused spring+scan kernel (fused.hip).
#include "common.h"
#include "engine.h"

typedef float gh_f2 __attribute__((ext_vector_type(2)));
#define GH_SCAN_QGROUP 256

// Query records: QS floats per query = coordinates, then tau.  D <= 3 packs into one 16-byte
// record (x, y, z|0, tau) so a query is ONE 16-byte load.
__host__ __device__ inline int gh_qs(int D, int LD) { return D <= 3 ? 4 : LD + 4; }
__host__ __device__ inline int gh_qtau(int D, int LD) { return D <= 3 ? 3 : LD; }

__device__ __forceinline__ void gh_append_candidate(uint64_t *__restrict__ cand, int32_t *__restrict__ cnt, int sg,
                                                    uint64_t key) {
    const int p = atomicAdd(&cnt[sg * GH_CNT_STRIDE], 1);
    if (p < GH_CAND_CAP) cand[(int64_t)sg * GH_CAND_CAP + p] = key;
}

// Loads of values another workgroup of the SAME launch produced and released (tau_core.h: thresholds inside the fused
// launch): coherent = agent-scope atomic loads (global_load ... sc1), served past whatever stale copy an L1 or this XCD's
// L2 holds; otherwise plain loads.
__device__ __forceinline__ uint32_t gh_ld_u32(const void *p, bool coherent) {
    return coherent ? __hip_atomic_load(static_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                    : *static_cast<const uint32_t *>(p);
}
__device__ __forceinline__ float gh_ld_f32(const float *p, bool coherent) { return __uint_as_float(gh_ld_u32(p, coherent)); }
// 16 bytes with ONE request (sc1 loads are L2-served at the plain rate when 16 bytes wide; a dword sc1 load costs a
// whole L2 request for 4 bytes -- the first version staged the queries that way and the fused kernel went from 127 to
// 169 us).  The wait sits in the same statement: the compiler does not track the counter of a load it cannot see.
__device__ __forceinline__ float4 gh_ld_f4(const float4 *p, bool coherent) {
    if (!coherent) return *p;
    float4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void gh_ld_f4x3(const float4 *p0, const float4 *p1, const float4 *p2, bool coherent, float4 &a,
                                           float4 &b, float4 &c) {
    if (!coherent) { a = *p0; b = *p1; c = *p2; return; }
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\t"
                 "global_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(p0), "v"(p1), "v"(p2) : "memory");
}

// Stage nq query records (plus one spare for the prefetch) of the group starting at s_lo.
// taush gets the exact thresholds tau of the same queries (element QT of the records qt): the
// rare exact path of the scan then needs no global load.
template <int QS, int QT, int NT = 256>
__device__ __forceinline__ void gh_stage_queries(const float *__restrict__ qscan, const float *__restrict__ qt,
                                                 int s_lo, int nq, float4 *qsh, float *taush, bool coherent = false) {
    const float4 *src = reinterpret_cast<const float4 *>(qscan) + (int64_t)s_lo * (QS / 4);
    for (int i = threadIdx.x; i < (nq + 1) * (QS / 4); i += NT)
        qsh[i] = i < nq * (QS / 4) ? gh_ld_f4(src + i, coherent) : make_float4(0.f, 0.f, 0.f, -1.f);
    for (int i = threadIdx.x; i < nq; i += NT) taush[i] = gh_ld_f32(qt + (int64_t)(s_lo + i) * QS + QT, coherent);
}

// Relative slack of the pre-filter below.  |filter value - exact fma-chain dist2| is bounded by
// about (2D + 12) * 2^-24 * (|q|^2 + |m|^2) (three roundings per norm, one per fma of the dot
// product, D + 1 in the exact chain); eps = (D + 6) * 2^-21 is more than 4x that for every D.
__host__ __device__ inline float gh_filter_eps(int D) { return (float)(D + 6) * 4.76837158203125e-07f; }

// Per-reference constant of the pre-filter: c0 = |m|^2 * (1 - eps)  (+inf for padding slots).
template <int D>
__device__ __forceinline__ float gh_ref_c0(const float *mv, bool valid) {
    if (!valid) return INFINITY;
    float rn = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) rn = fmaf(mv[d], mv[d], rn);
    return fmaf(-gh_filter_eps(D), rn, rn);
}

// GH_DIST_CDIST (cdist.hip): the value ATen's torch.cdist gives the pair (its matmul form, taken whenever either side has
// more than 25 rows -- always on the scan path): |x|^2 = sum of rounded squares left to right; acc = fma(-2 q_d, m_d, acc) from
// 0; acc += |q|^2; acc += |m|^2; sqrt(max(acc, 0)).  A candidate that passed the exact test d2 <= tau is parked with THIS value
// in its key, so that the selection ranks what the reference ranks without gathering the pair's rows again (round 3 re-valued
// the candidates in the selection launch: two dependent gathers, 27 against 15 us).
template <int D>
__device__ __forceinline__ float gh_aten_cdist(const float *q, const float *m) {
    float qn = 0.0f, mn = 0.0f, acc = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float sq = q[d] * q[d];
        qn = qn + sq;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float sq = m[d] * m[d];
        mn = mn + sq;
        acc = fmaf(q[d] * -2.0f, m[d], acc);
    }
    acc = acc + qn;
    acc = acc + mn;
    return sqrtf(fmaxf(acc, 0.0f)) + 0.0f;   // (+ 0: a -0 must not read as the largest key)
}

// R reference midpoints per thread, held as R/2 packed pairs so the arithmetic runs on
// v_pk_fma_f32 (two references per VALU instruction: measured, a plain fp32 VALU op occupies a
// SIMD for 4 cycles, so packed math is the only way past half of the fp32 vector peak).  The query
// group streams past as broadcast ds_read_b128, prefetched one query ahead.
//
// Hot loop = conservative PRE-FILTER in norm-expansion form: with the scan record
// (-2q, t) of a query and c0 of a reference,  a = c0 - 2 q.m  (D packed fmas per pair) and the
// reference can only be a candidate if a <= t, where t = tau - |q|^2 + eps*(2|q|^2 + tau) absorbs
// every rounding difference to the exact distance (gh_filter_eps).  Per query: D/2 fma per
// reference, one min-tree over the R values, ONE compare, ONE branch.
// Rare path (a handful of references per query and workgroup): the EXACT squared distance in
// difference form, an fma chain in coordinate order (bit-identical to the oracle's go_d2), decides
// with dist2 <= tau and forms the key.  Hits are parked in LDS (hkey/hq/hcount): the global
// returning atomic that reserves a list slot costs a ~1.5 us round trip and must not sit here.
template <int D, int R, int HITBUF>
__device__ __forceinline__ void gh_scan_queries(const gh_f2 (&m)[R / 2][D], const gh_f2 (&c0)[R / 2],
                                                const uint32_t (&id)[R], const float4 *qsh, int nq, int s_lo,
                                                const float *taush, uint64_t *hkey, int *hq,
                                                int *hcount, uint64_t *__restrict__ cand,
                                                int32_t *__restrict__ cnt, const int *qmap = nullptr /* LDS: query of slot s (ivf.hip) */,
                                                int cdist = 0 /* GH_DIST_CDIST: keys carry ATen's cdist value */) {
    constexpr int LD = D <= 4 ? 4 : D <= 8 ? 8 : 16;
    constexpr int QS = D <= 3 ? 4 : LD + 4;
    constexpr int QT = D <= 3 ? 3 : LD;
    float4 rec[QS / 4], nxt[QS / 4];
#pragma unroll
    for (int i = 0; i < QS / 4; ++i) nxt[i] = qsh[i];
#pragma unroll 2
    for (int s = 0; s < nq; ++s) {
#pragma unroll
        for (int i = 0; i < QS / 4; ++i) rec[i] = nxt[i];
#pragma unroll
        for (int i = 0; i < QS / 4; ++i) nxt[i] = qsh[(s + 1) * (QS / 4) + i];  // broadcast read, next query
        const float *qv = reinterpret_cast<const float *>(rec);  // (-2q_0 .. -2q_{D-1}, .., t)
        const float t = qv[QT];
        gh_f2 a[R / 2];
#pragma unroll
        for (int r = 0; r < R / 2; ++r) {
            gh_f2 acc = c0[r];
#pragma unroll
            for (int d = 0; d < D; ++d) acc = __builtin_elementwise_fma((gh_f2){qv[d], qv[d]}, m[r][d], acc);
            a[r] = acc;
        }
        float amin = fminf(a[0].x, a[0].y);
#pragma unroll
        for (int r = 1; r < R / 2; ++r) amin = fminf(amin, fminf(a[r].x, a[r].y));
        if (amin <= t) {  // rare: some reference of this thread may be a candidate
            const int sg = qmap ? qmap[s] : s_lo + s;
            float q[D];
#pragma unroll
            for (int d = 0; d < D; ++d) q[d] = -0.5f * qv[d];  // the record holds -2q: exact both ways
            const float tau = taush[s];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float ar = (r & 1) ? a[r / 2].y : a[r / 2].x;
                if (ar <= t) {
                    float d2 = 0.0f;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        const float df = q[d] - ((r & 1) ? m[r / 2][d].y : m[r / 2][d].x);
                        d2 = fmaf(df, df, d2);
                    }
                    if (d2 <= tau) {
                        if (cdist) {
                            float mr[D];
#pragma unroll
                            for (int d = 0; d < D; ++d) mr[d] = (r & 1) ? m[r / 2][d].y : m[r / 2][d].x;
                            d2 = gh_aten_cdist<D>(q, mr);
                        }
                        const int p = atomicAdd(hcount, 1);
                        if (p < HITBUF) { hkey[p] = gh_key(d2, id[r]); hq[p] = sg; }
                        else gh_append_candidate(cand, cnt, sg, gh_key(d2, id[r]));
                    }
                }
            }
        }
    }
}

template <int HITBUF, int NT = 256>
__device__ __forceinline__ void gh_flush_hits(const uint64_t *hkey, const int *hq, const int *hcount,
                                              uint64_t *__restrict__ cand, int32_t *__restrict__ cnt) {
    const int nh = min(*hcount, HITBUF);
    for (int i = threadIdx.x; i < nh; i += NT) gh_append_candidate(cand, cnt, hq[i], hkey[i]);
}

// ---------------------------------------------------------------------------------
// The same conservative pre-filter on the matrix pipe (D <= 3): f16 MFMA on SPLIT operands.
//
// For 32 queries x 32 references one v_mfma_f32_32x32x16_f16 evaluates
//     F[s][j] = C0_j - 2 q_s . m_j - T_s          (candidate only if F <= 0)
// with every fp32 input carried as a sum of f16 pieces, x = x_h + x_l (+ x_m), so that the 16
// products of the contraction are EXACT in the fp32 accumulator (11 x 11 significand bits):
//     k = 3d+0, 3d+1, 3d+2 (d = 0..2):  a_h*m_h, a_h*m_l, a_l*m_h      a = -2 q_d, m = m_d
//     k = 9..11:   1 * (C0_h, C0_m, C0_l)        C0 = |m|^2 (1 - eps)
//     k = 12..14:  (-T_h, -T_m, -T_l) * 1        T  = tau - |q|^2 + eps (|q|^2 + tau) + 2^-20
//     k = 15:      0
// What is lost: the a_l*m_l terms and the residues of the two-piece splits (<= 3 * 2^-22 |a||m| per
// coordinate, or 2^-25 absolute when a low piece is a f16 subnormal) and the roundings of the 16
// accumulations (<= 2^-20 (C0 + 2|q||m| + |T|)); together below 2^-18 (|q|^2 + |m|^2 + tau) + 2^-21.
// eps = 2^-17 on both sides plus the absolute 2^-20 is twice that, and also covers the rounding of
// the exact fma-chain distance the decision is finally taken on (gh_scan_queries' rare path, the
// same here).  f16 range: pieces stay finite for |coordinate| <= GH_MF_RANGE and tau <= GH_MF_TAU_MAX
// (C0 <= 3 * 128^2 = 49152, |T| < 65504); anything outside is marked "never passes" (GH_MF_NEVER in the C0 or
// -T slot, > any other term) and handled exactly by the caller: queries through the list qexact,
// references by the lane that owns them.
// Cost: 32 cycles of the matrix pipe + ~10 VALU instructions (min tree over the 16 accumulators,
// compare, branch) per 1024 pairs, against 5 packed VALU instructions per 4 pairs per lane above.
typedef _Float16 gh_h8 __attribute__((ext_vector_type(8)));
typedef float gh_f16x __attribute__((ext_vector_type(16)));
#define GH_MF_RANGE 128.0f
#define GH_MF_TAU_MAX 49000.0f
#define GH_MF_NEVER 60000.0f
#define GH_MF_EPS 7.62939453125e-06f   /* 2^-17 */
#define GH_MF_ABS 9.5367431640625e-07f /* 2^-20 */

__device__ __forceinline__ void gh_split2(float x, _Float16 &h, _Float16 &l) {
    h = (_Float16)x;
    l = (_Float16)(x - (float)h);
}
__device__ __forceinline__ void gh_split3(float x, _Float16 &h, _Float16 &m, _Float16 &l) {
    h = (_Float16)x;
    const float r = x - (float)h;
    m = (_Float16)r;
    l = (_Float16)(r - (float)m);
}

// A-operand row of one query (16 halfs; lanes 0-31 of a wave read elements 0..7, lanes 32-63
// elements 8..15).  q: D <= 3 coordinates (missing ones 0).  Returns false (and a never-pass row)
// when the query is outside the f16 range: the caller must put it on the exact list.
__device__ __forceinline__ bool gh_mf_query_row(const float *q /* 3 coordinates, those past D are 0 */, int D, float tau,
                                                _Float16 *row) {
    // always three rounds with constant indices (a zero coordinate contributes zeros everywhere: fma(0, 0, s) == s,
    // split(0) = (0, 0)); with a run-time trip count `row` is indexed dynamically and the compiler moves it to LDS
    (void)D;
    bool ok = tau <= GH_MF_TAU_MAX;  // false for inf / NaN too
    float qn = 0.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        ok = ok && fabsf(q[d]) <= GH_MF_RANGE;
        qn = fmaf(q[d], q[d], qn);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) row[k] = (_Float16)0.0f;
    if (!ok) {
        row[12] = (_Float16)GH_MF_NEVER;
        return false;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        _Float16 h, l;
        gh_split2(-2.0f * q[d], h, l);
        row[3 * d] = h; row[3 * d + 1] = h; row[3 * d + 2] = l;
    }
    row[9] = row[10] = row[11] = (_Float16)1.0f;
    const float T = (tau - qn) + (GH_MF_EPS * (qn + tau) + GH_MF_ABS);
    gh_split3(-T, row[12], row[13], row[14]);
    return true;
}

// B-operand column of one reference: the 8 halfs of lane half hsel (0: k = 0..7, 1: k = 8..15).
// mv = (m_0, m_1, m_2, ...) with missing coordinates 0.  valid = false: padding slot (never
// passes).  Returns false for a real reference outside the f16 range (never passes here; the
// caller scans it exactly).
__device__ __forceinline__ bool gh_mf_ref_col(const float *mv, bool valid, int hsel, gh_h8 &col) {
    const bool ok = fabsf(mv[0]) <= GH_MF_RANGE && fabsf(mv[1]) <= GH_MF_RANGE && fabsf(mv[2]) <= GH_MF_RANGE;
    const _Float16 z = (_Float16)0.0f;
    if (!valid || !ok) {
        // k = 9 (C0_h) = NEVER beats every in-range -T; k = 12..14 stay 1 so that a never-pass
        // query row (NEVER in its -T_h slot) still yields NEVER, not 0
        const _Float16 one = (_Float16)1.0f;
        col = hsel ? (gh_h8){z, (_Float16)GH_MF_NEVER, z, z, one, one, one, z} : (gh_h8){z, z, z, z, z, z, z, z};
        return !valid;
    }
    _Float16 h0, l0, h1, l1, h2, l2;
    gh_split2(mv[0], h0, l0);
    gh_split2(mv[1], h1, l1);
    gh_split2(mv[2], h2, l2);
    if (hsel == 0) {
        col = (gh_h8){h0, l0, h0, h1, l1, h1, h2, l2};
    } else {
        float c0 = 0.0f;
#pragma unroll
        for (int d = 0; d < 3; ++d) c0 = fmaf(mv[d], mv[d], c0);
        c0 = fmaf(-GH_MF_EPS, c0, c0);
        _Float16 ch, cm, cl;
        gh_split3(c0, ch, cm, cl);
        const _Float16 one = (_Float16)1.0f;
        col = (gh_h8){h2, ch, cm, cl, one, one, one, z};
    }
    return true;
}

// ---------------------------------------------------------------------------------
// Wide rows (5 <= D <= 16) on the matrix pipe: the same test F = C0_j - 2 q_s . m_j - T_s <= 0 with every
// coordinate as ONE f16 piece.  Contraction of 16 * KB (KB = 1 for D <= 10, else 2):
//     k = 0 .. D-1:            f16(-2 q_d) * f16(m_d)
//     k = base .. base+2:      1 * (C0_h, C0_m, C0_l)           base = 10 (KB = 1) / 16 (KB = 2)
//     k = base+3 .. base+5:    (-T_h, -T_m, -T_l) * 1
// Products of two f16 are exact in the fp32 accumulator; what is lost is the low part of every coordinate:
//     |a m - f16(a) f16(m)| <= |a| |m - f16(m)| + |a - f16(a)| |f16(m)|,   |x - f16(x)| <= max(2^-11 |x|, 2^-25)
// (2^-25: half the spacing of f16 subnormals), summed over the coordinates with a = -2q:
//     <= 2^-10 (1 + 2^-11) (|q|^2 + |m|^2) + 2^-25 (2 |q|_1 + |m|_1) (1 + 2^-11),
// plus the <= 32 roundings of the accumulation, <= 2^-19 (|q|^2 + 2 |m|^2 + tau), plus the three-piece splits of C0 and
// T (2^-33 relative, 2^-25 absolute).  Covered with margin by
//     C0 = |m|^2 (1 - eps) - 2^-24 |m|_1,      T = tau - |q|^2 + eps (|q|^2 + tau) + 2^-23 |q|_1 + 2^-20,
// eps = 1.25 * 2^-10 (needed: 2^-10 (1 + 2^-11) + 2^-18 = 0.98e-3 on the norms, 2^-18 on tau); the norms are the fp32
// fma chains (relative error <= 17 * 2^-24, far inside the margin) and the decision is finally taken on the exact chain
// of the difference form, as in the other forms.  Looser than the split form of D <= 3 by design: a pair passes when its
// squared distance is within ~2 eps (|q|^2 + |m|^2) of tau -- in 16 dimensions that is 0.04 against thresholds of
// several units; at D = 6 on a million vertices it lets ~3x the necessary pairs through to the exact check.
// f16 range: |coordinate| <= GH_MF_RANGE, |m|^2 and |q|^2 <= GH_MFW_NORM_MAX, tau <= GH_MF_TAU_MAX; anything outside is
// marked never-pass and handled exactly by the caller, as in the split form.
#define GH_MFW_EPS 1.220703125e-3f
#define GH_MFW_NORM_MAX 49000.0f
template <int KB> struct gh_mfw { static constexpr int base = KB == 1 ? 10 : 16; static constexpr int K = 16 * KB; };

// A-operand row of one query: 16 * KB halfs.  q: 16 coordinates (those past D are 0).  false: outside the f16 range
// (a never-pass row; the caller puts the query on the exact list).
template <int KB>
__device__ __forceinline__ bool gh_mfw_query_row(const float *q, float tau, _Float16 *row) {
    constexpr int base = gh_mfw<KB>::base, ND = KB == 1 ? 10 : 16;
    bool ok = tau <= GH_MF_TAU_MAX;  // false for inf / NaN too
    float qn = 0.0f, l1 = 0.0f;
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        ok = ok && fabsf(q[d]) <= GH_MF_RANGE;
        qn = fmaf(q[d], q[d], qn);
        l1 += fabsf(q[d]);
    }
    ok = ok && qn <= GH_MFW_NORM_MAX;
#pragma unroll
    for (int k = 0; k < 16 * KB; ++k) row[k] = (_Float16)0.0f;
    if (!ok) {
        row[base + 3] = (_Float16)GH_MF_NEVER;
        return false;
    }
#pragma unroll
    for (int d = 0; d < ND; ++d) row[d] = (_Float16)(-2.0f * q[d]);
    row[base] = row[base + 1] = row[base + 2] = (_Float16)1.0f;
    const float T = (tau - qn) + (GH_MFW_EPS * (qn + tau) + (1.1920928955078125e-07f * l1 + GH_MF_ABS));
    gh_split3(-T, row[base + 3], row[base + 4], row[base + 5]);
    return true;
}

// B-operand column of one reference: for each of the KB blocks of 16 the 8 halfs of lane half hsel.
// mv: LD >= D floats (pad 0).  valid = false: padding slot.  false: a real reference outside the f16 range (never passes
// here; the caller scans it exactly).
template <int D, int KB>
__device__ __forceinline__ bool gh_mfw_ref_col(const float *mv, bool valid, int hsel, gh_h8 (&col)[KB]) {
    constexpr int base = gh_mfw<KB>::base;
    static_assert(D <= (KB == 1 ? 10 : 16), "contraction too short for this dimension");
    bool ok = true;
    float c0 = 0.0f, l1 = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        ok = ok && fabsf(mv[d]) <= GH_MF_RANGE;
        c0 = fmaf(mv[d], mv[d], c0);
        l1 += fabsf(mv[d]);
    }
    ok = ok && c0 <= GH_MFW_NORM_MAX;
    _Float16 row[16 * KB];
#pragma unroll
    for (int k = 0; k < 16 * KB; ++k) row[k] = (_Float16)0.0f;
    row[base + 3] = row[base + 4] = row[base + 5] = (_Float16)1.0f;   // also on a never-pass column: a never-pass row then yields NEVER, not 0
    if (!valid || !ok) {
        row[base] = (_Float16)GH_MF_NEVER;   // C0_h: beats every in-range -T
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) row[d] = (_Float16)mv[d];
        const float c = fmaf(-GH_MFW_EPS, c0, c0) - 5.9604644775390625e-08f * l1;
        gh_split3(c, row[base], row[base + 1], row[base + 2]);
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int i = 0; i < 8; ++i) col[kb][i] = hsel ? row[kb * 16 + 8 + i] : row[kb * 16 + i];
    }
    return ok || !valid;
}
