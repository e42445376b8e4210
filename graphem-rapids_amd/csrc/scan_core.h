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

// Stage nq query records (plus one spare for the prefetch) of the group starting at s_lo.
// taush gets the exact thresholds tau of the same queries (element QT of the records qt): the
// rare exact path of the scan then needs no global load.
template <int QS, int QT, int NT = 256>
__device__ __forceinline__ void gh_stage_queries(const float *__restrict__ qscan, const float *__restrict__ qt,
                                                 int s_lo, int nq, float4 *qsh, float *taush) {
    const float4 *src = reinterpret_cast<const float4 *>(qscan) + (int64_t)s_lo * (QS / 4);
    for (int i = threadIdx.x; i < (nq + 1) * (QS / 4); i += NT)
        qsh[i] = i < nq * (QS / 4) ? src[i] : make_float4(0.f, 0.f, 0.f, -1.f);
    for (int i = threadIdx.x; i < nq; i += NT) taush[i] = qt[(int64_t)(s_lo + i) * QS + QT];
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
                                                int32_t *__restrict__ cnt) {
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
            const int sg = s_lo + s;
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
// The same pre-filter on the matrix pipe (D <= 3).  For a group of 16 queries and 16 references
//     F[q][r] = (-t_q) + sum_k A[q][k] * B[k][r],   A[q] = (-2q_x, -2q_y, -2q_z, 1),
//                                                    B[.][r] = (m_x, m_y, m_z, c0_r)
// is one v_mfma_f32_16x16x4_f32 with C = -t: a (16 x 4) x (4 x 16) GEMM tile, and a reference can
// only be a candidate of a query if F <= 0.  fp32 MFMA is an exact k-ordered fma chain, so the
// error margin built into t and c0 (gh_filter_eps) covers it.  256 pairs cost 32 cycles of the
// MFMA pipe plus 3 VALU instructions (min3, min, compare), against 12 packed VALU instructions:
// the VALU is left to the spring math of the co-resident workgroups.
// Layout (guide: A lane l = A[l&15][l>>4], B lane l = B[l>>4][l&15], D lane l reg i =
// D[(l>>4)*4+i][l&15]): each wave owns G groups of 16 references, one float per lane and group
// (bq[g] = component l>>4 of reference l&15 of group g); queries are staged transposed in LDS
// (qT[k][q], tneg[q]).  Hits are re-derived exactly as in gh_scan_queries.
typedef float gh_f4 __attribute__((ext_vector_type(4)));

template <int D, int G, int HITBUF>
__device__ __forceinline__ void gh_scan_queries_mfma(const float (&bq)[G], uint32_t id0 /* id of ref (group 0, col 0) */,
                                                     int nvalid /* valid refs of this wave */, const float *qT,
                                                     const float *tneg, int nq, int s_lo,
                                                     const float *__restrict__ qt, uint64_t *hkey, int *hq,
                                                     int *hcount, uint64_t *__restrict__ cand,
                                                     int32_t *__restrict__ cnt) {
    static_assert(D <= 3, "one K=4 MFMA per tile");
    static_assert(G % 8 == 0, "eight MFMAs are kept in flight");
    constexpr int QS = 4, QT = 3;
    const int lane = threadIdx.x & 63;
    const int col = lane & 15, kq = lane >> 4;
    const int ngroups = (nq + 15) >> 4;
    for (int qg = 0; qg < ngroups; ++qg) {
        // A operand: component kq of query qg*16 + col (k = 3 is the constant 1)
        const float a = kq < 3 ? qT[kq * GH_SCAN_QGROUP + qg * 16 + col] : 1.0f;
        // C operand: -t of the four queries this lane's accumulator rows belong to
        const float4 tn = *reinterpret_cast<const float4 *>(tneg + qg * 16 + kq * 4);
        const gh_f4 c = {tn.x, tn.y, tn.z, tn.w};
#pragma unroll
        for (int g0 = 0; g0 < G; g0 += 8) {
            gh_f4 acc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq[g0 + u], c, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                // F <= 0 somewhere  <=>  some sign bit set (t carries a +1e-30 bias, so a passing
                // F is never +0): one v_or3, one v_or, one integer compare per 256 pairs
                const int bits = __float_as_int(acc[u].x) | __float_as_int(acc[u].y) | __float_as_int(acc[u].z) |
                                 __float_as_int(acc[u].w);
                if (__builtin_amdgcn_ballot_w64(bits < 0) != 0) {  // rare, wave-uniform
                    const int g = g0 + u;
                    // exact coordinates of this lane's reference: components live in lanes col, 16+col, 32+col
                    float mref[3];
#pragma unroll
                    for (int d = 0; d < 3; ++d) mref[d] = __shfl(bq[g], d * 16 + col, GH_WAVE);
                    const int jref = g * 16 + col;
                    if (bits < 0 && jref < nvalid) {
                        const float av[4] = {acc[u].x, acc[u].y, acc[u].z, acc[u].w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int ql = qg * 16 + kq * 4 + i;
                            if (__float_as_int(av[i]) < 0 && ql < nq) {
                                const int sg = s_lo + ql;
                                float d2 = 0.0f;
#pragma unroll
                                for (int d = 0; d < D; ++d) {
                                    const float df = qt[(int64_t)sg * QS + d] - mref[d];
                                    d2 = fmaf(df, df, d2);
                                }
                                if (d2 <= qt[(int64_t)sg * QS + QT]) {
                                    const int p = atomicAdd(hcount, 1);
                                    const uint64_t key = gh_key(d2, id0 + (uint32_t)jref);
                                    if (p < HITBUF) { hkey[p] = key; hq[p] = sg; }
                                    else gh_append_candidate(cand, cnt, sg, key);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
}
