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
template <int QS, int NT = 256>
__device__ __forceinline__ void gh_stage_queries(const float *__restrict__ qt, int s_lo, int nq, float4 *qsh) {
    const float4 *src = reinterpret_cast<const float4 *>(qt) + (int64_t)s_lo * (QS / 4);
    for (int i = threadIdx.x; i < (nq + 1) * (QS / 4); i += NT)
        qsh[i] = i < nq * (QS / 4) ? src[i] : make_float4(0.f, 0.f, 0.f, -1.f);
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
                                                const float *__restrict__ qt, uint64_t *hkey, int *hq,
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
            for (int d = 0; d < D; ++d) q[d] = qt[(int64_t)sg * QS + d];
            const float tau = qt[(int64_t)sg * QS + QT];
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
