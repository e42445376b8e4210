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

// R reference midpoints per thread, held as R/2 packed pairs so the distance arithmetic runs on
// v_pk_add/mul/fma_f32 (two references per VALU instruction: measured, a plain fp32 VALU op
// occupies a SIMD for 4 cycles, so packed math is the only way past half of the fp32 vector
// peak).  The query group streams past as broadcast ds_read_b128, prefetched one query ahead.
// Per pair: D/2 sub, 1/2 mul, (D-1)/2 fma; per query one min-tree over the R distances, ONE
// compare and ONE branch.  Hits are parked in LDS (hkey/hq/hcount): the global returning atomic
// that reserves a list slot costs a ~1.5 us round trip and must not sit inside this loop.
template <int D, int R, int HITBUF>
__device__ __forceinline__ void gh_scan_queries(const gh_f2 (&m)[R / 2][D], const uint32_t (&id)[R],
                                                const float4 *qsh, int nq, int s_lo, uint64_t *hkey, int *hq,
                                                int *hcount, uint64_t *__restrict__ cand,
                                                int32_t *__restrict__ cnt) {
    constexpr int LD = D <= 4 ? 4 : D <= 8 ? 8 : 16;
    constexpr int QS = D <= 3 ? 4 : LD + 4;
    constexpr int QT = D <= 3 ? 3 : LD;
    float4 rec[QS / 4], nxt[QS / 4];
#pragma unroll
    for (int i = 0; i < QS / 4; ++i) nxt[i] = qsh[i];
    for (int s = 0; s < nq; ++s) {
#pragma unroll
        for (int i = 0; i < QS / 4; ++i) rec[i] = nxt[i];
#pragma unroll
        for (int i = 0; i < QS / 4; ++i) nxt[i] = qsh[(s + 1) * (QS / 4) + i];  // broadcast read, next query
        const float *qv = reinterpret_cast<const float *>(rec);
        const float tau = qv[QT];
        gh_f2 d2[R / 2];
#pragma unroll
        for (int r = 0; r < R / 2; ++r) {
            const gh_f2 t0 = (gh_f2){qv[0], qv[0]} - m[r][0];
            gh_f2 acc = t0 * t0;  // == fma(t0, t0, +0)
#pragma unroll
            for (int d = 1; d < D; ++d) {
                const gh_f2 td = (gh_f2){qv[d], qv[d]} - m[r][d];
                acc = __builtin_elementwise_fma(td, td, acc);
            }
            d2[r] = acc;
        }
        float dmin = fminf(d2[0].x, d2[0].y);
#pragma unroll
        for (int r = 1; r < R / 2; ++r) dmin = fminf(dmin, fminf(d2[r].x, d2[r].y));
        if (dmin <= tau) {  // rare: some reference of this thread is a candidate
            const int sg = s_lo + s;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float dr = (r & 1) ? d2[r / 2].y : d2[r / 2].x;
                if (dr <= tau) {
                    const int p = atomicAdd(hcount, 1);
                    if (p < HITBUF) { hkey[p] = gh_key(dr, id[r]); hq[p] = sg; }
                    else gh_append_candidate(cand, cnt, sg, gh_key(dr, id[r]));
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
