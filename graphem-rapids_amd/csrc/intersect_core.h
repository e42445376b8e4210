#pragma once
// Intersection repulsion of ONE candidate pair (reference pt.py:638-774), shared by the stand-alone
// intersect kernel (forces.hip) and the KNN kernels that finish a query and process its k pairs in
// the same launch (knn.hip).
#include "common.h"

__device__ __forceinline__ float gh_orient2d(const float *a, const float *b, const float *c) {
    return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]);
}

// Pair (i = sampled edge, j = neighbour edge): keep i < j (pt.py:672), drop pairs sharing a vertex
// (pt.py:685-692), keep pairs whose projections on coordinates 0,1 strictly cross (pt.py:760-772);
// then each of the four endpoints x gets
//   k_inter * (x - c) / (|x - c| + 1e-6)^2,   c = (((p1 + p2) + q1) + q2) / 4      (pt.py:722-734).
// Contributions are summed with fp64 atomics: with the handful of terms a vertex receives the fp64
// sum is exact, hence independent of arrival order.  diff = per-pair scratch row of LD floats.
__device__ inline void gh_intersect_pair(const float *__restrict__ pos, int D, int LD,
                                         const int32_t *__restrict__ edges, int32_t i, int32_t j, float k_inter,
                                         double *__restrict__ acc, int32_t *__restrict__ tflag,
                                         int32_t *__restrict__ touched, int32_t *__restrict__ tcount,
                                         float *__restrict__ diff) {
    if (!(i < j)) return;
    const int32_t v[4] = {edges[2 * (int64_t)i], edges[2 * (int64_t)i + 1], edges[2 * (int64_t)j], edges[2 * (int64_t)j + 1]};
    if (v[0] == v[2] || v[0] == v[3] || v[1] == v[2] || v[1] == v[3]) return;
    if (D < 2) return;
    const float *p1 = pos + (int64_t)v[0] * LD, *p2 = pos + (int64_t)v[1] * LD;
    const float *q1 = pos + (int64_t)v[2] * LD, *q2 = pos + (int64_t)v[3] * LD;
    const float o1 = gh_orient2d(p1, p2, q1), o2 = gh_orient2d(p1, p2, q2);
    const float o3 = gh_orient2d(q1, q2, p1), o4 = gh_orient2d(q1, q2, p2);
    if (!(o1 * o2 < 0.0f && o3 * o4 < 0.0f)) return;
    for (int role = 0; role < 4; ++role) {
        const float *x = pos + (int64_t)v[role] * LD;
        for (int d = 0; d < D; ++d) {
            const float cen = (((p1[d] + p2[d]) + q1[d]) + q2[d]) / 4.0f;
            diff[d] = x[d] - cen;
        }
        const float dist = sqrtf(gh_sumsq_rt(diff, D)) + 1e-6f;
        const float dd = dist * dist;
        for (int d = 0; d < D; ++d) atomicAdd(&acc[(int64_t)v[role] * LD + d], (double)((k_inter * diff[d]) / dd));
        if (atomicExch(&tflag[v[role]], 1) == 0) touched[atomicAdd(tcount, 1)] = v[role];
    }
}

// The same pair with everything in registers (D known at compile time: four vector row loads, no
// scratch round trips through memory).  Arithmetic and order as above, so results are identical.
template <int D, int LD>
__device__ __forceinline__ void gh_intersect_pair_t(const float *__restrict__ pos, const int32_t *__restrict__ edges,
                                                    int32_t i, int32_t j, float k_inter, double *__restrict__ acc,
                                                    int32_t *__restrict__ tflag, int32_t *__restrict__ touched,
                                                    int32_t *__restrict__ tcount) {
    if (!(i < j)) return;
    const int2 ei = reinterpret_cast<const int2 *>(edges)[i], ej = reinterpret_cast<const int2 *>(edges)[j];
    const int32_t v[4] = {ei.x, ei.y, ej.x, ej.y};
    if (v[0] == v[2] || v[0] == v[3] || v[1] == v[2] || v[1] == v[3]) return;
    float x[4][LD];
#pragma unroll
    for (int r = 0; r < 4; ++r) gh_load_row<LD>(pos, v[r], x[r]);
    const float o1 = gh_orient2d(x[0], x[1], x[2]), o2 = gh_orient2d(x[0], x[1], x[3]);
    const float o3 = gh_orient2d(x[2], x[3], x[0]), o4 = gh_orient2d(x[2], x[3], x[1]);
    if (!(o1 * o2 < 0.0f && o3 * o4 < 0.0f)) return;
    float cen[D];
#pragma unroll
    for (int d = 0; d < D; ++d) cen[d] = (((x[0][d] + x[1][d]) + x[2][d]) + x[3][d]) / 4.0f;
#pragma unroll
    for (int role = 0; role < 4; ++role) {
        float diff[D];
#pragma unroll
        for (int d = 0; d < D; ++d) diff[d] = x[role][d] - cen[d];
        const float dist = sqrtf(gh_sumsq<D>(diff)) + 1e-6f;
        const float dd = dist * dist;
#pragma unroll
        for (int d = 0; d < D; ++d) atomicAdd(&acc[(int64_t)v[role] * LD + d], (double)((k_inter * diff[d]) / dd));
        if (atomicExch(&tflag[v[role]], 1) == 0) touched[atomicAdd(tcount, 1)] = v[role];
    }
}

// Dispatch on the embedding dimension: register form for the usual D, scratch form otherwise.
__device__ __forceinline__ void gh_intersect_pair_any(const float *__restrict__ pos, int D, int LD,
                                                      const int32_t *__restrict__ edges, int32_t i, int32_t j,
                                                      float k_inter, double *__restrict__ acc,
                                                      int32_t *__restrict__ tflag, int32_t *__restrict__ touched,
                                                      int32_t *__restrict__ tcount, float *__restrict__ diff) {
#define GH_INTERSECT_ONE(DD, LL) \
    case DD: gh_intersect_pair_t<DD, LL>(pos, edges, i, j, k_inter, acc, tflag, touched, tcount); break;
    switch (D) {
        GH_FOR_EACH_DIM(GH_INTERSECT_ONE)
        default: gh_intersect_pair(pos, D, LD, edges, i, j, k_inter, acc, tflag, touched, tcount, diff);
    }
#undef GH_INTERSECT_ONE
}
