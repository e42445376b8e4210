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
                                         float *__restrict__ diff, int32_t own_lo = 0, int32_t own_hi = 0x7FFFFFFF /* a row
                                         partition accumulates only what lands on its own rows [own_lo, own_hi): nothing else is
                                         read by its integrate step, and the double-precision atomics are the phase's cost */) {
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
        if (v[role] < own_lo || v[role] >= own_hi) continue;
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
                                                    int32_t *__restrict__ tcount, int32_t own_lo = 0, int32_t own_hi = 0x7FFFFFFF) {
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
        if (v[role] < own_lo || v[role] >= own_hi) continue;
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

// ---------------------------------------------------------------------------------
// The same phase for one query's k candidate pairs as a WORKGROUP when rows are wide (LD >= 8).  A thread per pair
// issues 4 * D fp64 atomics one after another, each wave instruction touching 64 different rows -- the slowest shape
// an atomic can have (MI355X guide: 17x below the contiguous rate); at 16 components and 32 neighbours that was most of
// a 48 us select launch.  Here: (1) a thread per pair only DECIDES (i < j, no shared vertex, strict crossing on
// coordinates 0 / 1) and lists the crossing pairs in LDS; (2) 4 * LD consecutive lanes take one listed pair, lane =
// (role, coordinate): every lane recomputes centroid, difference and norm of its role with the arithmetic of
// gh_intersect_pair_t -- identical terms -- and adds ITS coordinate: one wave instruction = up to 4 contiguous rows.
struct gh_pair_list {
    int n;
    int nt, base;  // vertices this query touched first, and where their run starts in the global list
    int4 v[128];   // endpoints (p1, p2, q1, q2) of the crossing pairs of one query (k <= 127 on this path)
    int32_t t[512];
};

template <int D, int LD>
__device__ __forceinline__ void gh_intersect_query_wide(const float *__restrict__ pos, const int32_t *__restrict__ edges,
                                                        int32_t i, const uint64_t *best /* K keys, LDS */, int k,
                                                        float k_inter, double *__restrict__ acc,
                                                        int32_t *__restrict__ tflag, int32_t *__restrict__ touched,
                                                        int32_t *__restrict__ tcount, gh_pair_list *pl,
                                                        int32_t *__restrict__ own_count = nullptr /* thousands of queries: `touched`
                                                        is this query's OWN run of 4 k slots and its length goes here -- no atomic on
                                                        the shared counter at all; knn_touched_compact_kernel gathers the runs */,
                                                        int32_t own_lo = 0, int32_t own_hi = 0x7FFFFFFF) {
    if (threadIdx.x == 0) { pl->n = 0; pl->nt = 0; }
    __syncthreads();
    for (int c = threadIdx.x; c < k; c += blockDim.x) {
        const int32_t j = (int32_t)(uint32_t)best[c + 1];   // column 0 is dropped blindly (pt.py:421)
        if (!(i < j)) continue;
        const int2 ei = reinterpret_cast<const int2 *>(edges)[i], ej = reinterpret_cast<const int2 *>(edges)[j];
        if (ei.x == ej.x || ei.x == ej.y || ei.y == ej.x || ei.y == ej.y) continue;
        const float2 a = *reinterpret_cast<const float2 *>(pos + (int64_t)ei.x * LD), b = *reinterpret_cast<const float2 *>(pos + (int64_t)ei.y * LD);
        const float2 p = *reinterpret_cast<const float2 *>(pos + (int64_t)ej.x * LD), q = *reinterpret_cast<const float2 *>(pos + (int64_t)ej.y * LD);
        const float x0[2] = {a.x, a.y}, x1[2] = {b.x, b.y}, x2[2] = {p.x, p.y}, x3[2] = {q.x, q.y};
        const float o1 = gh_orient2d(x0, x1, x2), o2 = gh_orient2d(x0, x1, x3);
        const float o3 = gh_orient2d(x2, x3, x0), o4 = gh_orient2d(x2, x3, x1);
        if (!(o1 * o2 < 0.0f && o3 * o4 < 0.0f)) continue;
        pl->v[atomicAdd(&pl->n, 1)] = make_int4(ei.x, ei.y, ej.x, ej.y);
    }
    __syncthreads();
    constexpr int LPP = 4 * LD;                 // lanes per pair
    const int np = pl->n;
    const int sub = threadIdx.x % LPP, role = sub / LD, d = sub % LD;
    for (int p = threadIdx.x / LPP; p < np; p += blockDim.x / LPP) {
        const int4 vv = pl->v[p];
        const int32_t v[4] = {vv.x, vv.y, vv.z, vv.w};
        float x[4][LD];
#pragma unroll
        for (int r = 0; r < 4; ++r) gh_load_row<LD>(pos, v[r], x[r]);
        float diff[D];
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
            const float cen = (((x[0][dd] + x[1][dd]) + x[2][dd]) + x[3][dd]) / 4.0f;
            const float mine = role == 0 ? x[0][dd] : role == 1 ? x[1][dd] : role == 2 ? x[2][dd] : x[3][dd];
            diff[dd] = mine - cen;
        }
        const float dist = sqrtf(gh_sumsq<D>(diff)) + 1e-6f;
        const float dsq = dist * dist;
        float term = diff[0];
#pragma unroll
        for (int dd = 1; dd < D; ++dd) term = d == dd ? diff[dd] : term;
        const int32_t me = role == 0 ? v[0] : role == 1 ? v[1] : role == 2 ? v[2] : v[3];
        if (me < own_lo || me >= own_hi) continue;   // (a row partition: another rank's row)
        if (d < D) atomicAdd(&acc[(int64_t)me * LD + d], (double)((k_inter * term) / dsq));
        // first touch of a vertex: listed per query in LDS, ONE reservation in the global list per query below (a returning
        // atomic per vertex on the single counter serialised at ~11 ns each: 16 K queries' 30 K first touches were 320 us of
        // this launch, 256 queries' several hundred a third of its 14 us)
        if (d == 0 && atomicExch(&tflag[me], 1) == 0) pl->t[atomicAdd(&pl->nt, 1)] = me;
    }
    __syncthreads();
    const int nt = pl->nt;
    if (own_count) {
        if (threadIdx.x == 0) *own_count = nt;
        for (int i = threadIdx.x; i < nt; i += blockDim.x) touched[i] = pl->t[i];
        return;
    }
    if (nt == 0) return;   // uniform: read behind the barrier
    if (threadIdx.x == 0) pl->base = atomicAdd(tcount, nt);
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += blockDim.x) touched[pl->base + i] = pl->t[i];
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
