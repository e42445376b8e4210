// Shared device helpers for the gfx950 layout engine.  Compiled with
// -ffp-contract=off: every fused multiply-add below is an explicit fmaf, so the
// arithmetic matches the reference's op-by-op fp32 rounding ("Numerics").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GH_WAVE 64
// Neighbour rows a thread of the spring pull has in flight (tuning: -DGH_DEPTH8=.. -DGH_DEPTH16=..)
#ifndef GH_DEPTH8
#define GH_DEPTH8 4
#endif
#ifndef GH_DEPTH16
#define GH_DEPTH16 2
#endif
#define GH_GATHER_DEPTH(LD) ((LD) <= 4 ? 8 : (LD) <= 8 ? GH_DEPTH8 : GH_DEPTH16)

// Row stride (floats) of the padded position array for an embedding dimension D:
// rows are 16-byte aligned so a vertex is fetched with dwordx4 loads.
__host__ __device__ inline int gh_ld(int D) {
    return D <= 4 ? 4 : D <= 8 ? 8 : D <= 16 ? 16 : ((D + 3) & ~3);
}

// torch.norm(x, dim=1) reduction order of the reference's CPU backend (pt.py:623,
// pt.py:731), squared: full groups of 8 -> eight fma lane accumulators summed left
// to right; groups of 4 -> s + x*x; last D%4 -> fma chain.  D <= 3 is a plain fma
// chain.  Bit-identical to oracle/graphem_oracle.c:go_norm2 (before the sqrt).
template <int D>
__device__ __forceinline__ float gh_sumsq(const float *x) {
    float s = 0.0f;
    int d = 0;
    if constexpr (D >= 8) {
        float acc[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) acc[l] = 0.0f;
#pragma unroll
        for (d = 0; d + 8 <= D; d += 8) {
#pragma unroll
            for (int l = 0; l < 8; ++l) acc[l] = fmaf(x[d + l], x[d + l], acc[l]);
        }
        s = acc[0];
#pragma unroll
        for (int l = 1; l < 8; ++l) s = s + acc[l];
        d = D & ~7;
    }
#pragma unroll
    for (; d + 4 <= D; d += 4) {
#pragma unroll
        for (int l = 0; l < 4; ++l) s = s + x[d + l] * x[d + l];
    }
#pragma unroll
    for (; d < D; ++d) s = fmaf(x[d], x[d], s);
    return s;
}

// Same order with a runtime D and x in memory (generic-dimension kernels).
__device__ inline float gh_sumsq_rt(const float *x, int D) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int d = 0;
    for (; d + 8 <= D; d += 8)
        for (int l = 0; l < 8; ++l) acc[l] = fmaf(x[d + l], x[d + l], acc[l]);
    float s = acc[0];
    for (int l = 1; l < 8; ++l) s = s + acc[l];
    for (; d + 4 <= D; d += 4)
        for (int l = 0; l < 4; ++l) s = s + x[d + l] * x[d + l];
    for (; d < D; ++d) s = fmaf(x[d], x[d], s);
    return s;
}

// Load one padded position row (LD floats, 16-byte aligned) with dwordx4 loads.
template <int LD>
__device__ __forceinline__ void gh_load_row(const float *__restrict__ pos, int64_t v, float *out) {
    const float4 *p = reinterpret_cast<const float4 *>(pos + v * LD);
#pragma unroll
    for (int i = 0; i < LD / 4; ++i) {
        const float4 t = p[i];
        out[4 * i + 0] = t.x; out[4 * i + 1] = t.y; out[4 * i + 2] = t.z; out[4 * i + 3] = t.w;
    }
}

template <int LD>
__device__ __forceinline__ void gh_store_row(float *__restrict__ dst, int64_t v, const float *in) {
    float4 *p = reinterpret_cast<float4 *>(dst + v * LD);
#pragma unroll
    for (int i = 0; i < LD / 4; ++i) p[i] = make_float4(in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3]);
}

// (dist2, edge id) packed so that unsigned comparison orders by distance, then id.
// dist2 >= +0 always (fma chain from +0), so its bit pattern is monotone.
__device__ __forceinline__ uint64_t gh_key(float d2, uint32_t id) {
    return (static_cast<uint64_t>(__float_as_uint(d2)) << 32) | id;
}
__device__ __forceinline__ float gh_key_d2(uint64_t k) { return __uint_as_float(static_cast<uint32_t>(k >> 32)); }
__device__ __forceinline__ uint32_t gh_key_id(uint64_t k) { return static_cast<uint32_t>(k); }

#define GH_KEY_INF 0xFFFFFFFFFFFFFFFFull

// Wave-level sum of a double over 64 lanes (result valid in lane 0).
__device__ __forceinline__ double gh_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, GH_WAVE);
    return v;
}

// ---------------------------------------------------------------------------------
// IEEE-correct square root and division with fewer instructions than the general sequences the compiler emits for
// sqrtf() and '/' (17 and 11 per operation: they carry the scaling for denormal operands and results).  The spring phase
// computes, per neighbour, sqrt(sum of squares) and D quotients diff[d] / dist BY THE SAME dist: phase A of the fused
// kernel issued ~63 VALU instructions per neighbour, 50 of them these two operations, and the kernel is 56-66 % VALU-busy.
//   gh_sqrt_ieee   x >= 2^-96 (else sqrtf): v_sqrt_f32 (1 ulp) and the compiler's own correction step -- the candidate one
//                  ulp below / above wins when its residual x - c * s says so.
//   gh_div_by<D>   d in [2^-40, 2^20] and every |n| in [2^-100, d] (else '/'): ONE reciprocal refined once (3 instructions,
//                  shared), then per quotient q = n r; q += (n - d q) r twice (5 instructions) -- the hardware division's
//                  steps without v_div_scale / v_div_fmas / v_div_fixup, which are identities on this domain: quotient and
//                  reciprocal are normal, and the residual n - d q is exact (its last bit is >= 2^-147) so the final
//                  fused step rounds the true quotient.
// tests/test_hip_api.py::test_lean_sqrt_and_division_are_ieee compares both with sqrtf and '/' on 2^32 operand sets
// including the domain edges; the sha1 of the spring forces at 1 M vertices pins them end to end.
__device__ __forceinline__ float gh_sqrt_ieee(float x) {
    if (__builtin_expect(!(x >= 0x1p-96f), 0)) return sqrtf(x);   // tiny, zero, negative, NaN: the general sequence
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = fmaf(-sm, s, x), rp = fmaf(-sp, s, x);
    float out = rm <= 0.0f ? sm : s;
    out = rp > 0.0f ? sp : out;
    return x == INFINITY ? x : out;
}
template <int D>
__device__ __forceinline__ void gh_div_by(const float (&n)[D], float d, float (&q)[D]) {
    uint32_t emin = 255;
#pragma unroll
    for (int i = 0; i < D; ++i) emin = min(emin, (__float_as_uint(n[i]) >> 23) & 0xFFu);   // (zero and denormal: exponent 0)
    if (__builtin_expect(!(d >= 0x1p-40f && d <= 0x1p20f) || emin < 27u, 0)) {
#pragma unroll
        for (int i = 0; i < D; ++i) q[i] = n[i] / d;
        return;
    }
    float r = __builtin_amdgcn_rcpf(d);
    r = fmaf(fmaf(-d, r, 1.0f), r, r);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        float t = n[i] * r;
        t = fmaf(fmaf(-d, t, n[i]), r, t);
        q[i] = fmaf(fmaf(-d, t, n[i]), r, t);
    }
}

// ---------------------------------------------------------------------------------
// Spring force on one vertex, pull style (reference pt.py:595-636): the vertex walks its own
// neighbour list, nobody else writes its row, so there are no atomics and the sum is
// reproducible.  The list is stored in the reference's summation order (all edges where the
// vertex is the first endpoint, then all where it is the second, each in edge order: the
// two sequential index_add_ calls of pt.py:633-634).  For a neighbour y of x
//   diff = p_y - p_x,  dist = |diff| + 1e-6,  f = (-k_attr * (dist - L_min)) * (diff / dist)
// which equals +f of pt.py:629 when x is the first endpoint and -f when it is the second,
// bit for bit (negation commutes with every rounding involved).
template <int D, int LD, bool WRITE_MID>
__device__ __forceinline__ void spring_pull(const float *__restrict__ pos, const int32_t *__restrict__ adj,
                                            int beg, int end, int64_t self, const float *px, float L_min,
                                            float neg_k, float *F, float *__restrict__ mid, int64_t mid_row0) {
    // Neighbours are fetched C at a time so C independent row gathers are in flight per lane;
    // the forces are still accumulated strictly in list order.  Bit 31 of a list entry says that
    // this vertex OWNS the edge to that neighbour: the midpoint (pt.py:785) of an owned edge costs
    // nothing here, both endpoints being in registers, and spares the KNN scan its own random
    // gathers; owned midpoints go to consecutive rows of `mid` starting at mid_row0.
    constexpr int C = GH_GATHER_DEPTH(LD);
#pragma unroll
    for (int d = 0; d < LD; ++d) F[d] = 0.0f;
    for (int base = beg; base < end; base += C) {
        int64_t ys[C];
        bool own[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            // past the end of the list: the last entry again (an unconditional load; its row is fetched a second time from
            // the cache and ignored) -- a predicated load per slot was an exec-mask branch per slot
            const uint32_t a = (uint32_t)adj[min(base + j, end - 1)];
            ys[j] = (int64_t)(a & 0x7FFFFFFFu);
            own[j] = base + j < end && (a >> 31) != 0;
        }
        float py[C][LD];
#pragma unroll
        for (int j = 0; j < C; ++j) gh_load_row<LD>(pos, ys[j], py[j]);
#pragma unroll
        for (int j = 0; j < C; ++j) {
            if (base + j < end) {
                float diff[D];
#pragma unroll
                for (int d = 0; d < D; ++d) diff[d] = py[j][d] - px[d];
                const float dist = gh_sqrt_ieee(gh_sumsq<D>(diff)) + 1e-6f;
                const float fm = neg_k * (dist - L_min);
                float quot[D];
                gh_div_by<D>(diff, dist, quot);   // diff[d] / dist, IEEE-correct, one reciprocal for all D
#pragma unroll
                for (int d = 0; d < D; ++d) F[d] = F[d] + fm * quot[d];
                if (WRITE_MID && own[j]) {
                    float mrow[LD];
#pragma unroll
                    for (int d = 0; d < LD; ++d) mrow[d] = d < D ? (px[d] + py[j][d]) / 2.0f : 0.0f;
                    gh_store_row<LD>(mid, mid_row0++, mrow);
                }
            }
        }
    }
}


// Embedding dimensions with compile-time kernels (spring pull, fused spring+scan, hub sums, intersection
// pairs): every D from 2 to 16, so that e.g. n_components = 6 does not fall onto the generic
// one-thread-per-vertex kernels (2.4-2.8x slower at 1M vertices); larger D use those.  The norm order (gh_sumsq<D>) is the reference's for every D, so each D is its own
// instantiation rather than a padded neighbour.
__host__ __device__ inline bool gh_dim_templated(int D) {
    return D >= 2 && D <= 16;
}
// X(D, LD) for every templated dimension
#define GH_FOR_EACH_DIM(X)                                                                                    \
    X(2, 4) X(3, 4) X(4, 4) X(5, 8) X(6, 8) X(7, 8) X(8, 8) X(9, 16) X(10, 16) X(11, 16) X(12, 16) X(13, 16) X(14, 16) \
        X(15, 16) X(16, 16)

// ---------------------------------------------------------------------------------
// Long rows (hubs).  A thread walking a pull list of thousands of entries eight gathers at a time
// holds its whole workgroup up for milliseconds, so rows with more than GH_LONG_DEG neighbours get
// their spring force from spring_long_kernel (forces.hip) beforehand: one wave per row, 64
// neighbours gathered and their force terms computed in parallel, then ADDED IN LIST ORDER through
// v_readlane -- the reference's summation order, bit for bit.  Such rows own no edge whose other
// endpoint is short (api.hip: the short endpoint owns it); the few they do own (hub-hub edges) are
// listed apart so that their midpoints can still be emitted by the row's thread.
#define GH_LONG_DEG 128
// Small dense graphs (a SNAP social graph at 16 components: 4 K vertices of mean degree 44) leave a fused workgroup with
// two dozen vertices for 256 threads, each walking its list two 64-byte rows at a time -- 40 us of one wave's serial
// work.  There every row above GH_LONG_DEG_DENSE takes the long path (one thread per list entry, then the ordered sum).
#define GH_LONG_DEG_DENSE 16
__host__ inline int gh_long_degree(int64_t n, int64_t E) {
    return (E <= (int64_t)1 << 20 && 2 * E >= 24 * n) ? GH_LONG_DEG_DENSE : GH_LONG_DEG;   // mean degree >= 24, <= 1M edges
}
struct gh_long_args {
    const int32_t *rows;    // local ids of the long own rows, ascending
    const int32_t *ownptr;  // (n + 1) offsets into ownadj
    const int32_t *ownadj;  // neighbours across the edges the long rows own, pull-list order
    int n;                  // number of long own rows (0: the graph has none)
    int deg;                // rows with more neighbours than this are long (gh_long_degree)
    // fused kernels: the midpoints of the edges long rows own are made by the whole workgroup, one owned edge per thread
    // (gh_long_midpoints), not by the row's thread walking its list
    const uint8_t *own_long;   // per owned-edge slot: its owner row is long; null: the row's thread emits them (unfused kernels)
    const int32_t *own_eids;   // ids of the owned edges in slot order
    const int32_t *edges;      // (E, 2)
};

// Midpoints of the tile's owned edges whose owner row is long -> mid rows (slot j of the tile = owned edge fe0 + j).
// (p_u + p_v) / 2 is the same number whichever endpoint owns the edge.
template <int D, int LD, int NT>
__device__ __forceinline__ void gh_long_midpoints(const float *__restrict__ pos, const gh_long_args &la, int fe0, int nedges,
                                                  float *__restrict__ mid) {
    for (int j = threadIdx.x; j < nedges; j += NT) {
        if (!la.own_long[fe0 + j]) continue;
        const int2 uv = reinterpret_cast<const int2 *>(la.edges)[la.own_eids[fe0 + j]];
        float pu[LD], pv[LD], mrow[LD];
        gh_load_row<LD>(pos, uv.x, pu);
        gh_load_row<LD>(pos, uv.y, pv);
#pragma unroll
        for (int d = 0; d < LD; ++d) mrow[d] = d < D ? (pu[d] + pv[d]) / 2.0f : 0.0f;
        gh_store_row<LD>(mid, j, mrow);
    }
}

// One row of the spring phase: short rows pull, long rows take the force spring_long_kernel left
// in Fpre and only emit the midpoints of the edges they own.
template <int D, int LD, bool WRITE_MID>
__device__ __forceinline__ void spring_row(const float *__restrict__ pos, const int32_t *__restrict__ adj, int beg,
                                           int end, int64_t self, const float *px, float L_min, float neg_k, float *F,
                                           float *__restrict__ mid, int64_t mid_row0, const gh_long_args &la,
                                           int i_local, const float *__restrict__ Fpre) {
    if (la.n > 0 && end - beg > la.deg) {
        gh_load_row<LD>(Fpre, 0, F);
        if (WRITE_MID && !la.own_long) {
            int lo = 0, hi = la.n - 1;
            while (lo < hi) {  // la.rows holds i_local
                const int m = (lo + hi) >> 1;
                if (la.rows[m] < i_local) lo = m + 1; else hi = m;
            }
            constexpr int C = GH_GATHER_DEPTH(LD);   // neighbour rows in flight, as in spring_pull
            const int jend = la.ownptr[lo + 1];
            for (int base = la.ownptr[lo]; base < jend; base += C) {
                float py[C][LD];
#pragma unroll
                for (int j = 0; j < C; ++j) gh_load_row<LD>(pos, base + j < jend ? la.ownadj[base + j] : (int32_t)self, py[j]);
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    if (base + j < jend) {
                        float mrow[LD];
#pragma unroll
                        for (int d = 0; d < LD; ++d) mrow[d] = d < D ? (px[d] + py[j][d]) / 2.0f : 0.0f;
                        gh_store_row<LD>(mid, mid_row0++, mrow);
                    }
                }
            }
        }
        return;
    }
    spring_pull<D, LD, WRITE_MID>(pos, adj, beg, end, self, px, L_min, neg_k, F, mid, mid_row0);
}

// ---------------------------------------------------------------------------------
// Device sampler: the t-th value of a keyed pseudo-random permutation of [0, E) (stands in for
// torch.randperm(E)[t], pt.py:409).  A 4-round Feistel network on ceil(log2 E) bits with cycle
// walking: every thread computes its own id, ids are distinct by construction, and every rank
// gets the same ids for the same (seed, iteration).
__device__ __forceinline__ uint32_t gh_mix32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (uint32_t)x;
}
__device__ inline int32_t gh_sample_id(int64_t E, uint64_t seed, uint64_t iter, int64_t t) {
    int bits = 1;
    while (((int64_t)1 << bits) < E) ++bits;
    const int lb = bits / 2, hb = bits - lb;  // low / high half widths
    const uint64_t lmask = ((uint64_t)1 << lb) - 1, hmask = ((uint64_t)1 << hb) - 1;
    const uint64_t key = seed * 0x9E3779B97F4A7C15ull + iter * 0xD1B54A32D192ED03ull + 0x2545F4914F6CDD1Dull;
    uint64_t x = (uint64_t)t;
    do {
        uint64_t lo = x & lmask, hi = (x >> lb) & hmask;
        for (int rnd = 0; rnd < 4; ++rnd) {
            // alternate which half is modified so unequal widths stay a bijection
            if ((rnd & 1) == 0) hi = (hi ^ gh_mix32(key + ((uint64_t)rnd << 56) + lo)) & hmask;
            else lo = (lo ^ gh_mix32(key + ((uint64_t)rnd << 56) + hi)) & lmask;
        }
        x = (hi << lb) | lo;
    } while ((int64_t)x >= E);
    return (int32_t)x;
}
