#pragma once
// The per-iteration KNN set-up, shared by knn_setup_kernel (knn.hip: reads the positions) and the
// normalise kernels (forces.hip: run the NEXT iteration's set-up in the same launch, taking every
// position they need as normalise(new row) -- the same arithmetic the normalising threads apply, hence
// the same bits -- so it does not wait for them).
//
// Per query (gh_setup_query): sample id of the iteration (given / device sampler / arange,
// pt.py:403-413), query record = midpoint of that edge (pt.py:785, pt.py:410) + tau = inf,
// candidate-list and overflow reset.
//
// On the filtered-scan path the set-up also produces what the threshold tau of every query is taken
// from (gh_setup_block): the squared distances from all S queries to a subset of the own edges (every
// stride-th, M1 of them), reduced to GROUP MINIMA.  A workgroup takes a tile of GH_THR_TILE subset edges
// (two row gathers each, midpoints to LDS) and evaluates the exact distance chain of every query against
// them, keeping the minimum over each group of GH_THR_GSIZE consecutive subset edges ->
// gmin[query][tile * GH_THR_GROUPS + group].  The K-th smallest
// of a query's group minima bounds its K-th smallest distance from above (K different groups hold
// K different edges that close), and with 64 edges per group it is almost always THE K-th smallest of
// the subset; knn_tau_kernel (knn.hip) extracts it.  This replaced a kernel that streamed a
// materialised copy of the subset once per query and selected the exact K-th smallest (30 us at
// 1 M vertices; it was bound by its two K-smallest extractions and its L2 round trips).
#include "common.h"
#include "engine.h"
#include "scan_core.h"

struct gh_setup_args {
    const int32_t *edges;
    int32_t *sampled;        // ids of the iteration (read for mode 0, written otherwise)
    int mode;                // 0 ids already in `sampled`, 1 device sampler, 2 arange
    int64_t E;
    uint64_t seed, iter;
    int64_t S;
    int D, LD;
    float *qt;
    int32_t *cnt, *ovf;
    int64_t e_lo;
    const int32_t *own_eids;
    const int2 *sub_uv;      // (M1) endpoints of the subset edges (static: spares the set-up two dependent loads)
    int64_t M1, stride;      // subset of the own edges the thresholds come from (0 rows: not on the scan path)
    float *gmin;             // (S, Gpad) group minima as float bits
    int64_t Gpad;            // row stride of gmin = 16 * tiles
    int tiles;               // ceil(M1 / 256): workgroups of gh_setup_block
    unsigned *tau_flag;      // thresholds inside the fused launch (tau_core.h): the published-queries counter, zeroed here
    const uint64_t *iter_dev;  // replayed iterations (hipGraph, api.hip): the iteration number lives on the device and
                               // `iter` above is an offset to it; null: `iter` is the number
};
__device__ __forceinline__ uint64_t gh_setup_iter(const gh_setup_args &a) { return a.iter_dev ? *a.iter_dev + a.iter : a.iter; }

#define GH_THR_TILE 128    /* subset edges per set-up workgroup */
#define GH_THR_GSIZE 64    /* subset edges per group */
#define GH_THR_GROUPS (GH_THR_TILE / GH_THR_GSIZE)   /* group minima per tile and query */

// Sample id, query record and list reset of query t (every query exactly once per iteration).
template <class P /* float(int64_t vertex, int d) */>
__device__ __forceinline__ void gh_setup_item(const gh_setup_args &a, int64_t t, P getp) {
    if (t == 0 && a.tau_flag) *a.tau_flag = 0;
    if (t >= a.S) return;
    int32_t e32;
    if (a.mode == 1) { e32 = gh_sample_id(a.E, a.seed, gh_setup_iter(a), t); a.sampled[t] = e32; }
    else if (a.mode == 2) { e32 = (int32_t)t; a.sampled[t] = e32; }
    else e32 = a.sampled[t];
    const int QS = gh_qs(a.D, a.LD);
    const int64_t e = e32;
    const int64_t u = a.edges[2 * e], v = a.edges[2 * e + 1];
    for (int d = 0; d < QS; ++d) a.qt[t * QS + d] = d < a.D ? (getp(u, d) + getp(v, d)) / 2.0f : 0.0f;
    a.qt[t * QS + gh_qtau(a.D, a.LD)] = INFINITY;
    a.cnt[t * GH_CNT_STRIDE] = 0;
    a.ovf[t] = 0;
}

// Minimum over each row of 16 lanes of non-negative float bits (they order like unsigned integers):
// quad swaps, half-row and row mirrors on the DPP datapath; every lane returns its row's minimum.
template <int CTRL>
__device__ __forceinline__ uint32_t gh_dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t gh_row_min_u32(uint32_t v) {
    v = min(v, gh_dpp_u32<0xB1>(v));   // quad_perm [1,0,3,2]
    v = min(v, gh_dpp_u32<0x4E>(v));   // quad_perm [2,3,0,1]
    v = min(v, gh_dpp_u32<0x141>(v));  // row_half_mirror
    v = min(v, gh_dpp_u32<0x140>(v));  // row_mirror
    return v;
}

// Tile `blk` of the threshold subset against all S queries -> gmin (see the header comment).  Whole
// 256-thread workgroup; lds: GH_SETUP_LDS_BYTES.  A LANE OWNS A QUERY (its coordinates stay in
// registers: sample id, the edge, two row gathers), the tile's GH_THR_TILE subset midpoints sit in LDS
// and stream past every lane as broadcast reads: per pair the exact distance chain and one integer min,
// no cross-lane traffic; after every GH_THR_GSIZE references the running minimum is one group minimum.
// (The first version put references on lanes and reduced over them on the DPP datapath: 256 dependent
// LDS-read / DPP chains per wave, 40 us.)  Workgroup 0 also does the per-query set-up items.
#define GH_SETUP_LDS_BYTES (GH_THR_TILE * 16 * 4)
// The work is split at the point where the positions have to be KNOWN: gh_setup_fetch requests the rows -- RAW(vertex, row):
// this lane's first query (sample id -> edge -> two rows) and, for the first GH_THR_TILE lanes, a subset edge of the tile
// (endpoints -> two rows) -- and gh_setup_finish turns them into positions, NORM(raw value, d), and goes on.  Inside a
// normalise launch the rows are the un-normalised new positions and NORM needs mean and std: the fetch is issued BEFORE the
// statistics are read, so that the two chains of dependent cold loads and the statistics' own round trip run together.
template <int LD>
struct gh_setup_rows {
    float qa[LD], qb[LD], ma[LD], mb[LD];   // endpoint rows of the lane's query / of its subset edge
    int32_t e32;
};
template <int LD, class RAW /* void(int64_t vertex, float (&row)[LD]) */>
__device__ __forceinline__ void gh_setup_fetch(const gh_setup_args &a, int blk, RAW raw, gh_setup_rows<LD> &st) {
    const int t = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);  // inside a normalise launch these waves sit among streaming ones: their chains go first
    if (blk == 0 && t == 0 && a.tau_flag) *a.tau_flag = 0;
    st.e32 = 0;
#pragma unroll
    for (int d = 0; d < LD; ++d) { st.qa[d] = 0.0f; st.qb[d] = 0.0f; st.ma[d] = 0.0f; st.mb[d] = 0.0f; }
    // this lane's first query (every workgroup needs all of them; issued before the tile so that both chains of
    // dependent loads -- id -> edge -> rows here, endpoints -> rows below -- are in flight together)
    if (t < a.S) {
        if (a.mode == 1) st.e32 = gh_sample_id(a.E, a.seed, gh_setup_iter(a), t);
        else if (a.mode == 2) st.e32 = (int32_t)t;
        else st.e32 = a.sampled[t];
        const int2 uv = reinterpret_cast<const int2 *>(a.edges)[st.e32];
        raw(uv.x, st.qa);
        raw(uv.y, st.qb);
    }
    if (t < GH_THR_TILE) {
        const int64_t j = (int64_t)blk * GH_THR_TILE + t;
        if (j < a.M1) {
            const int2 uv = a.sub_uv[j];
            raw(uv.x, st.ma);
            raw(uv.y, st.mb);
        }
    }
}

template <int LD, class RAW, class NORM /* float(float raw value, int d) */>
__device__ __forceinline__ void gh_setup_finish(const gh_setup_args &a, int blk, RAW raw, NORM norm, const gh_setup_rows<LD> &st,
                                                unsigned char *lds, unsigned long long *stamps = nullptr /* diagnostic */) {
    // the tile in LDS as PAIRS of subset edges, component-interleaved: pair p, coordinate d -> (m_2p[d], m_2p+1[d]),
    // so that the distance chain of two references runs on packed fp32 instructions (v_pk_add_f32 / v_pk_fma_f32)
    gh_f2 *rsh = reinterpret_cast<gh_f2 *>(lds);     // [GH_THR_TILE / 2][LD]
    const int t = threadIdx.x;
    const uint64_t iter = gh_setup_iter(a);
    auto query = [&](int64_t s, int32_t &e32, float (&q)[LD]) {   // (queries past the first 256)
        if (a.mode == 1) e32 = gh_sample_id(a.E, a.seed, iter, s);
        else if (a.mode == 2) e32 = (int32_t)s;
        else e32 = a.sampled[s];
        const int2 uv = reinterpret_cast<const int2 *>(a.edges)[e32];
        float ra[LD], rb[LD];
        raw(uv.x, ra);
        raw(uv.y, rb);
#pragma unroll
        for (int d = 0; d < LD; ++d) q[d] = d < a.D ? (norm(ra[d], d) + norm(rb[d], d)) / 2.0f : 0.0f;
    };
    float q[LD];
    int32_t e32 = st.e32;
#pragma unroll
    for (int d = 0; d < LD; ++d) q[d] = d < a.D ? (norm(st.qa[d], d) + norm(st.qb[d], d)) / 2.0f : 0.0f;
    if (t < GH_THR_TILE) {
        const int64_t j = (int64_t)blk * GH_THR_TILE + t;
        float m[LD];
#pragma unroll
        for (int d = 0; d < LD; ++d) m[d] = 0.0f;
        if (j < a.M1) {
#pragma unroll
            for (int d = 0; d < LD; ++d)
                if (d < a.D) m[d] = (norm(st.ma[d], d) + norm(st.mb[d], d)) / 2.0f;
        } else {
            m[0] = INFINITY;  // padding slot: (q - inf)^2 = inf for every finite query
        }
        float *slot = reinterpret_cast<float *>(rsh + (t >> 1) * LD) + (t & 1);
#pragma unroll
        for (int d = 0; d < LD; ++d) slot[2 * d] = m[d];
    }
    __syncthreads();
    if (stamps && t == 0) stamps[2] = wall_clock64();
    for (int64_t s0 = 0; s0 < a.S; s0 += 256) {
        const int64_t s = s0 + t;
        if (s >= a.S) continue;   // no barrier below
        if (s0 > 0) query(s, e32, q);
        if (blk == 0) {  // the set-up item of this query (gh_setup_item), from the coordinates at hand
            if (a.mode != 0) a.sampled[s] = e32;
            const int QS = gh_qs(a.D, a.LD);
#pragma unroll
            for (int d = 0; d < LD; ++d)
                if (d < QS) a.qt[s * QS + d] = q[d];  // pad coordinates are 0
            for (int d = LD; d < QS; ++d) a.qt[s * QS + d] = 0.0f;
            a.qt[s * QS + gh_qtau(a.D, a.LD)] = INFINITY;
            a.cnt[s * GH_CNT_STRIDE] = 0;
            a.ovf[s] = 0;
        }
        uint32_t *dst = reinterpret_cast<uint32_t *>(a.gmin) + s * a.Gpad + (int64_t)blk * GH_THR_GROUPS;
        // a few pairs in flight per lane only: fully unrolled, the compiler hoists every LDS read of the tile
        // (512 registers and scratch spills, which also cost the normalising workgroups of the same kernel their occupancy)
        constexpr int UNR = LD == 4 ? 4 : LD == 8 ? 4 : 2;  // (LD = 4: 8 in flight ran the distance loop ~1 us faster, at 90 VGPRs for the whole normalise launch instead of 58)
        uint32_t mn[GH_THR_GROUPS];
#pragma unroll
        for (int gg = 0; gg < GH_THR_GROUPS; ++gg) {
            uint32_t best = 0x7F800000u;
#pragma unroll UNR
            for (int r = 0; r < GH_THR_GSIZE / 2; ++r) {
                const float4 *mrow = reinterpret_cast<const float4 *>(rsh + (gg * (GH_THR_GSIZE / 2) + r) * LD);
                gh_f2 d2 = {0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < LD / 2; ++i) {
                    const float4 mv = mrow[i];  // broadcast read: coordinates 2i, 2i+1 of both references
                    // the exact chain in coordinate order, two references at a time (pad coordinates are 0 on both sides)
                    const gh_f2 da = (gh_f2){q[2 * i], q[2 * i]} - (gh_f2){mv.x, mv.y};
                    d2 = __builtin_elementwise_fma(da, da, d2);
                    const gh_f2 db = (gh_f2){q[2 * i + 1], q[2 * i + 1]} - (gh_f2){mv.z, mv.w};
                    d2 = __builtin_elementwise_fma(db, db, d2);
                }
                best = min(best, min(__float_as_uint(d2.x), __float_as_uint(d2.y)));
            }
            mn[gg] = best;
        }
        static_assert(GH_THR_GROUPS == 2 || GH_THR_GROUPS % 4 == 0, "vector stores of the group minima");
        if constexpr (GH_THR_GROUPS == 2) {
            *reinterpret_cast<uint2 *>(dst) = make_uint2(mn[0], mn[1]);
        } else {
#pragma unroll
            for (int g4 = 0; g4 < GH_THR_GROUPS; g4 += 4)
                *reinterpret_cast<uint4 *>(dst + g4) = make_uint4(mn[g4], mn[g4 + 1], mn[g4 + 2], mn[g4 + 3]);
        }
    }
}

// Both halves at once, positions read through getp(vertex, d) (knn_setup_kernel, the gathered normalise).
template <int LD, class P>
__device__ __forceinline__ void gh_setup_block(const gh_setup_args &a, int blk, P getp, unsigned char *lds,
                                               unsigned long long *stamps = nullptr /* diagnostic */) {
    auto raw = [&](int64_t v, float (&row)[LD]) {
#pragma unroll
        for (int d = 0; d < LD; ++d) row[d] = d < a.D ? getp(v, d) : 0.0f;
    };
    auto norm = [](float x, int) { return x; };
    gh_setup_rows<LD> st;
    gh_setup_fetch<LD>(a, blk, raw, st);
    gh_setup_finish<LD>(a, blk, raw, norm, st, lds, stamps);
}

// Runtime row stride -> the instantiation (the scan path has LD in {4, 8, 16}).
template <class P>
__device__ __forceinline__ void gh_setup_block_any(const gh_setup_args &a, int blk, P getp, unsigned char *lds,
                                                   unsigned long long *stamps = nullptr) {
    if (a.LD == 4) gh_setup_block<4>(a, blk, getp, lds, stamps);
    else if (a.LD == 8) gh_setup_block<8>(a, blk, getp, lds, stamps);
    else gh_setup_block<16>(a, blk, getp, lds, stamps);
}
