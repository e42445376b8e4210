// GH_DIST_CDIST: neighbour rows as the reference's PyTorch-CPU backend gets them from torch.cdist + torch.topk
// (pt.py:580-583), value for value and tie for tie.  Included at the end of knn.hip (it uses that file's K-smallest
// extraction).
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
//   2. knn_select_cdist_kernel re-values the few hundred candidates with the formula above (two row gathers each),
//      extracts the K + 1 smallest (value, id) keys and PROVES the list complete for cdist's ranking: an edge outside
//      has exact distance > tau, hence acc > tau (1 - (7D + 15) u) - 3 (3D + 6) u |q|^2, u = 2^-24 (derivation at
//      cdist_lower_bound); if the (K+1)-th smallest value clears that bound and the K + 1 values are pairwise
//      different, the row is decided: ascending values, no heap order involved;
//   3. the other rows -- a tie among the K + 1 smallest values (about one row in a hundred at a million vertices), or a
//      list that could not be proven complete (queries far outside the bulk, where the quantum exceeds the neighbour
//      spacing) -- and EVERY row of a graph too small for the scan take the full pass: cdist_rows_kernel values all E
//      edges for them (chunks of edge ids, chunk minima on the side), cdist_sim_kernel replays partial_sort's
//      heap in index order, one wave per row, visiting only the chunks whose minimum is below the heap's maximum
//      (an element that does not beat the maximum leaves the heap untouched, so skipping it changes nothing).
//   Rows of graphs with K * 64 > E (tiny ones) are ranked by std::nth_element + std::sort in ATen; their values are
//   computed as above and equal values ordered by id, counted in gh_knn_cdist_stats when a tie is present.
#pragma once

namespace {

#define GH_CD_TILE 4096   /* chunk minima staged in LDS at a time (cdist_sim_kernel) */
struct cdist_args {
    const float *pos;          // (n, LD) rows
    const int32_t *edges;      // (E, 2)
    int D, LD;
    int64_t E;
    const float *qt;           // query records (knn.hip gh_qs): coordinates, then tau
    int QS, QT;
    int mm_form;               // ATen's matmul form (S > 25 or E > 25), else its direct kernel: sqrt of the exact-difference sum
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

// One workgroup per query: candidate list (exact d2 <= tau) -> the K smallest cdist keys, when the list proves enough.
template <int LDT>
__global__ __launch_bounds__(256) void knn_select_cdist_kernel(uint64_t *__restrict__ cand, int32_t *__restrict__ cnt, int K,
                                                               cdist_args a, uint64_t *__restrict__ out_keys,
                                                               int32_t *__restrict__ ovf, int32_t *__restrict__ dbg_cnt,
                                                               int32_t *__restrict__ rare) {
    __shared__ uint64_t best[GH_EXTRACT_MAX_K];
    __shared__ uint64_t best2[GH_EXTRACT_MAX_K];
    __shared__ uint64_t red[4 * GH_EXTRACT_MAX_K];
    __shared__ float qs[16];
    const int Ks = K + 1;
    const int64_t qi = blockIdx.x;
    const int c = cnt[qi * GH_CNT_STRIDE];
    if (threadIdx.x < 16) qs[threadIdx.x] = (int)threadIdx.x < a.D ? a.qt[qi * a.QS + threadIdx.x] : 0.0f;
    __syncthreads();
    if (threadIdx.x == 0) { cnt[qi * GH_CNT_STRIDE] = 0; dbg_cnt[qi] = c; }
    int reason = (c > GH_CAND_CAP || c < Ks) ? 2 : 0;   // the list overflowed, or tau was not a bound for K + 1 edges
    if (!reason) {
        const float qn = cdist_norm(qs, a.D);
        uint64_t *list = cand + qi * GH_CAND_CAP;
        for (int i = threadIdx.x; i < c; i += 256) {
            const uint32_t id = gh_key_id(list[i]);
            list[i] = gh_key(cdist_pair<LDT>(a, qs, qn, id), id);
        }
        __syncthreads();
        block_extract_list(list, c, Ks, best, best2, red);
        int bad = 0;
        for (int i = threadIdx.x; i + 1 < Ks; i += 256) bad |= (uint32_t)(best[i] >> 32) == (uint32_t)(best[i + 1] >> 32) ? 1 : 0;
        if (threadIdx.x == 0) {
            const double w = (double)gh_key_d2(best[Ks - 1]);   // the (K+1)-th smallest VALUE (a distance, not squared)
            if (!(w * w * (1.0 + 1e-6) <= cdist_lower_bound(a.qt[qi * a.QS + a.QT], qn, a.D))) bad |= 2;
        }
        // 1: a tie among the K + 1 smallest values, 2: the list is not provably complete (3: both)
        reason = (__syncthreads_or(bad & 1) ? 1 : 0) | (__syncthreads_or(bad & 2) ? 2 : 0);
    }
    if (reason) {
        if (threadIdx.x == 0) { rare[1 + atomicAdd(&rare[0], 1)] = (int32_t)qi; ovf[qi] = reason; }
        return;
    }
    for (int i = threadIdx.x; i < K; i += 256) out_keys[qi * K + i] = best[i];
}

// ---- the full pass -----------------------------------------------------------------------------------------------
// Values of the listed queries against ALL edges, in edge-id order: workgroup x takes the chunk of CH = 256 * NPT ids
// [x * CH, (x + 1) * CH), gathers its midpoints once (registers) and loops over the rows of this round
// (t = r0 + blockIdx.y, step gridDim.y); vbuf[slot][e] = value, cmin[slot][c] = the minimum over ids [256 c, 256 (c + 1)).
template <int LDT, int NPT>
__global__ __launch_bounds__(256) void cdist_rows_kernel(cdist_args a, const int32_t *__restrict__ rare, int all_rows, int r0, int R,
                                                         float *__restrict__ vbuf, int64_t vstride, float *__restrict__ cmin,
                                                         int nchunks) {
    const int nrare = all_rows ? all_rows : rare[0];
    const int t_end = min(nrare, r0 + R);
    if (r0 + (int)blockIdx.y >= t_end) return;
    constexpr int CH = 256 * NPT;
    constexpr int LM = LDT > 0 ? LDT : 1;
    __shared__ float qsh[LDT > 0 ? 16 : 1];
    __shared__ float wmin[4][NPT];
    const int64_t e0 = (int64_t)blockIdx.x * CH;
    float m[NPT][LM], mn[NPT];
    if constexpr (LDT > 0) {   // midpoints of this thread's ids, and their norms
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
    for (int t = r0 + (int)blockIdx.y; t < t_end; t += (int)gridDim.y) {
        const int64_t qi = all_rows ? t : rare[1 + t];
        const int slot = t - r0;
        float vmin[NPT];
        if constexpr (LDT > 0) {
            __syncthreads();
            if (threadIdx.x < 16) qsh[threadIdx.x] = (int)threadIdx.x < a.D ? a.qt[qi * a.QS + threadIdx.x] : 0.0f;
            __syncthreads();
            float q[LDT];
#pragma unroll
            for (int d = 0; d < LDT; ++d) q[d] = d < 16 ? qsh[d] : 0.0f;
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
                const float v = e < a.E ? sqrtf(fmaxf(acc, 0.0f)) + 0.0f : INFINITY;
                vbuf[(int64_t)slot * vstride + e] = v;
                vmin[j] = v;
            }
        } else {   // any dimension: rows read from memory per pair
            const float *q = a.qt + qi * a.QS;
            const float qn = cdist_norm(q, a.D);
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                const int64_t e = e0 + j * 256 + threadIdx.x;
                const float v = e < a.E ? cdist_pair<0>(a, q, qn, e) : INFINITY;
                vbuf[(int64_t)slot * vstride + e] = v;
                vmin[j] = v;
            }
        }
        // minima per 256 consecutive ids (round j of this workgroup): the granularity cdist_sim_kernel skips at
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) vmin[j] = fminf(vmin[j], __shfl_xor(vmin[j], off, 64));
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int j = 0; j < NPT; ++j) wmin[threadIdx.x >> 6][j] = vmin[j];
        }
        __syncthreads();
        if (threadIdx.x < NPT && (int)blockIdx.x * NPT + (int)threadIdx.x < nchunks)
            cmin[(int64_t)slot * nchunks + blockIdx.x * NPT + threadIdx.x] =
                fminf(fminf(wmin[0][threadIdx.x], wmin[1][threadIdx.x]), fminf(wmin[2][threadIdx.x], wmin[3][threadIdx.x]));
    }
}

// The comparator of ATen's topk on values (NaN last): "x before y".
__device__ __forceinline__ bool cdist_less(float x, float y) { return (x == x && y != y) || x < y; }

// Max-heap with the tie behaviour of the textbook sift-down-to-a-leaf-then-up form std::partial_sort is built on: the
// hole at `hole` sinks to the bottom along the larger child (the RIGHT one unless it is smaller than the left), then
// `val` climbs while its parent is smaller.  Two homes for the heap, one algorithm (adjust() below):
//   cdist_heap_reg  K <= 64: element i in lane i's registers, reads through v_readlane, writes by lane predicate --
//                   every step of a sift is a few scalar-pipe cycles.  (The first version kept every heap in LDS with lane 0
//                   walking it: ~12 DEPENDENT LDS round trips per entering element, ~140 entering elements per row:
//                   140 us per row at 100 K vertices, most of the kernel.)
//   cdist_heap_lds  any K: values and ids in LDS, lane 0 walks.
struct cdist_heap_reg {
    float v;
    int32_t id;
    int lane;
    __device__ __forceinline__ float val(int i) const { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), i)); }
    __device__ __forceinline__ int32_t idx(int i) const { return __builtin_amdgcn_readlane(id, i); }
    __device__ __forceinline__ void set(int i, float x, int32_t xi) { if (lane == i) { v = x; id = xi; } }
    __device__ __forceinline__ void sync() const {}
};
struct cdist_heap_lds {
    float *hv;
    int32_t *hid;
    int lane;
    __device__ __forceinline__ float val(int i) const { return hv[i]; }
    __device__ __forceinline__ int32_t idx(int i) const { return hid[i]; }
    __device__ __forceinline__ void set(int i, float x, int32_t xi) { if (lane == 0) { hv[i] = x; hid[i] = xi; } }
    __device__ __forceinline__ void sync() const { __syncthreads(); }   // one wave per workgroup: orders lane 0's writes before everybody's reads
};
template <class H>
__device__ __forceinline__ void cdist_heap_adjust(H &h, int hole, int len, float val, int32_t vid) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (cdist_less(h.val(child), h.val(child - 1))) --child;
        h.set(hole, h.val(child), h.idx(child));
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        h.set(hole, h.val(child - 1), h.idx(child - 1));
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && cdist_less(h.val(parent), val)) {
        h.set(hole, h.val(parent), h.idx(parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    h.set(hole, val, vid);
    h.sync();
}

// One wave per listed query: std::partial_sort(first, first + K, last) over the (value, index) pairs of ALL edges in
// index order, from the values of cdist_rows_kernel.  The first K pairs are heapified; then pair i enters (replacing
// the maximum) iff value_i < maximum -- so a chunk whose minimum is not below the current maximum holds nothing that
// would enter and is skipped; finally the heap is popped into ascending order.  The heap steps are uniform over the
// wave; the lanes fetch and pre-test 64 values at a time.
// nth_form (K * 64 > E): ATen ranks with std::nth_element + std::sort instead; the K smallest values are the same,
// equal values are put in id order here and the row is counted in stat[0] when any tie is present.
template <int NPL /* values per lane and chunk: CH / 64 */, bool REG /* K <= 64: the heap in registers */>
__global__ __launch_bounds__(64) void cdist_sim_kernel(const int32_t *__restrict__ rare, int all_rows, int r0, int R, int64_t E, int K,
                                                       const float *__restrict__ vbuf, int64_t vstride,
                                                       const float *__restrict__ cmin, int nchunks,
                                                       uint64_t *__restrict__ out_keys, int nth_form, int32_t *__restrict__ stat) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr int CH = 64 * NPL;
    const int nrare = all_rows ? all_rows : rare[0];
    const int t = r0 + (int)blockIdx.x;
    if (t >= min(nrare, r0 + R)) return;
    const int64_t qi = all_rows ? t : rare[1 + t];
    const float *v = vbuf + (int64_t)(t - r0) * vstride;
    const float *cm = cmin + (int64_t)(t - r0) * nchunks;
    const int lane = threadIdx.x;
    using heap_t = typename std::conditional<REG, cdist_heap_reg, cdist_heap_lds>::type;
    heap_t h;
    h.lane = lane;
    if constexpr (REG) {
        h.v = lane < K ? v[lane] : INFINITY;
        h.id = lane;
    } else {
        h.hv = reinterpret_cast<float *>(smem_raw);
        h.hid = reinterpret_cast<int32_t *>(h.hv + K);
        for (int i = lane; i < K; i += 64) { h.hv[i] = v[i]; h.hid[i] = i; }
        __syncthreads();
    }
    if (K >= 2)   // std::make_heap
        for (int parent = (K - 2) / 2; parent >= 0; --parent) cdist_heap_adjust(h, parent, K, h.val(parent), h.idx(parent));
    float hmax = h.val(0);
    float eq_out = -1.0f;   // nth_form: a value left outside the heap while equal to its maximum (boundary tie if it stays so)
    // "may hold an element that enters": minimum below the maximum (nth_form: or equal to it, for the boundary-tie count)
    auto live = [&](float mv) { return nth_form ? !cdist_less(hmax, mv) : cdist_less(mv, hmax); };
    // one chunk, its values in x (lane l holds ids base + j * 64 + l): in index order, whatever still beats the maximum enters
    auto process = [&](const float (&x)[NPL], int64_t base) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int64_t e = base + j * 64 + lane;
            const bool in = e >= K && e < E;
            if (nth_form && __ballot(in && x[j] == hmax)) eq_out = hmax;
            unsigned long long mask = __ballot(in && cdist_less(x[j], hmax));
            while (mask) {
                const int l = __builtin_ctzll(mask);
                const float val = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(x[j]), l));
                if (cdist_less(val, hmax)) {   // std::__pop_heap(first, middle, i): the maximum leaves, val sinks in from the root
                    cdist_heap_adjust(h, 0, K, val, (int32_t)(base + j * 64 + l));
                    const float old = hmax;
                    hmax = h.val(0);
                    if (nth_form && old == hmax) eq_out = hmax;   // one of several equal maxima was popped: it now waits outside
                }
                const unsigned long long later = l == 63 ? 0ull : ~((2ull << l) - 1ull);
                mask = __ballot(in && cdist_less(x[j], hmax)) & later;
                if (nth_form && (__ballot(in && x[j] == hmax) & later)) eq_out = hmax;
            }
        }
    };
    // Chunk minima through LDS, GH_CD_TILE at a time (one round of parallel loads instead of a dependent load per 64
    // chunks); the next P live chunks are fetched together and then processed in order, each re-tested against the maximum
    // of its moment.
    constexpr int P = 4;
    __shared__ float cml[GH_CD_TILE];
    for (int tile0 = (K / CH) / GH_CD_TILE * GH_CD_TILE; tile0 < nchunks; tile0 += GH_CD_TILE) {
        const int nt = min(GH_CD_TILE, nchunks - tile0);
        __syncthreads();
        for (int i = lane; i < nt; i += 64) cml[i] = cm[tile0 + i];
        __syncthreads();
        int p = max(0, K / CH - tile0);   // next chunk of the tile to look at
        while (p < nt) {
            int ids[P];
#pragma unroll
            for (int q = 0; q < P; ++q) {
                ids[q] = -1;
                while (p < nt) {
                    const int b = p & ~63;
                    const float mv = b + lane < nt ? cml[b + lane] : INFINITY;
                    const unsigned long long m = __ballot(b + lane < nt && live(mv)) & (~0ull << (p - b));
                    if (!m) { p = b + 64; continue; }
                    ids[q] = b + __builtin_ctzll(m);
                    p = ids[q] + 1;
                    break;
                }
            }
            float x[P][NPL];
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int64_t base = (int64_t)(tile0 + max(ids[q], 0)) * CH;
#pragma unroll
                for (int j = 0; j < NPL; ++j) x[q][j] = ids[q] >= 0 ? v[base + j * 64 + lane] : INFINITY;
            }
#pragma unroll
            for (int q = 0; q < P; ++q)
                if (ids[q] >= 0 && live(cml[ids[q]])) process(x[q], (int64_t)(tile0 + ids[q]) * CH);
        }
    }
    // pop the heap into ascending order (std::sort_heap)
    for (int last = K - 1; last > 0; --last) {
        const float val = h.val(last);
        const int32_t vid = h.idx(last);
        h.set(last, h.val(0), h.idx(0));
        cdist_heap_adjust(h, 0, last, val, vid);
    }
    if (nth_form) {
        bool tie = eq_out == h.val(K - 1);
        for (int i = 0; i + 1 < K; ++i) tie = tie || h.val(i) == h.val(i + 1);
        if (tie) {
            if (lane == 0) atomicAdd(&stat[0], 1);
            for (int i = 1; i < K; ++i) {   // equal values in id order (insertion sort within runs)
                const float val = h.val(i);
                const int32_t vid = h.idx(i);
                int j = i;
                while (j > 0 && h.val(j - 1) == val && h.idx(j - 1) > vid) { h.set(j, h.val(j - 1), h.idx(j - 1)); h.sync(); --j; }
                h.set(j, val, vid);
                h.sync();
            }
        }
    }
    if constexpr (REG) {
        if (lane < K) out_keys[qi * K + lane] = gh_key(h.v, (uint32_t)h.id);
    } else {
        for (int i = lane; i < K; i += 64) out_keys[qi * K + i] = gh_key(h.hv[i], (uint32_t)h.hid[i]);
    }
}

cdist_args make_cdist_args(gh_engine *h) {
    return cdist_args{h->d_pos, h->d_edges, h->D, h->LD, h->E, h->d_q, gh_qs(h->D, h->LD), gh_qtau(h->D, h->LD),
                      (h->S > 25 || h->E > 25) ? 1 : 0};
}

template <typename T>
gh_status cdist_dev_alloc(gh_engine *h, T **p, size_t count) {
    if (hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) {
        *p = nullptr;
        h->err = "hipMalloc failed (GH_DIST_CDIST buffers)";
        return GH_ERR_NOMEM;
    }
    return GH_OK;
}

}  // namespace

// Buffers of the full pass: values of up to cd_R queries against all edges (at most 4 GiB; more listed queries than
// that take further rounds of the two kernels -- every round is launched, an empty one returns at once).
gh_status gh_cdist_alloc(gh_engine *h) {
    if (!h->cdist || h->S == 0 || h->k == 0) return GH_OK;
    h->cd_CH = 256;   // ids per chunk minimum (the single wave of cdist_sim_kernel is instruction-bound: with 1024-id chunks it
                      // tested 16 sub-blocks per live chunk, 1 M vertices: 228 us per iteration for the rows that come here)
    h->cd_nchunks = (int)((h->E + h->cd_CH - 1) / h->cd_CH);
    const int64_t vstride = ((int64_t)h->cd_nchunks + 3) / 4 * 4 * h->cd_CH;   // whole workgroups of cdist_rows_kernel
    int64_t R = ((int64_t)1 << 30) / std::max<int64_t>(vstride, 1);
    if (R < 16) R = 16;
    if (R > h->S) R = h->S;
    h->cd_R = (int)R;
    GH_TRY_ST(cdist_dev_alloc(h, &h->d_rare, (size_t)h->S + 1));
    GH_TRY_ST(cdist_dev_alloc(h, &h->d_cd_vbuf, (size_t)(R * vstride)));
    GH_TRY_ST(cdist_dev_alloc(h, &h->d_cd_cmin, (size_t)(R * h->cd_nchunks)));
    GH_TRY_ST(cdist_dev_alloc(h, &h->d_cd_stat, 4));
    GH_HIP(hipMemsetAsync(h->d_rare, 0, sizeof(int32_t) * ((size_t)h->S + 1), h->stream));
    GH_HIP(hipMemsetAsync(h->d_cd_stat, 0, sizeof(int32_t) * 4, h->stream));
    return GH_OK;
}

// all_rows: the graph is too small for the filtered scan -- every query takes the full pass.  Otherwise the candidate
// lists of the scan are in place.  -> d_partial (S, K): the reference's rows, column 0 included.
gh_status gh_knn_finish_cdist(gh_engine *h, bool all_rows) {
    const cdist_args a = make_cdist_args(h);
    GH_HIP(hipMemsetAsync(h->d_cd_stat, 0, sizeof(int32_t), h->stream));
    if (!all_rows) {
        GH_HIP(hipMemsetAsync(h->d_rare, 0, sizeof(int32_t), h->stream));
        gh_scope t(h, "knn_select_cdist");
#define GH_CSEL(LL) knn_select_cdist_kernel<LL><<<dim3((unsigned)h->S), dim3(256), 0, h->stream>>>( \
        h->d_cand, h->d_cnt, h->K, a, h->d_partial, h->d_ovf, h->d_dbg_cnt + h->S, h->d_rare)
        if (h->LD == 4) GH_CSEL(4); else if (h->LD == 8) GH_CSEL(8); else GH_CSEL(16);
#undef GH_CSEL
        GH_LAUNCH_CHECK();
    }
    const int64_t vstride = ((int64_t)h->cd_nchunks + 3) / 4 * 4 * h->cd_CH;
    const int nth_form = (int64_t)h->K * 64 > h->E ? 1 : 0;
    const int all = all_rows ? (int)h->S : 0;
    const int npt = h->E >= (1 << 16) ? 4 : 1;   // chunks per workgroup of cdist_rows_kernel
    const unsigned nwg = (unsigned)((h->cd_nchunks + npt - 1) / npt);
    unsigned ys = (unsigned)std::max(1, std::min(h->cd_R, 1024 / (int)std::max(nwg, 1u)));
    for (int r0 = 0; r0 < (int)h->S; r0 += h->cd_R) {
        {
            gh_scope t(h, "cdist_rows");
            const dim3 grid(nwg, ys);
#define GH_CROWS(LL, NP) cdist_rows_kernel<LL, NP><<<grid, dim3(256), 0, h->stream>>>(a, h->d_rare, all, r0, h->cd_R, h->d_cd_vbuf, \
                                                                                     vstride, h->d_cd_cmin, h->cd_nchunks)
            if (npt == 4) {
                if (h->LD == 4) GH_CROWS(4, 4); else if (h->LD == 8) GH_CROWS(8, 4); else if (h->LD == 16) GH_CROWS(16, 4); else GH_CROWS(0, 4);
            } else {
                if (h->LD == 4) GH_CROWS(4, 1); else if (h->LD == 8) GH_CROWS(8, 1); else if (h->LD == 16) GH_CROWS(16, 1); else GH_CROWS(0, 1);
            }
#undef GH_CROWS
            GH_LAUNCH_CHECK();
        }
        gh_scope t(h, "cdist_sim");
        const size_t smem = h->K <= 64 ? 0 : (sizeof(float) + sizeof(int32_t)) * (size_t)h->K;
#define GH_CSIM(NPLv, REGv) cdist_sim_kernel<NPLv, REGv><<<dim3((unsigned)h->cd_R), dim3(64), smem, h->stream>>>( \
        h->d_rare, all, r0, h->cd_R, h->E, h->K, h->d_cd_vbuf, vstride, h->d_cd_cmin, h->cd_nchunks, h->d_partial, nth_form, h->d_cd_stat)
        if (h->K <= 64) GH_CSIM(4, true); else GH_CSIM(4, false);
#undef GH_CSIM
        GH_LAUNCH_CHECK();
    }
    h->intersect_done = false;
    h->stats_reduced = false;
    return GH_OK;
}
