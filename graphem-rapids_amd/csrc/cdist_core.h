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
    __shared__ uint64_t red[4 * GH_EXTRACT_MAX_K];
    __shared__ float qs[16];
    constexpr int NPT = GH_CAND_CAP / 256;
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
        block_extract_adaptive<NPT>(list, c, Ks, best, red);
        int bad = 0;
        for (int i = threadIdx.x; i + 1 < Ks; i += 256) bad |= (uint32_t)(best[i] >> 32) == (uint32_t)(best[i + 1] >> 32) ? 1 : 0;
        if (threadIdx.x == 0) {
            const double w = (double)gh_key_d2(best[Ks - 1]);   // the (K+1)-th smallest VALUE (a distance, not squared)
            if (!(w * w * (1.0 + 1e-6) <= cdist_lower_bound(a.qt[qi * a.QS + a.QT], qn, a.D))) bad |= 2;
        }
        reason = __syncthreads_or(bad);
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
// (t = r0 + blockIdx.y, step gridDim.y); vbuf[slot][e] = value, cmin[slot][chunk] = the chunk's minimum.
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
    __shared__ float wmin[4];
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
        float vmin = INFINITY;
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
                vmin = fminf(vmin, v);
            }
        } else {   // any dimension: rows read from memory per pair
            const float *q = a.qt + qi * a.QS;
            const float qn = cdist_norm(q, a.D);
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                const int64_t e = e0 + j * 256 + threadIdx.x;
                const float v = e < a.E ? cdist_pair<0>(a, q, qn, e) : INFINITY;
                vbuf[(int64_t)slot * vstride + e] = v;
                vmin = fminf(vmin, v);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmin = fminf(vmin, __shfl_xor(vmin, off, 64));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) wmin[threadIdx.x >> 6] = vmin;
        __syncthreads();
        if (threadIdx.x == 0) cmin[(int64_t)slot * nchunks + blockIdx.x] = fminf(fminf(wmin[0], wmin[1]), fminf(wmin[2], wmin[3]));
    }
}

// The comparator of ATen's topk on values (NaN last): "x before y".
__device__ __forceinline__ bool cdist_less(float x, float y) { return (x == x && y != y) || x < y; }

// Max-heap primitives with the tie behaviour of the textbook sift-down-to-a-leaf-then-up form std::partial_sort is
// built on: the hole at `hole` sinks to the bottom along the larger child (the RIGHT one unless it is smaller than the
// left), then `val` climbs while its parent is smaller.  hv / hid: values and ids, `len` elements.
__device__ inline void cdist_heap_adjust(float *hv, int32_t *hid, int hole, int len, float val, int32_t vid) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (cdist_less(hv[child], hv[child - 1])) --child;
        hv[hole] = hv[child]; hid[hole] = hid[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        hv[hole] = hv[child - 1]; hid[hole] = hid[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && cdist_less(hv[parent], val)) {
        hv[hole] = hv[parent]; hid[hole] = hid[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    hv[hole] = val; hid[hole] = vid;
}

// One wave per listed query: std::partial_sort(first, first + K, last) over the (value, index) pairs of ALL edges in
// index order, from the values of cdist_rows_kernel.  The first K pairs are heapified; then pair i enters (replacing
// the maximum) iff value_i < maximum -- so a chunk whose minimum is not below the current maximum holds nothing that
// would enter and is skipped; finally the heap is popped into ascending order.  The sequential steps are lane 0's, on
// the heap in LDS; the lanes fetch and pre-test 64 values at a time.
// nth_form (K * 64 > E): ATen ranks with std::nth_element + std::sort instead; the K smallest values are the same,
// equal values are put in id order here and the row is counted in stat[0] when any tie is present.
template <int NPL /* values per lane and chunk: CH / 64 */>
__global__ __launch_bounds__(64) void cdist_sim_kernel(const int32_t *__restrict__ rare, int all_rows, int r0, int R, int64_t E, int K,
                                                       const float *__restrict__ vbuf, int64_t vstride,
                                                       const float *__restrict__ cmin, int nchunks,
                                                       uint64_t *__restrict__ out_keys, int nth_form, int32_t *__restrict__ stat) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float *hv = reinterpret_cast<float *>(smem_raw);
    int32_t *hid = reinterpret_cast<int32_t *>(hv + K);
    constexpr int CH = 64 * NPL;
    const int nrare = all_rows ? all_rows : rare[0];
    const int t = r0 + (int)blockIdx.x;
    if (t >= min(nrare, r0 + R)) return;
    const int64_t qi = all_rows ? t : rare[1 + t];
    const float *v = vbuf + (int64_t)(t - r0) * vstride;
    const float *cm = cmin + (int64_t)(t - r0) * nchunks;
    const int lane = threadIdx.x;
    for (int i = lane; i < K; i += 64) { hv[i] = v[i]; hid[i] = i; }
    __syncthreads();
    if (lane == 0 && K >= 2)
        for (int parent = (K - 2) / 2; parent >= 0; --parent) {
            const float val = hv[parent];
            const int32_t vid = hid[parent];
            cdist_heap_adjust(hv, hid, parent, K, val, vid);
        }
    __syncthreads();
    float hmax = hv[0];
    float eq_out = -1.0f;   // nth_form: a value left outside the heap while equal to its maximum (boundary tie if it stays so)
    for (int c0 = K / CH; c0 < nchunks; c0 += 64) {
        const float mine = c0 + lane < nchunks ? cm[c0 + lane] : INFINITY;
        unsigned long long cmask = __ballot(nth_form ? !cdist_less(hmax, mine) : cdist_less(mine, hmax));
        while (cmask) {
            const int cl = __builtin_ctzll(cmask);
            cmask &= cmask - 1;
            const float cmv = __shfl(mine, cl, 64);
            if (!(nth_form ? !cdist_less(hmax, cmv) : cdist_less(cmv, hmax))) continue;   // the maximum has dropped since
            const int64_t base = (int64_t)(c0 + cl) * CH;
            float x[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) x[j] = v[base + j * 64 + lane];
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const int64_t e = base + j * 64 + lane;
                const bool in = e >= K && e < E;
                if (nth_form) {
                    const unsigned long long eqm = __ballot(in && x[j] == hmax);
                    if (eqm) eq_out = hmax;
                }
                unsigned long long mask = __ballot(in && cdist_less(x[j], hmax));
                while (mask) {
                    const int l = __builtin_ctzll(mask);
                    const float val = __shfl(x[j], l, 64);
                    if (cdist_less(val, hmax)) {
                        if (lane == 0) cdist_heap_adjust(hv, hid, 0, K, val, (int32_t)(base + j * 64 + l));
                        __syncthreads();
                        const float old = hmax;
                        hmax = hv[0];
                        if (nth_form && old == hmax) eq_out = hmax;   // one of several equal maxima was popped: it now waits outside
                        __syncthreads();
                    }
                    const unsigned long long later = l == 63 ? 0ull : ~((2ull << l) - 1ull);
                    mask = __ballot(in && cdist_less(x[j], hmax)) & later;
                    if (nth_form) {
                        const unsigned long long eqm = __ballot(in && x[j] == hmax) & later;
                        if (eqm) eq_out = hmax;
                    }
                }
            }
        }
    }
    // pop the heap into ascending order (std::sort_heap)
    if (lane == 0)
        for (int last = K - 1; last > 0; --last) {
            const float val = hv[last];
            const int32_t vid = hid[last];
            hv[last] = hv[0]; hid[last] = hid[0];
            cdist_heap_adjust(hv, hid, 0, last, val, vid);
        }
    __syncthreads();
    if (nth_form && lane == 0) {
        bool tie = eq_out == hv[K - 1];
        for (int i = 0; i + 1 < K; ++i) tie = tie || hv[i] == hv[i + 1];
        if (tie) {
            atomicAdd(&stat[0], 1);
            for (int i = 1; i < K; ++i) {   // equal values in id order (insertion sort within runs)
                const float val = hv[i];
                const int32_t vid = hid[i];
                int j = i;
                while (j > 0 && hv[j - 1] == val && hid[j - 1] > vid) { hv[j] = hv[j - 1]; hid[j] = hid[j - 1]; --j; }
                hv[j] = val; hid[j] = vid;
            }
        }
    }
    __syncthreads();
    for (int i = lane; i < K; i += 64) out_keys[qi * K + i] = gh_key(hv[i], (uint32_t)hid[i]);
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
    h->cd_CH = h->E >= (1 << 16) ? 1024 : 256;
    h->cd_nchunks = (int)((h->E + h->cd_CH - 1) / h->cd_CH);
    const int64_t vstride = (int64_t)h->cd_nchunks * h->cd_CH;
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
    const int64_t vstride = (int64_t)h->cd_nchunks * h->cd_CH;
    const int nth_form = (int64_t)h->K * 64 > h->E ? 1 : 0;
    const int all = all_rows ? (int)h->S : 0;
    unsigned ys = (unsigned)std::max(1, std::min(h->cd_R, 1024 / std::max(h->cd_nchunks, 1)));
    for (int r0 = 0; r0 < (int)h->S; r0 += h->cd_R) {
        {
            gh_scope t(h, "cdist_rows");
            const dim3 grid((unsigned)h->cd_nchunks, ys);
#define GH_CROWS(LL, NP) cdist_rows_kernel<LL, NP><<<grid, dim3(256), 0, h->stream>>>(a, h->d_rare, all, r0, h->cd_R, h->d_cd_vbuf, \
                                                                                     vstride, h->d_cd_cmin, h->cd_nchunks)
            if (h->cd_CH == 1024) {
                if (h->LD == 4) GH_CROWS(4, 4); else if (h->LD == 8) GH_CROWS(8, 4); else if (h->LD == 16) GH_CROWS(16, 4); else GH_CROWS(0, 4);
            } else {
                if (h->LD == 4) GH_CROWS(4, 1); else if (h->LD == 8) GH_CROWS(8, 1); else if (h->LD == 16) GH_CROWS(16, 1); else GH_CROWS(0, 1);
            }
#undef GH_CROWS
            GH_LAUNCH_CHECK();
        }
        gh_scope t(h, "cdist_sim");
        const size_t smem = (sizeof(float) + sizeof(int32_t)) * (size_t)h->K;
        if (h->cd_CH == 1024)
            cdist_sim_kernel<16><<<dim3((unsigned)h->cd_R), dim3(64), smem, h->stream>>>(h->d_rare, all, r0, h->cd_R, h->E, h->K, h->d_cd_vbuf, vstride,
                                                                                        h->d_cd_cmin, h->cd_nchunks, h->d_partial, nth_form, h->d_cd_stat);
        else
            cdist_sim_kernel<4><<<dim3((unsigned)h->cd_R), dim3(64), smem, h->stream>>>(h->d_rare, all, r0, h->cd_R, h->E, h->K, h->d_cd_vbuf, vstride,
                                                                                       h->d_cd_cmin, h->cd_nchunks, h->d_partial, nth_form, h->d_cd_stat);
        GH_LAUNCH_CHECK();
    }
    h->intersect_done = false;
    h->stats_reduced = false;
    return GH_OK;
}
