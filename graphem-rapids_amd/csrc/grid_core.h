// Sub-quadratic exact KNN of the sampled midpoints (the reference reaches for cuVS IVF
// indexes here, embedder_cuvs.py:255-313): a grid over this iteration's midpoints -- for D <= 3 over the midpoints
// themselves, for 4 <= D <= 16 over their PROJECTION onto the first three coordinates.  Dropping coordinates can only
// shorten a distance, so the projected ball of radius sqrt(tau) holds every midpoint within sqrt(tau) in all D
// coordinates: the cells it touches are a complete candidate set and the search stays EXACT (distances, thresholds and
// the final decision are always taken in all D coordinates); what the projection costs is selectivity, and it costs
// everything: on the 1 M-vertex bench state the shadow of the threshold ball holds more than the 8192 candidates a list
// takes and every query falls back to the exhaustive search -- 85 ms per iteration at D = 6, S = 4096 against 1.6 ms for
// the scan (profiles/r03/knn_method_sweep.log).  Hence the grid is for D <= 3 only (gh_grid_path; the wide form and its
// switch were removed in round 4);
// the matrix-pipe scan is the search for wide rows at every S tried (up to 65536).  Included at the end of knn.hip (it
// uses that file's K-smallest extraction).
//
// The filtered brute-force scan (fused.hip) costs S * E pre-filter evaluations per iteration and its thresholds
// S * E / stride exact distances: quadratic, fine up to a thousand queries or so (hidden under the spring phase's
// gathers).  This path costs O(E) per iteration to index the midpoints and then O(1) cells per query:
//
//   0. frame (one workgroup): centre = per-coordinate median of the query midpoints (a uniform sample of all
//      midpoints), scale = their inter-quartile ranges -- robust against the far outliers a layout carries (they
//      set the unit variance while the bulk shrinks); cells fine in the core and geometrically coarser outwards
//      (grid_coord below).  The frame stays in device memory (no host round trip);
//   1. cell id of every own midpoint, radix sort of (cell id, row) pairs (hipCUB = rocPRIM's device radix sort,
//      compiled in-tree), midpoints + edge ids gathered into cell order; the sorted ids double as the cell index
//      (binary search for the ends of a run of cells);
//   2. tau per query (one workgroup): the K-th smallest exact distance among the midpoints of the (2r+1)^D cells
//      around the query's cell, r = 1, 2, ... until they hold K: an upper bound of the true K-th distance, and a
//      tight one (these cells hold most of the true neighbours);
//   3. candidates per query: every midpoint within tau lies in a cell that intersects [q - sqrt(tau), q + sqrt(tau)]
//      (clamped like the points); rows of such cells are contiguous runs of the sorted array; exact squared
//      distance (the oracle's fma chain) <= tau -> the query's candidate list (a few dozen keys);
//   4. knn_select_kernel as on the scan path: K smallest keys, ties on the smaller edge id.
// The result is the exact KNN -- identical ids to the scan path and the oracle, not an approximation -- so
// "recall" against the exact kernel is 1 by construction (tests/test_hip_grid_knn.py checks identity).
#pragma once
#include <hipcub/hipcub.hpp>

namespace {

// Cell coordinate along one axis: monotone BY CONSTRUCTION in x (float subtraction, multiplication by a positive
// constant, |t| + 1 and the bit pattern of a positive float are all monotone), fine near the centre and
// geometrically coarser outwards: with t = (x - median) / sigma and v = |t| + 1, every binade [2^e, 2^(e+1)) of v is
// cut into P = 2^(23 - shift) equal cells -- P cells per sigma in the core, cell width doubling with every binade,
// GH_GRID_OCTAVES binades per side (|t| < 2^OCTAVES - 1; beyond: the outermost cells, unbounded).  A layout keeps a
// dense core and a halo of outliers hundreds of sigma out (they set the unit variance): a uniform grid either
// starves the core of resolution or leaves the halo in unbounded border cells whose contents say nothing about
// distance (first version: thresholds of halo queries 10^4 times too loose).
#define GH_GRID_OCTAVES 8
struct grid_frame {      // device-resident description of the grid
    float med[3];        // centre per coordinate (median of the query midpoints)
    float inv_s[3];      // 1 / robust sigma per coordinate (their inter-quartile range / 1.349)
    int shift;           // 23 - log2(cells per binade)
    int half;            // cells per side = P * GH_GRID_OCTAVES; G = 2 * half
};

__device__ __forceinline__ int grid_coord(float x, float med, float inv_s, int shift, int half) {
    const float t = (x - med) * inv_s;
    const float v = fabsf(t) + 1.0f;
    uint32_t k = (__float_as_uint(v) - 0x3F800000u) >> shift;   // NaN / inf: clamped below
    if (k > (uint32_t)(half - 1)) k = (uint32_t)(half - 1);
    return t >= 0.0f ? half + (int)k : half - 1 - (int)k;
}

// Frame from (up to 1024 of) the query midpoints: per-coordinate median and inter-quartile range by a bitonic
// sort in LDS.  One 256-thread workgroup.
__global__ __launch_bounds__(256) void grid_frame_kernel(const float *__restrict__ qt, int QS, int64_t S, int D, int half, int shift,
                                                        grid_frame *__restrict__ frame, int32_t *__restrict__ tcount_reset) {
    // set-up done inside the previous normalise launch: the touched-list counter is reset here (as knn_tau_kernel does)
    if (tcount_reset && threadIdx.x == 0) *tcount_reset = 0;
    __shared__ float v[1024];
    const int ns = (int)(S < 1024 ? S : 1024);
    const int64_t step = S / ns;
    for (int d = 0; d < 3; ++d) {
        if (d >= D) {
            if (threadIdx.x == 0) { frame->med[d] = 0.0f; frame->inv_s[d] = 1.0f; }
            continue;
        }
        for (int i = threadIdx.x; i < 1024; i += 256) v[i] = i < ns ? qt[(i * step) * QS + d] : INFINITY;
        __syncthreads();
        for (int k = 2; k <= 1024; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < 1024; i += 256) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const float a = v[i], b = v[ixj];
                        if ((a > b) == ((i & k) == 0)) { v[i] = b; v[ixj] = a; }
                    }
                }
                __syncthreads();
            }
        if (threadIdx.x == 0) {
            const float iqr = v[(3 * ns) / 4] - v[ns / 4];
            const float sig = iqr > 1e-30f && iqr < 1e30f ? iqr / 1.349f : 1.0f;   // sigma of a Gaussian with that IQR
            frame->med[d] = v[ns / 2] < INFINITY ? v[ns / 2] : 0.0f;
            frame->inv_s[d] = 1.0f / sig;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { frame->shift = shift; frame->half = half; }
}

template <int D /* grid dimensions: min(n_components, 3) */>
__global__ __launch_bounds__(256) void grid_cell_kernel(const float *__restrict__ mid, int LD, int64_t M,
                                                       const grid_frame *__restrict__ frame,
                                                       uint32_t *__restrict__ keys, uint32_t *__restrict__ rows) {
    const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j >= M) return;
    const grid_frame f = *frame;
    const float4 m = *reinterpret_cast<const float4 *>(mid + j * LD);   // the first coordinates of the row
    const float c[3] = {m.x, m.y, m.z};
    uint32_t key = 0;
#pragma unroll
    for (int d = D - 1; d >= 0; --d) key = key * (uint32_t)(2 * f.half) + (uint32_t)grid_coord(c[d], f.med[d], f.inv_s[d], f.shift, f.half);  // x fastest
    keys[j] = key;
    rows[j] = (uint32_t)j;
}

// First sorted position whose cell id is >= key (M when none): the ends of a run of cells.  The sorted ids stand in
// for a table of cell offsets: with cells fine in the core and coarse outside most of the 16.7 M cells are empty, and
// filling their offsets cost 1 ms per iteration (first version); a binary search is 22 L2 hits per run end, and a
// query touches a handful of runs.
__device__ __forceinline__ int grid_lower_bound(const uint32_t *__restrict__ skeys, int M, int64_t key) {
    int lo = 0, hi = M;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)skeys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// midpoints (whole rows of LD floats) and edge ids in cell order
__global__ __launch_bounds__(256) void grid_gather_kernel(const float *__restrict__ mid, int LD4 /* float4 per row */,
                                                         const uint32_t *__restrict__ srows, int64_t M, int64_t e_lo,
                                                         const int32_t *__restrict__ own_eids, float4 *__restrict__ smid,
                                                         uint32_t *__restrict__ sid) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= M * LD4) return;
    const int64_t i = t / LD4;
    const int part = (int)(t % LD4);
    const uint32_t j = srows[i];
    smid[t] = reinterpret_cast<const float4 *>(mid)[(int64_t)j * LD4 + part];
    if (part == 0) sid[i] = own_eids ? (uint32_t)own_eids[j] : (uint32_t)(e_lo + j);
}

// Squared distance in all D coordinates: the exact fma chain in coordinate order (the oracle's go_d2); row i of smid.
template <int D, int LD>
__device__ __forceinline__ float grid_d2(const float (&q)[LD], const float4 *__restrict__ smid, int64_t i) {
    float mm[LD];
#pragma unroll
    for (int p = 0; p < LD / 4; ++p) {
        const float4 v = smid[i * (LD / 4) + p];
        mm[4 * p] = v.x; mm[4 * p + 1] = v.y; mm[4 * p + 2] = v.z; mm[4 * p + 3] = v.w;
    }
    float d2 = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float df = q[d] - mm[d];
        d2 = fmaf(df, df, d2);
    }
    return d2;
}
// query record -> coordinates (LD floats, pad 0) and the offset of tau in it (knn.hip gh_qs / gh_qtau)
template <int D, int LD>
__device__ __forceinline__ void grid_query(const float *__restrict__ qt, int qi, float (&q)[LD]) {
    constexpr int QS = D <= 3 ? 4 : LD + 4;
#pragma unroll
    for (int p = 0; p < LD / 4; ++p) {
        const float4 v = reinterpret_cast<const float4 *>(qt + (int64_t)qi * QS)[p];
        q[4 * p] = v.x; q[4 * p + 1] = v.y; q[4 * p + 2] = v.z; q[4 * p + 3] = v.w;
    }
    if constexpr (D <= 3) q[3] = 0.0f;   // (x, y, z|0, tau): slot 3 is tau, not a coordinate
}

// tau of one query: K-th smallest exact distance among the midpoints of the cells within r of its own cell,
// the smallest r in 1 .. GH_GRID_RMAX whose block holds K midpoints (else tau = inf: exact fallback in the select
// kernel).  One workgroup; chunks of 2048 keys through the K-smallest extraction of knn.hip.
#define GH_GRID_RMAX 6
template <int D, int LD>
__global__ __launch_bounds__(256) void grid_tau_kernel(const float4 *__restrict__ smid, const uint32_t *__restrict__ skeys, int M,
                                                      const grid_frame *__restrict__ frame, float *__restrict__ qt, int K) {
    constexpr int GD = D < 3 ? D : 3;                 // dimensions of the grid
    constexpr int QS = D <= 3 ? 4 : LD + 4, QT = D <= 3 ? 3 : LD;
    __shared__ uint64_t best[GH_EXTRACT_MAX_K];
    __shared__ uint64_t red[4 * GH_EXTRACT_MAX_K];
    __shared__ int run_beg[(2 * GH_GRID_RMAX + 1) * (2 * GH_GRID_RMAX + 1)], run_len[(2 * GH_GRID_RMAX + 1) * (2 * GH_GRID_RMAX + 1)];
    __shared__ int total;
    const int qi = blockIdx.x;
    const grid_frame f = *frame;
    float q[LD];
    grid_query<D, LD>(qt, qi, q);
    int c[3] = {0, 0, 0};
    const int G = 2 * f.half;
#pragma unroll
    for (int d = 0; d < GD; ++d) c[d] = grid_coord(q[d], f.med[d], f.inv_s[d], f.shift, f.half);
    constexpr int NPT = 8;
    float tau = INFINITY;
    for (int r = 1; r <= GH_GRID_RMAX; ++r) {
        int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
#pragma unroll
        for (int d = 0; d < GD; ++d) { lo[d] = max(c[d] - r, 0); hi[d] = min(c[d] + r, G - 1); }
        const int ny = hi[1] - lo[1] + 1, nz = GD == 3 ? hi[2] - lo[2] + 1 : 1, nrun = ny * nz;
        __syncthreads();
        if (threadIdx.x == 0) total = 0;
        __syncthreads();
        for (int t = threadIdx.x; t < nrun; t += 256) {   // runs of this block and their lengths
            const int y = lo[1] + t % ny, z = lo[2] + t / ny;
            const int64_t base = ((int64_t)z * G + y) * G;
            const int b = grid_lower_bound(skeys, M, base + lo[0]), e = grid_lower_bound(skeys, M, base + hi[0] + 1);
            run_beg[t] = b;
            run_len[t] = e - b;
            atomicAdd(&total, e - b);
        }
        __syncthreads();
        if (total < K && r < GH_GRID_RMAX) continue;   // not enough midpoints yet: next ring
        if (total < K) break;                         // sparse beyond the last ring: tau stays inf
        for (int i = threadIdx.x; i < K; i += 256) best[i] = GH_KEY_INF;
        __syncthreads();
        // flat index over the runs, 2048 keys at a time
        for (int base = 0; base < total; base += 256 * NPT) {
            const uint64_t tk = best[K - 1];
            uint64_t keys[NPT + 1];
            int any = 0;
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                int p = base + j * 256 + (int)threadIdx.x;
                uint64_t key = GH_KEY_INF;
                if (p < total) {
                    int t = 0;
                    while (p >= run_len[t]) { p -= run_len[t]; ++t; }   // a handful of runs
                    const int i = run_beg[t] + p;
                    key = gh_key(grid_d2<D, LD>(q, smid, i), (uint32_t)i);
                    if (key < tk) any = 1; else key = GH_KEY_INF;
                }
                keys[j] = key;
            }
            if (__syncthreads_or(any)) {
                keys[NPT] = threadIdx.x < K ? best[threadIdx.x] : GH_KEY_INF;
                __syncthreads();
                block_extract_smallest<NPT + 1>(keys, K, best, red);
            }
        }
        tau = gh_key_d2(best[K - 1]);
        break;
    }
    if (threadIdx.x == 0) qt[qi * QS + QT] = tau;
}

// Queries in regions so sparse that GH_GRID_RMAX rings hold fewer than K midpoints (far outliers of the layout)
// still need a finite tau: the K-th smallest of the workgroup's 1024 per-thread minima over every `stride`-th midpoint of
// the cell-sorted array -- the group-minima bound of setup_core.h over a global subset.  Workgroups of queries that
// already have their tau leave at once.
template <int D, int LD>
__global__ __launch_bounds__(1024) void grid_tau_fallback_kernel(const float4 *__restrict__ smid, int64_t M, int64_t stride,
                                                                float *__restrict__ qt, int K) {
    constexpr int QS = D <= 3 ? 4 : LD + 4, QT = D <= 3 ? 3 : LD;
    __shared__ uint64_t best[GH_EXTRACT_MAX_K];
    __shared__ uint64_t red[16 * GH_EXTRACT_MAX_K];
    const int qi = blockIdx.x;
    if (qt[(int64_t)qi * QS + QT] < INFINITY) return;
    float q[LD];
    grid_query<D, LD>(qt, qi, q);
    float mn = INFINITY;
    for (int64_t j = threadIdx.x; j * stride < M; j += 1024) mn = fminf(mn, grid_d2<D, LD>(q, smid, j * stride));
    uint64_t keys[1] = {mn < INFINITY ? gh_key(mn, threadIdx.x) : GH_KEY_INF};
    block_extract_smallest<1, 1024>(keys, K, best, red);
    if (threadIdx.x == 0 && best[K - 1] != GH_KEY_INF) qt[(int64_t)qi * QS + QT] = gh_key_d2(best[K - 1]);
}

// Candidates of one query: every run of cells its box touches, exact distance, keys within tau appended.  The runs are
// dealt over gridDim.y workgroups: an outlier's box can cover the whole bulk, and one workgroup would read it alone.
template <int D, int LD>
__global__ __launch_bounds__(256) void grid_scan_kernel(const float4 *__restrict__ smid, const uint32_t *__restrict__ sid,
                                                       const uint32_t *__restrict__ skeys, int M,
                                                       const grid_frame *__restrict__ frame,
                                                       const float *__restrict__ qt, uint64_t *__restrict__ cand,
                                                       int32_t *__restrict__ cnt) {
    constexpr int GD = D < 3 ? D : 3;
    constexpr int QS = D <= 3 ? 4 : LD + 4, QT = D <= 3 ? 3 : LD;
    const int qi = blockIdx.x;
    const grid_frame f = *frame;
    float q[LD];
    grid_query<D, LD>(qt, qi, q);
    const float tau = qt[(int64_t)qi * QS + QT];
    if (!(tau < INFINITY)) {   // fewer than K midpoints in reach of any bound: mark the list as overflowed, the select
        if (blockIdx.y == 0 && threadIdx.x == 0) cnt[qi * GH_CNT_STRIDE] = GH_CAND_CAP + 1;   // kernel searches this query exactly
        return;
    }
    // sqrt rounded up a little: the box must contain the ball of the EXACT test below (d2 <= tau); in the grid's
    // coordinates alone a midpoint within sqrt(tau) in all D coordinates is within sqrt(tau) too
    const float rad = sqrtf(tau) * 1.000001f + 1e-30f;
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < GD; ++d) {
        lo[d] = grid_coord(q[d] - rad, f.med[d], f.inv_s[d], f.shift, f.half);
        hi[d] = grid_coord(q[d] + rad, f.med[d], f.inv_s[d], f.shift, f.half);
    }
    const int G = 2 * f.half;
    const int ny = hi[1] - lo[1] + 1, nz = GD == 3 ? hi[2] - lo[2] + 1 : 1;
    // rows of cells along x: one contiguous run each; 256 runs at a time, their ends found by 256 lanes side by side
    __shared__ int rbeg[256], rend[256];
    const int nrun = ny * nz;
    for (int r0 = blockIdx.y * 256; r0 < nrun; r0 += gridDim.y * 256) {
        __syncthreads();
        const int r = r0 + (int)threadIdx.x;
        if (r < nrun) {
            const int y = lo[1] + r % ny, z = lo[2] + r / ny;
            const int64_t base = ((int64_t)z * G + y) * G;
            rbeg[threadIdx.x] = grid_lower_bound(skeys, M, base + lo[0]);
            rend[threadIdx.x] = grid_lower_bound(skeys, M, base + hi[0] + 1);
        }
        __syncthreads();
        const int nr = min(256, nrun - r0);
        for (int t = 0; t < nr; ++t)
            for (int i = rbeg[t] + (int)threadIdx.x; i < rend[t]; i += 256) {
                const float d2 = grid_d2<D, LD>(q, smid, i);
                if (d2 <= tau) gh_append_candidate(cand, cnt, qi, gh_key(d2, sid[i]));
            }
    }
}

}  // namespace

bool gh_grid_path(const gh_engine *h) {
    // 2 or 3 components only.  (Round 2 also ran it over the first three of 4..16 coordinates: exact, but measured 50-80x SLOWER
    // than the scan on the 1 M-vertex graph, profiles/r03/knn_method_sweep.log -- the three-coordinate shadow of a 6- or
    // 16-dimensional threshold ball holds more midpoints than a candidate list takes; removed in round 4.)
    return h->prm.knn_method == GH_KNN_GRID && h->D >= 2 && h->D <= 3 && gh_knn_scan_path(h);
}

gh_status gh_grid_alloc(gh_engine *h) {
    if (!gh_grid_path(h)) return GH_OK;
    // cells per binade and axis: 16 (3-D: a central cell holds ~60 of 4M midpoints of a Gaussian core) / 128 (2-D)
    const int P = h->D >= 3 ? 16 : 128;
    const int G = 2 * P * GH_GRID_OCTAVES;
    h->grid_G = G;
    int64_t ncells = 1;
    for (int d = 0; d < std::min(h->D, 3); ++d) ncells *= G;
    h->grid_cells = ncells;
    const size_t M = (size_t)h->own_count;
    size_t temp = 0;
    int bits = 1;
    while (((int64_t)1 << bits) < ncells) ++bits;
    h->grid_bits = bits;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr,
                                           (uint32_t *)nullptr, (int)M, 0, bits, h->stream) != hipSuccess) {
        h->err = "hipcub radix sort size query failed";
        return GH_ERR_HIP;
    }
    h->grid_temp_bytes = temp;
    const size_t words = 4 * M + M + 16;   // keys, rows (in + out), edge ids, frame
    if (hipMalloc(reinterpret_cast<void **>(&h->d_grid_u32), sizeof(uint32_t) * words) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&h->d_grid_smid), sizeof(float) * ((size_t)h->LD * M + 4)) != hipSuccess ||
        hipMalloc(&h->d_grid_temp, temp + 16) != hipSuccess) {
        h->err = "hipMalloc of the grid buffers failed";
        return GH_ERR_NOMEM;
    }
    return GH_OK;
}

// d_mid (this iteration's own midpoints) + the query records -> tau of every query and its candidate list.
gh_status gh_grid_search(gh_engine *h) {
    const int64_t M = h->own_count;
    uint32_t *keys = h->d_grid_u32, *rows = keys + M, *skeys = rows + M, *srows = skeys + M;
    uint32_t *sid = srows + M;
    grid_frame *frame = reinterpret_cast<grid_frame *>(sid + M);
    float4 *smid = reinterpret_cast<float4 *>(h->d_grid_smid);
    const unsigned gridM = (unsigned)((M + 255) / 256);
    const int QS = gh_qs(h->D, h->LD), LD4 = h->LD / 4;
    {
        gh_scope t(h, "grid_build");
        grid_frame_kernel<<<dim3(1), dim3(256), 0, h->stream>>>(h->d_q, QS, h->S, std::min(h->D, 3), h->grid_G / 2, h->D >= 3 ? 23 - 4 : 23 - 7, frame,
                                                                  h->tcount_reset_pending ? h->d_tcount : nullptr);
        if (h->D == 2) grid_cell_kernel<2><<<dim3(gridM), dim3(256), 0, h->stream>>>(h->d_mid, h->LD, M, frame, keys, rows);
        else grid_cell_kernel<3><<<dim3(gridM), dim3(256), 0, h->stream>>>(h->d_mid, h->LD, M, frame, keys, rows);
        size_t temp = h->grid_temp_bytes;
        GH_HIP(hipcub::DeviceRadixSort::SortPairs(h->d_grid_temp, temp, keys, skeys, rows, srows, (int)M, 0, h->grid_bits, h->stream));
        grid_gather_kernel<<<dim3((unsigned)((M * LD4 + 255) / 256)), dim3(256), 0, h->stream>>>(h->d_mid, LD4, srows, M, h->part.edge_lo,
                                                                                               h->d_own_eids, smid, sid);
        GH_LAUNCH_CHECK();
    }
    gh_scope t(h, "grid_tau_scan");
    const int64_t fb_stride = M >= 8 * 64 * (int64_t)h->K ? 8 : 1;   // sparse queries only: a tight bound keeps their boxes small
    const dim3 sgrid((unsigned)h->S, h->S <= 2048 ? 32 : h->S <= 16384 ? 16 : 4);   // an outlier's box can hold the whole bulk: its runs over several workgroups
#define GH_GRID_ONE(DD, LL)                                                                                                              \
    case DD:                                                                                                                              \
        grid_tau_kernel<DD, LL><<<dim3((unsigned)h->S), dim3(256), 0, h->stream>>>(smid, skeys, (int)M, frame, h->d_q, h->K);                 \
        grid_tau_fallback_kernel<DD, LL><<<dim3((unsigned)h->S), dim3(1024), 0, h->stream>>>(smid, M, fb_stride, h->d_q, h->K);               \
        grid_scan_kernel<DD, LL><<<sgrid, dim3(256), 0, h->stream>>>(smid, sid, skeys, (int)M, frame, h->d_q, h->d_cand, h->d_cnt);          \
        break;
    switch (h->D) {
        GH_FOR_EACH_DIM(GH_GRID_ONE)
        default: h->err = "grid KNN: unsupported dimension"; return GH_ERR_RUNTIME;
    }
#undef GH_GRID_ONE
    GH_LAUNCH_CHECK();
    return GH_OK;
}
