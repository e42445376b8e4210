// GH_KNN_IVF: an inverted-file (IVF-Flat) search of the sampled midpoints, rebuilt every iteration like the reference's
// cuVS backend rebuilds its index (embedder_cuvs.py:255-313 builds IVF-Flat / IVF-PQ over the midpoints of the iteration,
// :384-430 searches it).  Two modes (gh_params.ivf_probes):
//   >= 0  APPROXIMATE, the cuVS semantics: a query sees the members of the `probes` lists whose centroids are nearest to it
//         and gets the k + 1 smallest (distance, id) keys among them -- exact inside those lists (the exact-difference fma
//         chain every other method ranks on), blind outside;
//   <  0  EXACT: a query probes every list that can hold one of its k + 1 nearest (ivf_probe_kernel<.., true>: centroid
//         within sqrt(tau) + min(list radius, sqrt(tau) + distance to the query's nearest centroid), f16 errors on the safe
//         side), so the rows are those of the scan, id for id.  What GH_KNN_AUTO takes for 2-8 components and thousands of
//         queries (api.hip).
//
// What makes it an MI355X design rather than a port of a query-major IVF search:
//  * the coarse quantiser is a flat argmin over C centroids on the MATRIX pipe: one v_mfma_f32_32x32x16_f16 gives
//    |c|^2 - 2 c.m for 32 centroids x 32 midpoints with single-piece f16 operands (the assignment only has to be A
//    partition -- the exact mode's bounds carry the f16 error --), the winning group of four rows rides in the two low
//    mantissa bits of the score through the min tree, the row among the four is settled once per midpoint;
//  * list positions without device-wide atomic traffic: LDS counters per workgroup, one reservation per (workgroup, list);
//  * the search is LIST-major: (query, probed list) pairs are bucketed by list, and a workgroup takes one 512-member tile
//    of one list and runs that list's queries over it with the filtered scan of scan_core.h (matrix-pipe pre-filter for
//    4-16 components, packed fp32 below) -- a list is read once per iteration (E * LD * 4 bytes in all), not once per
//    probing query;
//  * thresholds come from a sample of the query's nearest lists: 256 group minima per query, the K-th smallest of them
//    (the existing threshold kernel, tau_core.h) bounds the K-th smallest distance from above, so the filtered scan finds
//    at least K candidates and usually few more.
// Everything after the candidate lists (selection, intersection phase) is the code of the other methods.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "engine.h"
#include "scan_core.h"

#define GH_IVF_TILE 512          /* members per scan workgroup (R = 2 references per thread) */
#define GH_IVF_MAX_LISTS 2048
#define GH_IVF_GROUPS 256        /* group minima per query the threshold is taken from */
#define GH_IVF_MB 4              /* member blocks of 32 a wave of the assignment holds */
// Counters that many workgroups bump with returning atomics sit one per 128-byte line (adjacent counters serialise on
// their L2 line, ~11 ns per returning atomic).  The members of a list are counted in two levels: a workgroup of the
// assignment files its share (thousands of midpoints) under LDS counters -- the position inside the workgroup's share --
// and reserves one range per list with ONE global atomic; 4 M device-scope returning atomics were a floor of 0.2 ms
// under the kernel whatever the number of lists (and 0.7 ms while 32 counters shared a line).
#define GH_IVF_CSTRIDE 32
#define GH_IVF_ASSIGN_WGS 3072   /* at most this many workgroups (each walks its share in steps of 512 midpoints) */

struct gh_ivf {
    int C = 0, P = 0;             // lists, probes per query (exact mode: the most a query may probe before it falls back)
    bool exact = false;           // ivf_probes < 0: every list that can hold one of the k + 1 nearest is probed (exact search)
    uint32_t *r2 = nullptr;       // (C * CSTRIDE) exact mode: upper bound of the squared radius of every list, float bits
    int64_t share = 0;            // midpoints per workgroup of the assignment (a multiple of 512)
    int awgs = 0;                 // workgroups of the assignment
    int64_t M = 0;                // own midpoints
    int64_t cap_rows = 0;         // rows of the list-ordered copy: M + C * GH_IVF_TILE (every list padded to whole tiles)
    int64_t max_tiles = 0;
    unsigned char *blob = nullptr;
    float *cent = nullptr;        // (C, LD) centroids, fp32
    uint4 *A = nullptr;           // (C, 2) f16 operand rows: -2 c, 8 halfs per lane half
    float *cnorm = nullptr;       // (C) |c|^2
    uint32_t *assign = nullptr;   // (M) list of every own midpoint
    uint32_t *rank = nullptr;     // (M) its position inside the list
    int32_t *count = nullptr;     // (C * CSTRIDE) members per list, one counter per line
    int32_t *wgbase = nullptr;    // (awgs, C) first position inside the list of what workgroup w filed under it
    int32_t *lcount = nullptr;    // (C) members
    int32_t *lstart = nullptr;    // (C + 1) first row of every list in the padded order
    int32_t *tile_list = nullptr; // (max_tiles) list of every tile
    int32_t *meta = nullptr;      // [0] tiles in use
    float *lmid = nullptr;        // (cap_rows, LD) midpoints in list order
    uint32_t *lids = nullptr;     // (cap_rows) their edge ids; 0xFFFFFFFF = padding
    int32_t *lqcount = nullptr;   // (C * CSTRIDE) probing queries per list, one counter per line
    int32_t *qstart = nullptr;    // (C + 1)
    uint32_t *pair_l = nullptr;   // (S, P) probed list of pair (query, r)
    uint32_t *pair_slot = nullptr;// (S, P) position of the query among the list's queries
    int32_t *pair_q = nullptr;    // (S * P) queries bucketed by list
};

namespace {

typedef _Float16 ivf_h8 __attribute__((ext_vector_type(8)));
typedef float ivf_f16x __attribute__((ext_vector_type(16)));

__device__ __forceinline__ _Float16 ivf_half(float x) { return (_Float16)fminf(fmaxf(x, -30000.0f), 30000.0f); }
// (This file is compiled with -fno-honor-nans, Makefile: fminf() otherwise makes the compiler quiet a possible signalling
// NaN in every operand first -- one v_max_f32 x, x, x each, 32 of them per matrix result in the assignment's loop.  No value
// here is ever a NaN: coordinates are finite, infinities only mark "no bound" and are never subtracted from one another.
// Inline-asm minima are not an option: the compiler does not insert the wait states between a matrix instruction and an
// asm statement that reads its result -- tried, wrong assignments.)
__device__ __forceinline__ float ivf_min3(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float ivf_min2(float a, float b) { return fminf(a, b); }

// Centroids = this iteration's midpoints of C evenly spaced own edges (a sample of the data's own density, as k-means++
// seeding without the refinement passes: lists come out at roughly equal mass), their operand rows and norms; counters
// of the iteration zeroed.
template <int LD>
__global__ __launch_bounds__(256) void ivf_centroid_kernel(const float *__restrict__ mid, int64_t M, int C, float *__restrict__ cent,
                                                           uint4 *__restrict__ A, float *__restrict__ cnorm,
                                                           int32_t *__restrict__ count, int32_t *__restrict__ lqcount,
                                                           uint32_t *__restrict__ r2) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    count[c * GH_IVF_CSTRIDE] = 0;     // the iteration's counters (only the first word of a line is ever used)
    lqcount[c * GH_IVF_CSTRIDE] = 0;
    if (r2) r2[c * GH_IVF_CSTRIDE] = 0;
    const int64_t j = (int64_t)c * M / C + (M / C) / 2;
    float v[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) v[d] = 0.0f;
    gh_load_row<LD>(mid, j, v);
    gh_store_row<LD>(cent, c, v);
    float nn = 0.0f;
#pragma unroll
    for (int d = 0; d < 16; ++d) nn = fmaf(v[d], v[d], nn);
    cnorm[c] = nn;
    ivf_h8 lo, hi;
#pragma unroll
    for (int d = 0; d < 8; ++d) { lo[d] = ivf_half(-2.0f * v[d]); hi[d] = ivf_half(-2.0f * v[8 + d]); }
    A[2 * c] = __builtin_bit_cast(uint4, lo);
    A[2 * c + 1] = __builtin_bit_cast(uint4, hi);
}

// Nearest centroid of every own midpoint.  A wave holds GH_IVF_MB blocks of 32 midpoints as B operands (lane = column,
// lane half = which 8 of the 16 coordinates) and walks the centroids 32 at a time: accumulator initialised with |c|^2,
// one MFMA per member block, minimum per group of four rows with the group in the low 2 mantissa bits, minimum of the four,
// running best per column.  The two lane halves of a column hold different rows: combined at the end.  A workgroup walks its
// share of the midpoints in steps of 512 with the centroid table staged once; positions inside the lists: see
// GH_IVF_CSTRIDE above.
template <int LD>
__global__ __launch_bounds__(256) void ivf_assign_kernel(const float *__restrict__ mid, int64_t M, int64_t share, int C,
                                                         const uint4 *__restrict__ A, const float *__restrict__ cnorm,
                                                         uint32_t *__restrict__ assign, uint32_t *__restrict__ rank,
                                                         int32_t *__restrict__ count, int32_t *__restrict__ wgbase,
                                                         uint32_t *__restrict__ r2 /* exact mode, else null */) {
    extern __shared__ __align__(16) unsigned char ivf_smem[];
    uint4 *Ash = reinterpret_cast<uint4 *>(ivf_smem);              // (C, 2)
    float *nsh = reinterpret_cast<float *>(Ash + 2 * (size_t)C);     // (C)
    int *hist = reinterpret_cast<int *>(nsh + C);                    // (C) members this workgroup filed under the list
    uint32_t *rmax = reinterpret_cast<uint32_t *>(hist + C);         // (C) exact mode: their radius bound
    for (int i = threadIdx.x; i < 2 * C; i += 256) Ash[i] = A[i];
    for (int i = threadIdx.x; i < C; i += 256) { nsh[i] = cnorm[i]; hist[i] = 0; rmax[i] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = lane & 31, hsel = lane >> 5;
    const int64_t lo = (int64_t)blockIdx.x * share, hi = min(M, lo + share);
    const int ncb = C / 32;
    for (int64_t step = lo; step < hi; step += 4 * 32 * GH_IVF_MB) {
        const int64_t base = step + w * (32 * GH_IVF_MB);
        ivf_h8 B[GH_IVF_MB];
        float mnorm[GH_IVF_MB];   // this lane half's share of |m|^2; +inf when a coordinate is beyond what the f16 operand can carry
#pragma unroll
        for (int mb = 0; mb < GH_IVF_MB; ++mb) {
            const int64_t j = base + mb * 32 + col;
            float v[8];   // this lane half's 8 coordinates (coordinates past LD are 0)
#pragma unroll
            for (int d = 0; d < 8; ++d) v[d] = 0.0f;
            if (j < hi && hsel * 8 < LD) {
                constexpr int NQ = LD >= 8 ? 2 : 1;
                const float4 *src = reinterpret_cast<const float4 *>(mid + j * LD + hsel * 8);
#pragma unroll
                for (int p = 0; p < NQ; ++p) {
                    const float4 x = src[p];
                    v[4 * p] = x.x; v[4 * p + 1] = x.y; v[4 * p + 2] = x.z; v[4 * p + 3] = x.w;
                }
            }
            mnorm[mb] = 0.0f;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                B[mb][d] = ivf_half(v[d]);
                mnorm[mb] = fabsf(v[d]) <= 30000.0f ? fmaf(v[d], v[d], mnorm[mb]) : INFINITY;
            }
        }
        float best[GH_IVF_MB];
        int bestcb[GH_IVF_MB];
#pragma unroll
        for (int mb = 0; mb < GH_IVF_MB; ++mb) { best[mb] = INFINITY; bestcb[mb] = 0; }
        for (int cb = 0; cb < ncb; ++cb) {
            const ivf_h8 a = __builtin_bit_cast(ivf_h8, Ash[(cb * 32 + col) * 2 + hsel]);
            ivf_f16x cinit;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 n4 = *reinterpret_cast<const float4 *>(nsh + cb * 32 + 8 * g + 4 * hsel);
                cinit[4 * g] = n4.x; cinit[4 * g + 1] = n4.y; cinit[4 * g + 2] = n4.z; cinit[4 * g + 3] = n4.w;
            }
            ivf_f16x f = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B[0], cinit, 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < GH_IVF_MB; ++mb) {
                ivf_f16x fn = cinit;
                if (mb + 1 < GH_IVF_MB) fn = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B[mb + 1], cinit, 0, 0, 0);   // in the matrix pipe while block mb is reduced
                // minimum of each group of four rows (rows 8g + 4 half + 0..3: four CONSECUTIVE centroids), the group in the two
                // low mantissa bits, minimum of the four: 17 vector instructions per result instead of the 27 of carrying all
                // four index bits through 16 values; which of the four it was is settled once per midpoint at the end
                float pg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float mg = ivf_min2(ivf_min3(f[4 * g], f[4 * g + 1], f[4 * g + 2]), f[4 * g + 3]);
                    pg[g] = __uint_as_float((__float_as_uint(mg) & ~3u) | (uint32_t)g);
                }
                float mn = ivf_min2(ivf_min3(pg[0], pg[1], pg[2]), pg[3]);
                const bool better = mn < best[mb];
                best[mb] = better ? mn : best[mb];
                bestcb[mb] = better ? cb : bestcb[mb];
                f = fn;
            }
        }
#pragma unroll
        for (int mb = 0; mb < GH_IVF_MB; ++mb) {
            const float ob = __shfl_xor(best[mb], 32, 64);
            const int ocb = __shfl_xor(bestcb[mb], 32, 64);
            const float mn2 = mnorm[mb] + __shfl_xor(mnorm[mb], 32, 64);
            const uint4 own8 = __builtin_bit_cast(uint4, B[mb]);   // the other lane half's 8 coordinates of the same midpoint
            const uint4 oth8 = make_uint4(__shfl_xor(own8.x, 32, 64), __shfl_xor(own8.y, 32, 64), __shfl_xor(own8.z, 32, 64), __shfl_xor(own8.w, 32, 64));
            const int64_t j = base + mb * 32 + col;
            if (hsel == 0 && j < hi) {
                const bool mine = best[mb] <= ob;
                const float bq = mine ? best[mb] : ob;
                const int cb = mine ? bestcb[mb] : ocb;
                const int l0 = cb * 32 + 8 * (int)(__float_as_uint(bq) & 3u) + 4 * (mine ? 0 : 1);
                // the four centroids of the winning group: their scores again, from the same f16 operands (fp32 fma chain: it
                // differs from the matrix pipe's accumulation by rounding only), smallest first
                const ivf_h8 mlo = __builtin_bit_cast(ivf_h8, own8), mhi = __builtin_bit_cast(ivf_h8, oth8);
                float bv = INFINITY;
                int l = l0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const ivf_h8 alo = __builtin_bit_cast(ivf_h8, Ash[(l0 + e) * 2]), ahi = __builtin_bit_cast(ivf_h8, Ash[(l0 + e) * 2 + 1]);
                    float sc = nsh[l0 + e];
#pragma unroll
                    for (int d = 0; d < 8; ++d) sc = fmaf((float)alo[d], (float)mlo[d], sc);
                    if constexpr (LD > 8) {
#pragma unroll
                        for (int d = 0; d < 8; ++d) sc = fmaf((float)ahi[d], (float)mhi[d], sc);
                    }
                    if (sc < bv) { bv = sc; l = l0 + e; }
                }
                assign[j] = (uint32_t)l;
                rank[j] = (uint32_t)atomicAdd(&hist[l], 1);   // LDS: position inside this workgroup's share of the list
                if (r2) {
                    // |m - c|^2 <= score + |m|^2 + what the f16 operands dropped: <= 2^-10 (|c|^2 + |m|^2) over the products, the
                    // accumulation and the four index bits far below; doubled.  A centroid or member beyond the f16 operands' clamp
                    // makes the bound infinite (the list is then probed by every query).
                    // (f16 subnormals: absolute 2^-25 per operand, <= 1e-6 sqrt(|c|^2 + |m|^2) over the 16 products)
                    const float cn = nsh[l];
                    const float up = fmaxf(bv + mn2, 0.0f) + (1.953125e-3f * (cn + mn2) + 1e-6f * sqrtf(cn + mn2)) + 1e-30f;
                    const bool fin = mn2 < INFINITY && cn <= 15000.0f * 15000.0f;   // no coordinate of -2c reached the clamp
                    atomicMax(&rmax[l], fin ? __float_as_uint(up) : 0x7F800000u);
                }
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int n = hist[c];
        if (n > 0) wgbase[(int64_t)blockIdx.x * C + c] = atomicAdd(&count[c * GH_IVF_CSTRIDE], n);
        if (r2 && rmax[c] > 0) atomicMax(&r2[c * GH_IVF_CSTRIDE], rmax[c]);
    }
}

// Exclusive prefix over 1024 per-thread sums (one workgroup); returns what precedes this thread, *total = the sum.
__device__ __forceinline__ int ivf_block_prefix(int sum, int *part, int *total) {
    const int t = threadIdx.x;
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    *total = part[1023];
    return part[t] - sum;
}

// One workgroup: every list padded to whole tiles -> lstart (C + 1), lcount, the list of every tile, the tiles in use.
__global__ __launch_bounds__(1024) void ivf_list_layout_kernel(const int32_t *__restrict__ count, int C, int32_t *__restrict__ lcount,
                                                               int32_t *__restrict__ lstart, int32_t *__restrict__ tile_list,
                                                               int32_t *__restrict__ meta) {
    __shared__ int part[1024];
    constexpr int PER = GH_IVF_MAX_LISTS / 1024;
    const int t = threadIdx.x;
    int tot[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = t * PER + i;
        tot[i] = c < C ? count[c * GH_IVF_CSTRIDE] : 0;
        sum += (tot[i] + GH_IVF_TILE - 1) / GH_IVF_TILE * GH_IVF_TILE;
    }
    int total;
    int at = ivf_block_prefix(sum, part, &total);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = t * PER + i;
        if (c >= C) break;
        lstart[c] = at;
        lcount[c] = tot[i];
        const int nt = (tot[i] + GH_IVF_TILE - 1) / GH_IVF_TILE;
        for (int k = 0; k < nt; ++k) tile_list[at / GH_IVF_TILE + k] = c;
        at += nt * GH_IVF_TILE;
    }
    if (t == 0) { lstart[C] = total; meta[0] = total / GH_IVF_TILE; }
}

// One workgroup: probing queries per list -> qstart (C + 1).
__global__ __launch_bounds__(1024) void ivf_query_layout_kernel(const int32_t *__restrict__ lqcount, int C, int32_t *__restrict__ qstart) {
    __shared__ int part[1024];
    constexpr int PER = GH_IVF_MAX_LISTS / 1024;
    const int t = threadIdx.x;
    int cnt[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = t * PER + i;
        cnt[i] = c < C ? lqcount[c * GH_IVF_CSTRIDE] : 0;
        sum += cnt[i];
    }
    int total;
    int at = ivf_block_prefix(sum, part, &total);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = t * PER + i;
        if (c >= C) break;
        qstart[c] = at;
        at += cnt[i];
    }
    if (t == 0) qstart[C] = total;
}

// Midpoints and edge ids into list order.
template <int LD>
__global__ __launch_bounds__(256) void ivf_scatter_kernel(const float *__restrict__ mid, int64_t M, const uint32_t *__restrict__ assign,
                                                          const uint32_t *__restrict__ rank, const int32_t *__restrict__ lstart,
                                                          const int32_t *__restrict__ wgbase, int64_t share, int C,
                                                          int64_t e_lo, const int32_t *__restrict__ eids, float *__restrict__ lmid,
                                                          uint32_t *__restrict__ lids) {
    constexpr int Q = LD / 4;   // 16-byte pieces per row: consecutive threads move consecutive pieces
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t j = i / Q;
    const int p = (int)(i % Q);
    if (j >= M) return;
    const uint32_t l = assign[j];
    const int64_t dst = (int64_t)lstart[l] + wgbase[(j / share) * C + l] + rank[j];
    reinterpret_cast<float4 *>(lmid)[dst * Q + p] = reinterpret_cast<const float4 *>(mid)[j * Q + p];
    if (p == 0) lids[dst] = eids ? (uint32_t)eids[j] : (uint32_t)(e_lo + j);
}

__device__ __forceinline__ uint64_t ivf_wave_min_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint64_t o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

// One wave per query, 16 queries per workgroup: distances to the C centroids (NV per lane) from the f16 operand table
// staged in LDS -- |c|^2 - 2 c.q + |q|^2 with c to 11 bits --, the lists to probe -> pairs, and the group minima (exact
// distances) of a sample of the nearest lists' members, from which the threshold is taken.
//   EXACT = false: the P lists with the smallest centroid distance (the ORDER of the lists is all that is needed).
//   EXACT = true:  every list that can hold one of the K nearest: with tau >= the K-th smallest distance (the same
//                  K-th smallest of the group minima the threshold kernel extracts), a member x with |q - x|^2 <= tau of a
//                  list with centroid c and radius R >= |x - c| has |q - c| <= sqrt(tau) + R; the centroid distance enters
//                  with its f16 error subtracted.  A query that would probe more than P lists is handed to the exhaustive
//                  search of the selection kernel (its candidate counter is pushed past the capacity).
template <int LD, int NV, bool EXACT>
__global__ __launch_bounds__(1024) void ivf_probe_kernel(const float *__restrict__ qt, int QS, int S, int D, const uint4 *__restrict__ A,
                                                         const float *__restrict__ cnorm, int C, int P, int K, const int32_t *__restrict__ lstart,
                                                         const int32_t *__restrict__ lcount, const float *__restrict__ lmid, int tau_members,
                                                         const uint32_t *__restrict__ r2, uint32_t *__restrict__ pair_l,
                                                         uint32_t *__restrict__ pair_slot, int32_t *__restrict__ lqcount,
                                                         uint32_t *__restrict__ gmin, int32_t *__restrict__ cnt) {
    extern __shared__ __align__(16) unsigned char ivf_smem[];
    uint4 *Ash = reinterpret_cast<uint4 *>(ivf_smem);              // (C, 2)
    float *nsh = reinterpret_cast<float *>(Ash + 2 * (size_t)C);     // (C)
    float *rsh = nsh + C;                                            // (C) list radii (EXACT)
    for (int i = threadIdx.x; i < 2 * C; i += 1024) Ash[i] = A[i];
    for (int i = threadIdx.x; i < C; i += 1024) {
        nsh[i] = cnorm[i];
        if constexpr (EXACT) {
            rsh[i] = sqrtf(__uint_as_float(r2[i * GH_IVF_CSTRIDE])) * 1.00001f;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 16 + (threadIdx.x >> 6);
    if (qi >= S) return;
    float q[LD];
#pragma unroll
    for (int d = 0; d < LD; ++d) q[d] = d < D ? qt[(int64_t)qi * QS + d] : 0.0f;
    float qn = 0.0f;
#pragma unroll
    for (int d = 0; d < LD; ++d) qn = fmaf(q[d], q[d], qn);
    uint32_t v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = j * 64 + lane;
        v[j] = 0x7F800000u;
        if (c < C) {
            float acc = nsh[c] + qn;
            const ivf_h8 lo = __builtin_bit_cast(ivf_h8, Ash[2 * c]);
#pragma unroll
            for (int d = 0; d < 8; ++d) if (d < LD) acc = fmaf((float)lo[d], q[d], acc);
            if constexpr (LD > 8) {
                const ivf_h8 hi = __builtin_bit_cast(ivf_h8, Ash[2 * c + 1]);
#pragma unroll
                for (int d = 0; d < 8; ++d) acc = fmaf((float)hi[d], q[8 + d], acc);
            }
            v[j] = __float_as_uint(fmaxf(acc, 0.0f));
        }
    }
    uint32_t took = 0;   // bit j: this lane's list j * 64 + lane is one the threshold sample may come from
    if constexpr (!EXACT) {
        // the P-th smallest centroid distance, bit by bit from the top (as gh_tau_kth).  All 31 bits: in 16 dimensions the
        // centroid distances of a query crowd together, and a band of 2^-8 around the P-th held a dozen lists more than P,
        // of which the first P in LIST order were taken -- not the nearest
        uint32_t prefix = 0;
        for (int bit = 30; bit >= 0; --bit) {
            const uint32_t t = prefix | (1u << bit);
            int below = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j) below += __popcll(__ballot(v[j] < t));
            if (below < P) prefix = t;
        }
        const uint32_t thr = prefix;
        int base = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const bool in = v[j] <= thr && j * 64 + lane < C;
            const unsigned long long b = __ballot(in);
            const int r = base + __popcll(b & ((1ull << lane) - 1ull));
            if (in && r < P) {
                took |= 1u << j;
                const int l = j * 64 + lane;
                pair_l[(int64_t)qi * P + r] = (uint32_t)l;
                pair_slot[(int64_t)qi * P + r] = (uint32_t)atomicAdd(&lqcount[l * GH_IVF_CSTRIDE], 1);
            }
            base += __popcll(b);
        }
        for (int r = base + lane; r < P; r += 64) pair_l[(int64_t)qi * P + r] = 0xFFFFFFFFu;   // fewer than P lists in all
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) took |= (j * 64 + lane < C) ? 1u << j : 0u;
    }
    // group minima over the nearest of those lists, nearest first, until tau_members members have been seen
    constexpr int NG = GH_IVF_GROUPS / 64;
    float gm[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) gm[g] = INFINITY;
    int seen = 0;
    for (int round = 0; round < C && seen < tau_members; ++round) {
        uint64_t mine = ~0ull;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const uint64_t key = ((took >> j) & 1u) ? ((uint64_t)v[j] << 32) | (uint32_t)(j * 64 + lane) : ~0ull;
            mine = key < mine ? key : mine;
        }
        const uint64_t win = ivf_wave_min_u64(mine);
        if (win == ~0ull) break;   // no list left
        const int l = (int)(uint32_t)win;
#pragma unroll
        for (int j = 0; j < NV; ++j) took &= (j * 64 + lane == l) ? ~(1u << j) : ~0u;
        const int64_t r0 = lstart[l];
        const int n = min(lcount[l], tau_members - seen);   // any subset of the probed members bounds their K-th distance from above
        for (int i0 = 0; i0 < n; i0 += 64 * NG) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int i = i0 + g * 64 + lane;
                if (i < n) {
                    float m[LD];
                    gh_load_row<LD>(lmid, r0 + i, m);
                    float d2 = 0.0f;
#pragma unroll
                    for (int d = 0; d < LD; ++d) { const float t = q[d] - m[d]; d2 = fmaf(t, t, d2); }
                    gm[g] = fminf(gm[g], d2);
                }
            }
        }
        seen += n;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) gmin[(int64_t)qi * GH_IVF_GROUPS + g * 64 + lane] = __float_as_uint(gm[g]);
    if constexpr (EXACT) {
        // tau = the K-th smallest of the group minima as a multiset: what knn_tau_kernel will extract from gmin
        uint32_t prefix = 0;
        for (int bit = 30; bit >= 0; --bit) {
            const uint32_t t = prefix | (1u << bit);
            int below = 0;
#pragma unroll
            for (int g = 0; g < NG; ++g) below += __popcll(__ballot(__float_as_uint(gm[g]) < t));
            if (below < K) prefix = t;
        }
        const float st = sqrtf(__uint_as_float(prefix)) * 1.00001f;   // inf when the sample held fewer than K members
        // Second bound, from the assignment itself: a member x is filed under the centroid with the smallest (f16) score, so
        // |x - c_l|^2 <= |x - c*|^2 + e for ANY other centroid c* -- take the one nearest to q, at distance <= dstar -- and with
        // |q - x| <= r:  |q - c_l| <= r + |x - c_l| <= r + sqrt((r + dstar)^2 + e_l).  It does not know the list's radius, which
        // one far-out member of a list at the rim of the cloud makes several times the list's typical size.
        // e_l: what two f16 scores can differ from the truth, <= 2^-10 (|c|^2 + |m|^2) each, with |m| <= |q| + r; doubled.
        uint64_t nearest = ~0ull;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = j * 64 + lane;
            const bool ok = c < C && nsh[c < C ? c : 0] <= 15000.0f * 15000.0f;   // its operand row is the centroid, not a clamp
            const uint64_t key = ok ? ((uint64_t)v[j] << 32) | (uint32_t)c : ~0ull;
            nearest = key < nearest ? key : nearest;
        }
        nearest = ivf_wave_min_u64(nearest);
        float dstar = INFINITY, cstar = 0.0f;
        if (nearest != ~0ull) {
            cstar = nsh[(uint32_t)nearest];
            const float a2 = __uint_as_float((uint32_t)(nearest >> 32));
            dstar = sqrtf(a2 + (9.765625e-4f * (cstar + qn) + 1e-6f * sqrtf(cstar + qn))) * 1.00001f;
        }
        const float mmax = sqrtf(qn) + st;   // |m| of a member within r of q
        uint32_t in_mask = 0;
        int total = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = j * 64 + lane;
            bool in = false;
            if (c < C) {
                const float cn = nsh[c];
                // what the f16 centroid dropped from the distance above: <= 2^-11 * 2 |c| |q| (+ subnormal and accumulation terms)
                const float e = 9.765625e-4f * (cn + qn) + 1e-6f * sqrtf(cn + qn);
                const float lb2 = fmaxf(__uint_as_float(v[j]) - e, 0.0f);
                const float el = 3.90625e-3f * (cn + cstar + 2.0f * mmax * mmax) + 4e-6f * sqrtf(cn + cstar + 2.0f * mmax * mmax);
                const float R = rsh[c];
                const float reach = fminf(R, sqrtf((st + dstar) * (st + dstar) + el) * 1.00001f);
                const float rhs = st + reach;
                in = !(lb2 > rhs * rhs * 1.00001f) || !(R < INFINITY);   // (no finite radius: a clamped member or centroid -- no bound at all)
            }
            in_mask |= in ? 1u << j : 0u;
            total += __popcll(__ballot(in));
        }
        if (total > P) {   // too many: the exhaustive search takes the query
            if (lane == 0) atomicAdd(&cnt[(int64_t)qi * GH_CNT_STRIDE], GH_CAND_CAP + 1);
            total = 0;
            in_mask = 0;
        }
        int base = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const bool in = (in_mask >> j) & 1u;
            const unsigned long long b = __ballot(in);
            const int r = base + __popcll(b & ((1ull << lane) - 1ull));
            if (in) {
                const int l = j * 64 + lane;
                pair_l[(int64_t)qi * P + r] = (uint32_t)l;
                pair_slot[(int64_t)qi * P + r] = (uint32_t)atomicAdd(&lqcount[l * GH_IVF_CSTRIDE], 1);
            }
            base += __popcll(b);
        }
        for (int r = base + lane; r < P; r += 64) pair_l[(int64_t)qi * P + r] = 0xFFFFFFFFu;
    }
}

__global__ __launch_bounds__(256) void ivf_pair_scatter_kernel(const uint32_t *__restrict__ pair_l, const uint32_t *__restrict__ pair_slot,
                                                               int64_t npairs, int P, const int32_t *__restrict__ qstart,
                                                               int32_t *__restrict__ pair_q) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    const uint32_t l = pair_l[i];
    if (l != 0xFFFFFFFFu) pair_q[qstart[l] + pair_slot[i]] = (int32_t)(i / P);
}

// One tile of one list against the queries that probe the list: the filtered scan of scan_core.h (pre-filter in
// norm-expansion form on packed fp32, exact fma chain on what passes, hits parked in LDS).
template <int D>
__global__ __launch_bounds__(256) void ivf_scan_kernel(const float *__restrict__ lmid, const uint32_t *__restrict__ lids,
                                                       const int32_t *__restrict__ tile_list, const int32_t *__restrict__ meta,
                                                       const int32_t *__restrict__ qstart, const int32_t *__restrict__ pair_q,
                                                       const float *__restrict__ qt, const float *__restrict__ qscan,
                                                       uint64_t *__restrict__ cand, int32_t *__restrict__ cnt) {
    constexpr int R = GH_IVF_TILE / 256;
    constexpr int LD = D <= 4 ? 4 : D <= 8 ? 8 : 16;
    constexpr int QS = D <= 3 ? 4 : LD + 4;
    constexpr int QT = D <= 3 ? 3 : LD;
    __shared__ float4 qsh[(GH_SCAN_QGROUP + 1) * (QS / 4)];
    __shared__ float taush[GH_SCAN_QGROUP];
    __shared__ int qmap[GH_SCAN_QGROUP];
    __shared__ uint64_t hkey[1024];
    __shared__ int hq[1024];
    __shared__ int hcount;
    const int tile = blockIdx.x;
    if (tile >= meta[0]) return;
    const int l = tile_list[tile];
    const int q0 = qstart[l], q1 = qstart[l + 1];
    if (q0 == q1) return;
    gh_f2 m[R / 2][D], c0[R / 2];
    uint32_t id[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t j = (int64_t)tile * GH_IVF_TILE + r * 256 + threadIdx.x;
        id[r] = lids[j];
        const bool valid = id[r] != 0xFFFFFFFFu;
        float mv[LD];
#pragma unroll
        for (int d = 0; d < LD; ++d) mv[d] = 0.0f;
        if (valid) gh_load_row<LD>(lmid, j, mv);
        const float c = gh_ref_c0<D>(mv, valid);
        if (r & 1) c0[r / 2].y = c; else c0[r / 2].x = c;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (r & 1) m[r / 2][d].y = mv[d];
            else m[r / 2][d].x = mv[d];
        }
    }
    for (int qb = q0; qb < q1; qb += GH_SCAN_QGROUP) {
        const int nq = min(q1 - qb, GH_SCAN_QGROUP);
        __syncthreads();
        if (threadIdx.x == 0) hcount = 0;
        for (int i = threadIdx.x; i < (nq + 1) * (QS / 4); i += 256) {
            const int s = i / (QS / 4), p = i % (QS / 4);
            qsh[i] = s < nq ? reinterpret_cast<const float4 *>(qscan)[(int64_t)pair_q[qb + s] * (QS / 4) + p] : make_float4(0.f, 0.f, 0.f, -1.f);
        }
        for (int i = threadIdx.x; i < nq; i += 256) {
            const int s = pair_q[qb + i];
            qmap[i] = s;
            taush[i] = qt[(int64_t)s * QS + QT];
        }
        __syncthreads();
        gh_scan_queries<D, R, 1024>(m, c0, id, qsh, nq, 0, taush, hkey, hq, &hcount, cand, cnt, qmap);
        __syncthreads();
        gh_flush_hits<1024>(hkey, hq, &hcount, cand, cnt);
    }
}

// The same tile-of-a-list x queries-of-the-list search with the pre-filter on the MATRIX pipe, 4 <= D <= 16: the wide form of
// scan_core.h (F = C0 - 2 q.m - T <= 0 from single-piece f16 coordinates and three-piece constants, conservative; the
// decision is taken on the exact fma chain), organised as phase B of the fused spring+scan kernel (fused.hip): each wave
// owns 4 column blocks of 32 members whose operands stay in registers, the list's queries stream past as A operands from
// LDS, pairs that pass are listed and decided by all threads after the group's last matrix instruction.  The fp32 tile
// and query records stay in LDS for those exact checks.  (Packed fp32 VALU: D / 2 instructions per pair and lane.)
template <int D, int LD>
__global__ __launch_bounds__(256) void ivf_scan_mfma_kernel(const float *__restrict__ lmid, const uint32_t *__restrict__ lids,
                                                            const int32_t *__restrict__ tile_list, const int32_t *__restrict__ meta,
                                                            const int32_t *__restrict__ qstart, const int32_t *__restrict__ pair_q,
                                                            const float *__restrict__ qt, const float *__restrict__ qscan,
                                                            uint64_t *__restrict__ cand, int32_t *__restrict__ cnt) {
    constexpr int NT = 256, R = GH_IVF_TILE / NT, TILE = GH_IVF_TILE, NB = 2 * R;
    constexpr int HITBUF = LD == 16 ? 256 : 512;
    constexpr int KB = D <= 10 ? 1 : 2;
    constexpr int QS = LD + 4, QT = LD;
    constexpr int PENDCAP = 512;
    __shared__ float4 tile[TILE * LD / 4];
    __shared__ gh_h8 qa[GH_SCAN_QGROUP * KB * 2];
    __shared__ float4 qsh[(GH_SCAN_QGROUP + 1) * (QS / 4)];
    __shared__ float taush[GH_SCAN_QGROUP];
    __shared__ int qmap[GH_SCAN_QGROUP];
    __shared__ uint64_t hkey[HITBUF];
    __shared__ int hq[HITBUF];
    __shared__ uint16_t badlist[TILE];
    __shared__ uint32_t ids[TILE];
    __shared__ uint16_t exq[GH_SCAN_QGROUP];
    __shared__ uint32_t pend[PENDCAP];
    __shared__ int hcount, nbad, nexq, npend;
    const int t_idx = blockIdx.x;
    if (t_idx >= meta[0]) return;
    const int l = tile_list[t_idx];
    const int q0 = qstart[l], q1 = qstart[l + 1];
    if (q0 == q1) return;
    float *mids = reinterpret_cast<float *>(tile);
    if (threadIdx.x == 0) { hcount = 0; nbad = 0; nexq = 0; npend = 0; }
    for (int j = threadIdx.x; j < TILE; j += NT) ids[j] = lids[(int64_t)t_idx * TILE + j];
    {
        const float4 *src = reinterpret_cast<const float4 *>(lmid) + (int64_t)t_idx * TILE * (LD / 4);
        for (int i = threadIdx.x; i < TILE * LD / 4; i += NT) tile[i] = src[i];   // (padding rows are never read as members)
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = lane & 31, hsel = lane >> 5;
    gh_h8 B[NB][KB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = w * (64 * R) + b * 32 + col;
        const bool valid = ids[j] != 0xFFFFFFFFu;
        float mv[LD];
        gh_load_row<LD>(mids, valid ? j : 0, mv);
        if (!gh_mfw_ref_col<D, KB>(mv, valid, hsel, B[b]) && hsel == 0) badlist[atomicAdd(&nbad, 1)] = (uint16_t)j;
    }
    auto park = [&](int s, int j) {  // exact decision on pair (query s of the group, member j of the tile)
        const float *qv = reinterpret_cast<const float *>(qsh) + s * QS;   // (-2q_0 .. -2q_{D-1}, .., t)
        float mv[LD];
        gh_load_row<LD>(mids, j, mv);
        float d2 = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float df = -0.5f * qv[d] - mv[d];   // the record holds -2q: exact both ways
            d2 = fmaf(df, df, d2);
        }
        if (d2 <= taush[s]) {
            const uint32_t id = ids[j];
            const int p = atomicAdd(&hcount, 1);
            if (p < HITBUF) { hkey[p] = gh_key(d2, id); hq[p] = qmap[s]; }
            else gh_append_candidate(cand, cnt, qmap[s], gh_key(d2, id));
        }
    };
    const gh_f16x zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int qb0 = q0; qb0 < q1; qb0 += GH_SCAN_QGROUP) {
        const int nq = min(q1 - qb0, GH_SCAN_QGROUP);
        __syncthreads();  // the previous group's rows and lists are still being read (first group: the operands above)
        if (qb0 > q0) {
            const bool flush = hcount >= HITBUF / 4;
            if (flush) gh_flush_hits<HITBUF, NT>(hkey, hq, &hcount, cand, cnt);
            __syncthreads();
            if (threadIdx.x == 0) { nexq = 0; npend = 0; if (flush) hcount = 0; }
            __syncthreads();
        }
        for (int i = threadIdx.x; i < (nq + 1) * (QS / 4); i += NT) {
            const int s = i / (QS / 4), p = i % (QS / 4);
            qsh[i] = s < nq ? reinterpret_cast<const float4 *>(qscan)[(int64_t)pair_q[qb0 + s] * (QS / 4) + p] : make_float4(0.f, 0.f, 0.f, -1.f);
        }
        {
            const int q = threadIdx.x;   // one query per thread: its A row from the query record and the threshold
            _Float16 row[16 * KB];
            if (q < nq) {
                const int sq = pair_q[qb0 + q];
                const float4 *src = reinterpret_cast<const float4 *>(qscan) + (int64_t)sq * (QS / 4);
                float qv[16];
#pragma unroll
                for (int d = 0; d < 16; ++d) qv[d] = 0.0f;
#pragma unroll
                for (int i = 0; i < LD / 4; ++i) {
                    const float4 v = src[i];
                    qv[4 * i] = -0.5f * v.x; qv[4 * i + 1] = -0.5f * v.y; qv[4 * i + 2] = -0.5f * v.z; qv[4 * i + 3] = -0.5f * v.w;
                }
                const float tau = qt[(int64_t)sq * QS + QT];
                qmap[q] = sq;
                taush[q] = tau;
                if (!gh_mfw_query_row<KB>(qv, tau, row)) exq[atomicAdd(&nexq, 1)] = (uint16_t)q;
            } else {  // padding row: never passes
#pragma unroll
                for (int k = 0; k < 16 * KB; ++k) row[k] = (_Float16)0.0f;
                row[gh_mfw<KB>::base + 3] = (_Float16)GH_MF_NEVER;
            }
#pragma unroll
            for (int i = 0; i < 2 * KB; ++i) {
                gh_h8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = row[i * 8 + e];
                qa[q * (2 * KB) + i] = hv;
            }
        }
        __syncthreads();
        const int nqb = (nq + 31) / 32;
        for (int qb = 0; qb < nqb; ++qb) {
            gh_h8 a[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) a[kb] = qa[((qb * 32 + col) * KB + kb) * 2 + hsel];
            auto tile_of = [&](int b) {
                gh_f16x f = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], B[b][0], zero, 0, 0, 0);
                if constexpr (KB == 2) f = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], B[b][1], f, 0, 0, 0);
                return f;
            };
            gh_f16x f = tile_of(0);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                gh_f16x fn = zero;
                if (b + 1 < NB) fn = tile_of(b + 1);   // in the matrix pipe while block b is tested
                int mn = min(__float_as_int(f[0]), __float_as_int(f[1]));
#pragma unroll
                for (int i = 2; i < 16; ++i) mn = min(mn, __float_as_int(f[i]));
                asm volatile("" : "+v"(mn));
                if (mn <= 0) {  // rare: result row (i&3) + 8(i>>2) + 4*half, column = this lane's member
                    uint32_t m = 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) m = __builtin_amdgcn_alignbit(m, __float_as_uint(f[i]), 31);
                    const int j = w * (64 * R) + b * 32 + col;
                    while (m) {
                        const int bit = 31 - __builtin_clz(m);
                        m &= ~(1u << bit);
                        const int i = 15 - bit;
                        const int s = qb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hsel;
                        if (s < nq) {
                            const int p = atomicAdd(&npend, 1);
                            if (p < PENDCAP) pend[p] = ((uint32_t)s << 16) | (uint32_t)j;
                            else park(s, j);   // list full: decided on the spot
                        }
                    }
                }
                f = fn;
            }
        }
        __syncthreads();
        for (int p = threadIdx.x, np = min(npend, PENDCAP); p < np; p += NT) park((int)(pend[p] >> 16), (int)(pend[p] & 0xFFFFu));
        // outside the f16 range: exact scan of this group's listed queries over the whole tile ...
        const int nex = nexq;
        for (int x = 0; x < nex; ++x) {
            const int s = exq[x];
            for (int j = threadIdx.x; j < TILE; j += NT)
                if (ids[j] != 0xFFFFFFFFu) park(s, j);
        }
        // ... and of the tile's out-of-range members against every other query of the group
        const int nb_ = nbad;
        for (int p = threadIdx.x; p < nb_ * nq; p += NT) {
            const int s = p % nq;
            const gh_h8 tv = qa[(s * KB + (gh_mfw<KB>::base + 3) / 16) * 2 + ((gh_mfw<KB>::base + 3) % 16) / 8];
            if ((float)tv[(gh_mfw<KB>::base + 3) % 8] == GH_MF_NEVER) continue;  // a listed query: the loop above has done this pair
            park(s, badlist[p / nq]);
        }
    }
    __syncthreads();
    gh_flush_hits<HITBUF, NT>(hkey, hq, &hcount, cand, cnt);
}

size_t ivf_align(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

bool gh_ivf_path(const gh_engine *h) {
    return h->prm.knn_method == GH_KNN_IVF && !h->cdist && h->D >= 2 && h->LD <= 16 && gh_knn_scan_path(h) &&
           h->own_count >= 64 * 64 && h->own_count < ((int64_t)1 << 31) - (int64_t)GH_IVF_MAX_LISTS * GH_IVF_TILE;
}

void gh_ivf_free(gh_engine *h) {
    if (!h->ivf) return;
    if (h->ivf->blob) (void)hipFree(h->ivf->blob);
    delete h->ivf;
    h->ivf = nullptr;
}

gh_status gh_ivf_alloc(gh_engine *h) {
    if (!gh_ivf_path(h)) return GH_OK;
    gh_ivf *v = new gh_ivf();
    h->ivf = v;
    const int64_t M = h->own_count;
    v->M = M;
    // lists: about sqrt(M) / 2, a multiple of 64, at least 64 members on average (the assignment costs M * C score
    // evaluations: 4 M midpoints, 1984 lists 520 us, 1024 lists 270 us, and the same recall for the same scan work);
    // probes unless told otherwise: a sixteenth of the lists for more than 8 components (4 M midpoints, 1024 lists: recall
    // 0.989 - 0.992 in 16 dimensions), a thirty-second for 5 - 8 and a sixty-fourth below (6 components: 8 of 1024 lists 0.995,
    // 16 of 1984 0.999, 32 of 1984 1.0000; 4 components: 8 of 1984 0.997)
    // default: at most 512 lists up to 4 components, 1024 above (the scan's share shrinks like S / C while the assignment grows
    // like C: 16 M midpoints, 3 components, 16 K queries: 1984 lists 4.1 ms per iteration, 1024 2.95, 512 2.76)
    int64_t C = h->prm.ivf_lists > 0 ? h->prm.ivf_lists
                                     : std::min<int64_t>(h->D <= 4 ? 512 : 1024, (int64_t)std::llround(std::sqrt((double)M) / 2.0));
    C = std::min<int64_t>(C, M / 64);
    C = std::max<int64_t>(64, std::min<int64_t>(GH_IVF_MAX_LISTS, (C + 32) / 64 * 64));
    int64_t P = h->prm.ivf_probes > 0 ? h->prm.ivf_probes : C / (h->D > 8 ? 16 : h->D > 4 ? 32 : 64);
    P = std::max<int64_t>(1, std::min<int64_t>(P, C));
    v->exact = h->prm.ivf_probes < 0;
    // exact mode: room for every list (a query far out, or many components: the ball reaches most of them -- it then costs
    // what a brute-force scan costs, not the single-workgroup exhaustive search), within 1 GiB per pair array
    if (v->exact) P = std::min<int64_t>(C, std::max<int64_t>(64, ((int64_t)1 << 28) / std::max<int64_t>(1, h->S)));
    v->C = (int)C;
    v->P = (int)P;
    v->cap_rows = M + C * GH_IVF_TILE;
    v->share = ((M + GH_IVF_ASSIGN_WGS - 1) / GH_IVF_ASSIGN_WGS + 511) / 512 * 512;
    v->awgs = (int)((M + v->share - 1) / v->share);
    v->max_tiles = v->cap_rows / GH_IVF_TILE + 1;
    const size_t S = (size_t)h->S, LD = (size_t)h->LD;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += ivf_align(bytes); return o; };
    const size_t o_cent = take(4 * C * LD), o_A = take(32 * C), o_cn = take(4 * C), o_as = take(4 * M), o_rk = take(4 * M),
                 o_sc = take(4 * C * GH_IVF_CSTRIDE), o_r2 = take(v->exact ? 4 * C * GH_IVF_CSTRIDE : 16), o_ss = take(4 * (size_t)v->awgs * C), o_lc = take(4 * C), o_ls = take(4 * (C + 1)), o_tl = take(4 * v->max_tiles), o_me = take(16),
                 o_lm = take(4 * v->cap_rows * LD), o_li = take(4 * v->cap_rows), o_lq = take(4 * C * GH_IVF_CSTRIDE), o_qs = take(4 * (C + 1)),
                 o_pl = take(4 * S * P), o_ps = take(4 * S * P), o_pq = take(4 * S * P);
    if (hipMalloc(reinterpret_cast<void **>(&v->blob), off) != hipSuccess) {
        h->err = "hipMalloc of the IVF buffers failed";
        return GH_ERR_NOMEM;
    }
    if (hipMemset(v->blob, 0, off) != hipSuccess) { h->err = "hipMemset of the IVF buffers failed"; return GH_ERR_HIP; }
    unsigned char *b = v->blob;
    v->cent = reinterpret_cast<float *>(b + o_cent);
    v->A = reinterpret_cast<uint4 *>(b + o_A);
    v->cnorm = reinterpret_cast<float *>(b + o_cn);
    v->assign = reinterpret_cast<uint32_t *>(b + o_as);
    v->rank = reinterpret_cast<uint32_t *>(b + o_rk);
    v->count = reinterpret_cast<int32_t *>(b + o_sc);
    v->r2 = v->exact ? reinterpret_cast<uint32_t *>(b + o_r2) : nullptr;
    v->wgbase = reinterpret_cast<int32_t *>(b + o_ss);
    v->lcount = reinterpret_cast<int32_t *>(b + o_lc);
    v->lstart = reinterpret_cast<int32_t *>(b + o_ls);
    v->tile_list = reinterpret_cast<int32_t *>(b + o_tl);
    v->meta = reinterpret_cast<int32_t *>(b + o_me);
    v->lmid = reinterpret_cast<float *>(b + o_lm);
    v->lids = reinterpret_cast<uint32_t *>(b + o_li);
    v->lqcount = reinterpret_cast<int32_t *>(b + o_lq);
    v->qstart = reinterpret_cast<int32_t *>(b + o_qs);
    v->pair_l = reinterpret_cast<uint32_t *>(b + o_pl);
    v->pair_slot = reinterpret_cast<uint32_t *>(b + o_ps);
    v->pair_q = reinterpret_cast<int32_t *>(b + o_pq);
    return GH_OK;
}

extern "C" gh_status gh_knn_ivf_list_sizes(gh_handle h, int32_t *sizes, int32_t count) {
    if (!h) return GH_ERR_INVALID;
    if (!h->ivf || !sizes || count != h->ivf->C) { h->err = "gh_knn_ivf_list_sizes: not a GH_KNN_IVF engine, or count != lists"; return GH_ERR_INVALID; }
    GH_HIP(hipStreamSynchronize(h->stream));
    GH_HIP(hipMemcpy(sizes, h->ivf->lcount, sizeof(int32_t) * (size_t)count, hipMemcpyDeviceToHost));
    return GH_OK;
}

extern "C" gh_status gh_knn_ivf_config(gh_handle h, int32_t *lists, int32_t *probes) {
    if (!h) return GH_ERR_INVALID;
    if (lists) *lists = h->ivf ? h->ivf->C : 0;
    if (probes) *probes = h->ivf ? h->ivf->P : 0;
    return GH_OK;
}

// d_mid (this iteration's own midpoints) + the query records -> tau of every query and its candidate list.
gh_status gh_ivf_search(gh_engine *h) {
    gh_ivf *v = h->ivf;
    const int64_t M = v->M;
    const int C = v->C, P = v->P, QS = gh_qs(h->D, h->LD), S = (int)h->S;
    // members the threshold is taken from: the K-th smallest of an m-sample of the N probed members sits near rank K * N / m
    // of them, which is what the filtered scan then lets through per query: m = N / 16 above 8 components (N / 64 left the
    // queries next to the crowded lists of a 16-dimensional cloud with more than the 16384 keys a candidate list holds), N / 64
    // below (there the nearest lists hold the nearest members, and at 16 K queries the sample was the probe kernel's 0.55 ms);
    // 1024 ... 8192
    const int64_t probed = (int64_t)P * (M / C);
    const int tau_members = v->exact ? std::max(4096, 16 * h->Ksel) /* a tight threshold keeps the ball small */ : (int)std::min<int64_t>(8192, std::max<int64_t>(std::max(1024, 16 * h->Ksel), probed / (h->D > 8 ? 16 : 64)));
#define GH_IVF_LD(X)                          \
    switch (h->LD) {                          \
        case 4: { X(4) } break;               \
        case 8: { X(8) } break;               \
        default: { X(16) } break;             \
    }
    {
        gh_scope t(h, "ivf_build");
        GH_HIP(hipMemsetAsync(v->lids, 0xFF, sizeof(uint32_t) * (size_t)v->cap_rows, h->stream));
#define GH_X(L) ivf_centroid_kernel<L><<<dim3((unsigned)((C + 255) / 256)), dim3(256), 0, h->stream>>>(h->d_mid, M, C, v->cent, v->A, v->cnorm, v->count, v->lqcount, v->r2);
        GH_IVF_LD(GH_X)
#undef GH_X
        GH_LAUNCH_CHECK();
    }
    {
        gh_scope t(h, "ivf_assign");
        const size_t lds = (size_t)C * 44;
#define GH_X(L)                                                                                                                                  \
    if (lds > 48 * 1024) GH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ivf_assign_kernel<L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    ivf_assign_kernel<L><<<dim3((unsigned)v->awgs), dim3(256), lds, h->stream>>>(h->d_mid, M, v->share, C, v->A, v->cnorm, v->assign, v->rank, v->count, v->wgbase, v->r2);
        GH_IVF_LD(GH_X)
#undef GH_X
        GH_LAUNCH_CHECK();
    }
    {
        gh_scope t(h, "ivf_layout");
        ivf_list_layout_kernel<<<dim3(1), dim3(1024), 0, h->stream>>>(v->count, C, v->lcount, v->lstart, v->tile_list, v->meta);
        const int Q = h->LD / 4;
#define GH_X(L) ivf_scatter_kernel<L><<<dim3((unsigned)((M * Q + 255) / 256)), dim3(256), 0, h->stream>>>(h->d_mid, M, v->assign, v->rank, v->lstart, v->wgbase, v->share, C, h->part.edge_lo, h->d_own_eids, v->lmid, v->lids);
        GH_IVF_LD(GH_X)
#undef GH_X
        GH_LAUNCH_CHECK();
    }
    {
        gh_scope t(h, "ivf_probe");
        const unsigned pb = (unsigned)((S + 15) / 16);
        const int nv = (C + 63) / 64;
        const size_t lds = (size_t)C * 40;
#define GH_PROBE2(L, NVv, EX)                                                                                                                         \
    {                                                                                                                                                 \
        if (lds > 48 * 1024) GH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ivf_probe_kernel<L, NVv, EX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        ivf_probe_kernel<L, NVv, EX><<<dim3(pb), dim3(1024), lds, h->stream>>>(h->d_q, QS, S, h->D, v->A, v->cnorm, C, P, h->Ksel, v->lstart, v->lcount, v->lmid, \
                                                                              tau_members, v->r2, v->pair_l, v->pair_slot, v->lqcount,                \
                                                                              reinterpret_cast<uint32_t *>(h->d_gmin), h->d_cnt);                    \
    }
#define GH_PROBE(L, NVv)                      \
    if (v->exact) GH_PROBE2(L, NVv, true)     \
    else GH_PROBE2(L, NVv, false)
#define GH_X(L)                               \
    if (nv <= 8) { GH_PROBE(L, 8) }           \
    else if (nv <= 16) { GH_PROBE(L, 16) }    \
    else { GH_PROBE(L, 32) }
        GH_IVF_LD(GH_X)
#undef GH_X
#undef GH_PROBE2
#undef GH_PROBE
        ivf_query_layout_kernel<<<dim3(1), dim3(1024), 0, h->stream>>>(v->lqcount, C, v->qstart);
        const int64_t npairs = (int64_t)S * P;
        ivf_pair_scatter_kernel<<<dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, h->stream>>>(v->pair_l, v->pair_slot, npairs, P, v->qstart, v->pair_q);
        GH_LAUNCH_CHECK();
    }
    GH_TRY_ST(gh_knn_thresholds(h, GH_IVF_GROUPS));
    {
        gh_scope t(h, "ivf_scan");
        const dim3 grid((unsigned)v->max_tiles);
#define GH_X(DD) ivf_scan_kernel<DD><<<grid, dim3(256), 0, h->stream>>>(v->lmid, v->lids, v->tile_list, v->meta, v->qstart, v->pair_q, h->d_q, h->d_qscan, h->d_cand, h->d_cnt)
#define GH_XM(DD, LL) ivf_scan_mfma_kernel<DD, LL><<<grid, dim3(256), 0, h->stream>>>(v->lmid, v->lids, v->tile_list, v->meta, v->qstart, v->pair_q, h->d_q, h->d_qscan, h->d_cand, h->d_cnt)
        if (h->D >= 4) {
            switch (h->D) {
                case 4: GH_XM(4, 4); break;
                case 5: GH_XM(5, 8); break;
                case 6: GH_XM(6, 8); break;
                case 7: GH_XM(7, 8); break;
                case 8: GH_XM(8, 8); break;
                case 9: GH_XM(9, 16); break;
                case 10: GH_XM(10, 16); break;
                case 11: GH_XM(11, 16); break;
                case 12: GH_XM(12, 16); break;
                case 13: GH_XM(13, 16); break;
                case 14: GH_XM(14, 16); break;
                case 15: GH_XM(15, 16); break;
                default: GH_XM(16, 16); break;
            }
        } else {
            switch (h->D) {
                case 2: GH_X(2); break;
                case 3: GH_X(3); break;
                case 4: GH_X(4); break;
                default:
                    if (h->LD == 8) GH_X(8);
                    else GH_X(16);
            }
        }
#undef GH_XM
#undef GH_X
        GH_LAUNCH_CHECK();
    }
#undef GH_IVF_LD
    return GH_OK;
}
