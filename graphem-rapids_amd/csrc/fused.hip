// Fused spring + KNN scan: the dominant kernel of an iteration.
//
// Why fused: the spring pull is bound by the chip's rate for random row requests (8 neighbour rows
// of 16 bytes per vertex, ~65-73 G rows/s whatever the byte rate), the KNN scan by the matrix pipe /
// the fp32 VALU.  As separate kernels they run back to back (146 + 110 us on the 1M-vertex graph) and
// the midpoints make a round trip through memory.  Here one workgroup owns a range of consecutive own
// rows holding at most TILE owned edges (an edge is owned by ONE of its endpoints -- by a hash of the
// edge id, or by endpoint 0 for range partitions; bit 31 of a pull-list entry marks the edges a row owns):
//   phase A  spring pull of its vertices (reference pt.py:595-636, summation order of the
//            two index_add_ calls) -> Fs; the midpoints (pt.py:785) of the owned edges fall
//            out of the gathered rows and are kept in LDS, never written to memory;
//   phase B  those midpoints become the workgroup's reference tile: R per thread in packed
//            registers, all S queries stream past (scan_core.h).
// Workgroups in phase A (waiting on gathers) and workgroups in phase B (issuing VALU) share
// the CUs, so the two bounds overlap instead of adding.
#include "common.h"
#include "engine.h"
#include "scan_core.h"
#include "tau_core.h"

#include <stdio.h>
#include <stdlib.h>

// Fused-kernel geometry: 256 threads x 2 references per thread = tiles of 512 owned edges per workgroup.
// (Measured on the 1M-vertex graph, D = 3, before the matrix-pipe filter: 256x8 225 us, 256x4 197 us -- a smaller tile costs
// more filter work per pair but less LDS per workgroup, so more workgroups are resident and their gather / filter phases
// interleave better.)
// The pre-filter of phase B runs on the matrix pipe for every fused dimension: split-f16 operands for D <= 3 (1024 pairs per
// matrix instruction), single-piece f16 operands for 4 <= D <= 16 (scan_core.h).  Against the packed-fp32 VALU form it
// replaced (rounds 1-2; removed in round 4 together with its GRAPHEM_HIP_MFMA / _FUSED_CFG switches): 1M vertices, 256
// queries: 4700 vs 4440 it/s in the steady state of a long run; 100K vertices 10760 vs 9730; 1024 queries 433 vs 566 us per
// iteration, 4096 queries 640 vs 1434 us for the kernel; D = 12: 818 -> 508 us per iteration.
static bool fused_mfma(int LD, int D, int64_t S) {
    (void)S;
    return D <= 16 && (LD == 4 || LD == 8 || LD == 16);   // D <= 3: split form; 4..16: wide form (scan_core.h)
}
// operand rows the threshold computation writes: only the split form has any (the wide form builds its rows at staging)
static int fused_mfma_kb(int LD, int D, int64_t S) { return fused_mfma(LD, D, S) && D <= 3 ? 0 : -1; }
// Workgroups of 256 threads, R = 2 references per thread: tiles of 512 owned edges (95 VGPRs / 27 KB of LDS for the split
// form; the wide form keeps the fp32 tile in LDS for the exact checks).
// (Round 5, for graphs whose launch is one under-filled round of workgroups -- 100 K vertices: 782 workgroups on 256 CUs --
// tiles of 256 edges, R = 1, twice the workgroups: 59.1 us per iteration against 54.5, the fused kernel 32.4 against 28.0.)
static void fused_cfg(int LD, int D, int64_t S, int64_t own_edges, int *nt, int *r) {
    (void)LD; (void)D; (void)S; (void)own_edges;
    *nt = 256;
    *r = 2;
}
bool gh_fused_uses_mfma(const gh_engine *h) { return h->fused_scan && fused_mfma(h->LD, h->D, h->S); }
int gh_fused_mfma_kb(const gh_engine *h) { return h->fused_scan ? fused_mfma_kb(h->LD, h->D, h->S) : -1; }
int gh_fused_tile(const gh_engine *h) {
    int nt, r;
    fused_cfg(h->LD, h->D, h->S, h->own_count, &nt, &r);
    return nt * r;
}

namespace {

// Phase A of both fused kernels for one workgroup: spring forces of the vertices v0..v1 -> Fs,
// midpoints of their owned edges -> LDS, and new0 = pos + Fs -> out_new (pt.py:796-799 for every
// vertex the intersection phase does not touch: Fs + 0 == Fs exactly; the few touched vertices are
// redone by stats_fix_kernel).  Returns this thread's column sums of new0 for the fp64 statistics.
template <int D, int LD, int NT, bool LONG>
__device__ __forceinline__ void gh_phase_a(const float *__restrict__ pos, const int32_t *__restrict__ rowptr,
                                           const int32_t *__restrict__ adj, const int32_t *__restrict__ first_edge,
                                           int v0, int v1, int fe0, int nedges, int64_t row_lo, float L_min, float neg_k,
                                           float *__restrict__ Fs, float *__restrict__ out_new, float *mids,
                                           double (&sx)[LD], double (&sxx)[LD], const gh_long_args &la,
                                           const float *__restrict__ Fpre = nullptr /* hub forces (the Fs array) */) {
    if (!Fpre) Fpre = Fs;
#pragma unroll
    for (int d = 0; d < LD; ++d) { sx[d] = 0.0; sxx[d] = 0.0; }
    for (int i = v0 + threadIdx.x; i < v1; i += NT) {
        const int64_t x = row_lo + i;
        float px[LD], F[LD], nw[LD];
        gh_load_row<LD>(pos, x, px);
        // LONG: the graph has hub rows (their forces come from spring_long_kernel); kept out of the
        // common instantiation, where the extra path costs 8 VGPRs and with them a wave of occupancy
        if constexpr (LONG)
            spring_row<D, LD, true>(pos, adj, rowptr[i], rowptr[i + 1], x, px, L_min, neg_k, F, mids, first_edge[i] - fe0,
                                    la, i, Fpre + (int64_t)i * LD);
        else
            spring_pull<D, LD, true>(pos, adj, rowptr[i], rowptr[i + 1], x, px, L_min, neg_k, F, mids, first_edge[i] - fe0);
        if (Fs) gh_store_row<LD>(Fs, i, F);
#pragma unroll
        for (int d = 0; d < LD; ++d) {
            nw[d] = px[d] + F[d];
            sx[d] += (double)nw[d];
            sxx[d] += (double)nw[d] * (double)nw[d];
        }
        if (out_new) gh_store_row<LD>(out_new, i, nw);
    }
    if constexpr (LONG)
        if (la.n > 0 && la.own_long) gh_long_midpoints<D, LD, NT>(pos, la, fe0, nedges, mids);
}

// Workgroup reduction of the per-thread column sums -> blockstats[entry][blockIdx.x], entry < 2*LD (fixed order).
template <int LD, int NT>
__device__ __forceinline__ void gh_block_stats(const double (&sx)[LD], const double (&sxx)[LD], double *red /* [NT/64][2*LD] */,
                                               double *__restrict__ blockstats, int bx, int nbx, int nrows) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // a wave none of whose threads had a row (a tile of 512 owned edges is ~128 rows for 256 threads) holds zeros: it
    // stores them and skips the 2 LD shuffle trees -- 240 instructions, half of them LDS permutes, per idle wave (round 3:
    // the trees of all four waves were 4 M of the kernel's 42 M VALU instructions and most of its LDS instructions)
    if (w * 64 >= nrows) {
        if (lane < 2 * LD) red[w * 2 * LD + lane] = 0.0;
    } else {
#pragma unroll
        for (int d = 0; d < LD; ++d) {
            const double a = gh_wave_sum(sx[d]), b = gh_wave_sum(sxx[d]);
            if (lane == 0) { red[w * 2 * LD + d] = a; red[w * 2 * LD + LD + d] = b; }
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * LD && blockstats) {
        double v = red[threadIdx.x];
#pragma unroll
        for (int ww = 1; ww < NT / 64; ++ww) v += red[ww * 2 * LD + threadIdx.x];
        blockstats[(int64_t)threadIdx.x * nbx + bx] = v;  // [entry][workgroup]: the reducers read contiguously
    }
}


// MFMA form of phase B (D <= 3, 256 threads, tiles of 256*R edges): the pre-filter of scan_core.h
// on v_mfma_f32_32x32x16_f16 with split-f16 operands.  Each wave owns 64*R references of the tile
// as NB = 2R column blocks of 32 whose B operands live in registers for the whole scan; the
// queries stream past as A operands, 32 per MFMA, read from LDS.  A lane owns one reference column
// and 16 query rows of every 32x32 result: a min tree over its 16 accumulators and one compare
// decide whether any of those pairs needs the exact re-check (fp32 fma chain on the fp32 midpoint
// and query kept in LDS -- bit-identical to the VALU form and to the oracle).
// DEFER: the pairs that pass the filter are listed and decided by all threads after the group's last matrix instruction
// (as in spring_scan_mfmaw_kernel) instead of on the spot by the lane that found them.  Pays where a workgroup meets many
// of them -- a graph of 100 K vertices, ~70 per workgroup: 59.4 -> 58.2 us per iteration -- and costs three VGPRs, i.e. a wave
// of occupancy: the launcher takes it for graphs whose workgroups all run at once anyway (1 M vertices with it: 173 -> 189 us).
template <int D, int R, bool DEFER = false>
__global__ __launch_bounds__(256) void spring_scan_mfma_kernel(
    const float *__restrict__ pos, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ adj,
    const int32_t *__restrict__ first_edge, const int32_t *__restrict__ own_eids,
    const int32_t *__restrict__ vblock, int64_t row_lo, float L_min, float neg_k, float *__restrict__ Fs,
    float *__restrict__ out_new, double *__restrict__ blockstats, const float *__restrict__ qt,
    const gh_h8 *__restrict__ qA, const int32_t *__restrict__ qexact, int S, uint64_t *__restrict__ cand,
    int32_t *__restrict__ cnt, gh_long_args la, gh_tau_args ta, unsigned long long *__restrict__ stamps) {
    constexpr int LD = 4, NT = 256, TILE = NT * R, NB = 2 * R, HITBUF = 512;
    if ((int)blockIdx.x < ta.nblocks) {  // thresholds of this launch (tau_core.h); diagnostic stamps: the last records of the buffer
        gh_tau_produce<NT, 0>(ta, stamps ? stamps + ((int64_t)gridDim.x - 2 * ta.nblocks + GH_STAMP_EXTRA + blockIdx.x) * 8 : nullptr);
        return;
    }
    const bool coh = ta.nblocks > 0;
    const int nbx = (int)gridDim.x - ta.nblocks;
    const int bx = (int)blockIdx.x - ta.nblocks;
#define GH_STAMP(k) do { if (stamps && threadIdx.x == 0 && blockIdx.y == 0) stamps[(int64_t)bx * 8 + (k)] = wall_clock64(); } while (0)
    GH_STAMP(0);
    if (stamps && threadIdx.x == 0) { stamps[(int64_t)bx * 8 + 6] = vblock[bx + 1] - vblock[bx]; stamps[(int64_t)bx * 8 + 7] = first_edge[vblock[bx + 1]] - first_edge[vblock[bx]]; }
    static_assert(D <= 3, "one 16-deep contraction holds three split coordinates");
    __shared__ float4 tile[TILE];                    // fp32 midpoints of the owned edges (x, y, z, 0)
    __shared__ gh_h8 qa[GH_SCAN_QGROUP * 2];         // A rows: [query][half]
    __shared__ float4 qrec[GH_SCAN_QGROUP];          // (q_0, q_1, q_2, tau) for the exact re-check
    __shared__ uint64_t hkey[HITBUF];
    __shared__ int hq[HITBUF];
    __shared__ uint16_t badlist[TILE];               // references outside the f16 range of the filter
    __shared__ uint32_t ids[TILE];                   // edge ids of the tile's references (the exact path must not wait for memory)
    constexpr int PENDCAP = DEFER ? 512 : 1;
    __shared__ uint32_t pend[PENDCAP];   // (query of the group) << 16 | reference
    __shared__ int hcount, nbad, npend;
    float *mids = reinterpret_cast<float *>(tile);

    const int v0 = vblock[bx], v1 = vblock[bx + 1];
    const int fe0 = first_edge[v0];
    const int nedges = first_edge[v1] - fe0;
    if (threadIdx.x == 0) { hcount = 0; nbad = 0; npend = 0; }
    // The tile's edge ids do not depend on the spring phase: fetched now, under its gathers.  (So were the first query
    // group's operand rows while the thresholds had a launch of their own -- measured neutral at 100 K and 1 M vertices,
    // tools/stamp_probe.py: the scan's 8 us per workgroup there are its ~70 divergent hits, not its staging.)
    auto stage_queries = [&](int s_lo, int nq) {
        const int q = threadIdx.x;  // one query per thread: 32 B of A row, 16 B of exact record
        if (q < nq) {
            const float4 *arow = reinterpret_cast<const float4 *>(qA + 2 * (s_lo + q));
            float4 a0, a1, rec;
            gh_ld_f4x3(arow, arow + 1, reinterpret_cast<const float4 *>(qt) + s_lo + q, coh, a0, a1, rec);
            reinterpret_cast<float4 *>(qa)[2 * q] = a0;
            reinterpret_cast<float4 *>(qa)[2 * q + 1] = a1;
            qrec[q] = rec;
        } else {  // padding row: never passes
            const _Float16 z = (_Float16)0.0f;
            qa[2 * q] = (gh_h8){z, z, z, z, z, z, z, z};
            qa[2 * q + 1] = (gh_h8){z, z, z, z, (_Float16)GH_MF_NEVER, z, z, z};
            qrec[q] = make_float4(0.f, 0.f, 0.f, -1.f);
        }
    };
    for (int j = threadIdx.x; j < nedges; j += NT) ids[j] = own_eids ? (uint32_t)own_eids[fe0 + j] : (uint32_t)(fe0 + j);

    __shared__ double red[(NT / 64) * 2 * LD];
    {
        double sx[LD], sxx[LD];
        gh_phase_a<D, LD, NT, true>(pos, rowptr, adj, first_edge, v0, v1, fe0, nedges, row_lo, L_min, neg_k, Fs, out_new, mids, sx, sxx, la);
        GH_STAMP(1);
        gh_block_stats<LD, NT>(sx, sxx, red, blockstats, bx, nbx, v1 - v0);  // contains the barrier that ends phase A
    }
    GH_STAMP(2);

    // B operands of this wave's NB column blocks; a reference outside the f16 range never passes the
    // MFMA filter and is scanned exactly by its lane (half 0) below
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = lane & 31, hsel = lane >> 5;
    gh_h8 B[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = w * (64 * R) + b * 32 + col;
        const float4 mv4 = tile[j < nedges ? j : 0];
        const float mv[3] = {mv4.x, mv4.y, mv4.z};
        // out of the f16 range: listed, and scanned exactly by the whole workgroup after each query group
        if (!gh_mf_ref_col(mv, j < nedges, hsel, B[b]) && hsel == 0) badlist[atomicAdd(&nbad, 1)] = (uint16_t)j;
    }

    auto park = [&](int s_lo, int s, int j) {  // exact decision on pair (query s of the group, reference j)
        const float4 qr = qrec[s];
        const float4 mv = tile[j];
        const float q[3] = {qr.x, qr.y, qr.z}, m[3] = {mv.x, mv.y, mv.z};
        float d2 = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float df = q[d] - m[d];
            d2 = fmaf(df, df, d2);
        }
        if (d2 <= qr.w) {
            if (ta.cdist) d2 = gh_aten_cdist<D>(q, m);   // parity mode: the key carries the value the reference ranks
            const uint32_t id = ids[j];
            const int p = atomicAdd(&hcount, 1);
            if (p < HITBUF) { hkey[p] = gh_key(d2, id); hq[p] = s_lo + s; }
            else gh_append_candidate(cand, cnt, s_lo + s, gh_key(d2, id));
        }
    };
    if (ta.nblocks > 0) {  // thresholds of this launch: out by now, as a rule
        if (stamps && threadIdx.x == 0) stamps[(int64_t)bx * 8 + 6] = wall_clock64();
        if (threadIdx.x == 0) gh_tau_wait(ta);
        if (stamps && threadIdx.x == 0) stamps[(int64_t)bx * 8 + 7] = wall_clock64();
        __syncthreads();
    }
    const int nex = (int)gh_ld_u32(qexact, coh);
    GH_STAMP(3);

    for (int s_lo = 0; s_lo < S; s_lo += GH_SCAN_QGROUP) {
        const int nq = min(S - s_lo, GH_SCAN_QGROUP);
        if (s_lo > 0) {
            __syncthreads();  // the previous group's rows (and pair list) are still being read
            if (hcount >= HITBUF / 4) {  // many query groups: the parked hits leave before the buffer fills
                gh_flush_hits<HITBUF, NT>(hkey, hq, &hcount, cand, cnt);
                __syncthreads();
                if (threadIdx.x == 0) hcount = 0;
            }
            if constexpr (DEFER) { if (threadIdx.x == 0) npend = 0; }
        }
        stage_queries(s_lo, nq);
        __syncthreads();   // staged rows, edge ids, the list of out-of-range references: visible to every thread
        const int nqb = (nq + 31) / 32;
        const gh_f16x zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int qb = 0; qb < nqb; ++qb) {
            const gh_h8 a = qa[2 * (qb * 32 + col) + hsel];
            // the MFMA of block b+1 is issued before block b's result is tested: its 32 cycles in the
            // matrix pipe run under the ~10 VALU instructions of the test
            gh_f16x f = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B[0], zero, 0, 0, 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                gh_f16x fn = zero;
                if (b + 1 < NB) fn = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B[b + 1], zero, 0, 0, 0);
                // any F <= 0 <=> the smallest of the 16 values read as int32 is <= 0 (sign-magnitude
                // floats order like integers across zero; -0 reads as INT_MIN)
                int mn = min(__float_as_int(f[0]), __float_as_int(f[1]));
#pragma unroll
                for (int i = 2; i < 16; ++i) mn = min(mn, __float_as_int(f[i]));
                asm volatile("" : "+v"(mn));  // opaque: else the compiler drops the tree for 16 tests + branches
                if (mn <= 0) {  // rare: result row (i&3) + 8(i>>2) + 4*half, column = this lane's reference
                    // sign bits of the 16 values, f[0] in bit 15 ... f[15] in bit 0 (a candidate has F < 0:
                    // the slack in C0 and T is twice the error bound), then one exact re-check per set bit
                    uint32_t m = 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) m = __builtin_amdgcn_alignbit(m, __float_as_uint(f[i]), 31);
                    const int j = w * (64 * R) + b * 32 + col;
                    if (j < nedges) {
                        while (m) {
                            const int bit = 31 - __builtin_clz(m);
                            m &= ~(1u << bit);
                            const int i = 15 - bit;
                            const int s = qb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hsel;
                            if (s < nq) {
                                if constexpr (DEFER) {
                                    const int p = atomicAdd(&npend, 1);
                                    if (p < PENDCAP) pend[p] = ((uint32_t)s << 16) | (uint32_t)j;
                                    else park(s_lo, s, j);   // list full: decided on the spot
                                } else {
                                    park(s_lo, s, j);
                                }
                            }
                        }
                    }
                }
                f = fn;
            }
        }
        if constexpr (DEFER) {
            __syncthreads();
            for (int p = threadIdx.x, np = min(npend, PENDCAP); p < np; p += NT) park(s_lo, (int)(pend[p] >> 16), (int)(pend[p] & 0xFFFFu));
        }
        // outside the f16 range: exact scan of this group's listed queries over the whole tile ...
        for (int x = 0; x < nex; ++x) {
            const int s = (int)gh_ld_u32(qexact + 1 + x, coh) - s_lo;
            if (s < 0 || s >= nq) continue;
            for (int j = threadIdx.x; j < nedges; j += NT) park(s_lo, s, j);
        }
        // ... and of the tile's out-of-range references against every query of the group: all threads, one
        // pair each (a layout in which a hub has flown off takes thousands of midpoints out of range)
        const int nb_ = nbad;  // complete: the staging barrier of this group came after the B operands were built
        for (int p = threadIdx.x; p < nb_ * nq; p += NT) {
            const int s = p % nq;
            const gh_h8 hi = qa[2 * s + 1];
            if ((float)hi[4] == GH_MF_NEVER) continue;  // a listed query: the loop above has done this pair
            park(s_lo, s, badlist[p / nq]);
        }
    }
    __syncthreads();
    GH_STAMP(4);
    gh_flush_hits<HITBUF, NT>(hkey, hq, &hcount, cand, cnt);
    GH_STAMP(5);
#undef GH_STAMP
}

// Wide rows (5 <= D <= 16) with the pre-filter on the matrix pipe (scan_core.h "Wide rows"): phase A as in
// spring_scan_kernel, phase B as in spring_scan_mfma_kernel with 16 * KB deep contractions -- each wave owns 4 column
// blocks of 32 references whose f16 operands stay in registers, the queries stream past as A operands from LDS, a lane
// owns one reference column and 16 query rows of every 32x32 result; the fp32 tile and query records stay in LDS for
// the exact re-check of the few pairs that pass.  Tile = 512 owned edges, 256 threads.
// Packed fp32 VALU needs D/2 instructions per pair and lane; this form one (two) matrix instructions per 1024 pairs plus
// the same ~10 VALU instructions to test them.
template <int D, int LD, bool LONG>
__global__ __launch_bounds__(256) void spring_scan_mfmaw_kernel(
    const float *__restrict__ pos, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ adj,
    const int32_t *__restrict__ first_edge, const int32_t *__restrict__ own_eids,
    const int32_t *__restrict__ vblock, int64_t row_lo, float L_min, float neg_k, float *__restrict__ Fs,
    float *__restrict__ out_new, double *__restrict__ blockstats, const float *__restrict__ qt,
    const float *__restrict__ qscan, int S,
    uint64_t *__restrict__ cand, int32_t *__restrict__ cnt, gh_long_args la, gh_tau_args ta,
    unsigned long long *__restrict__ stamps) {
    constexpr int NT = 256, R = 2, TILE = NT * R, NB = 2 * R;
    constexpr int HITBUF = LD == 16 ? 256 : 512;   // (16-float rows: two workgroups per CU must fit 160 KB of LDS)
    constexpr int KB = D <= 10 ? 1 : 2;
    constexpr int QS = LD + 4, QT = LD;
    if ((int)blockIdx.x < ta.nblocks) {
        if (blockIdx.y == 0) gh_tau_produce<NT, -1>(ta, stamps ? stamps + ((int64_t)gridDim.x - 2 * ta.nblocks + GH_STAMP_EXTRA + blockIdx.x) * 8 : nullptr);
        return;
    }
    const bool coh = ta.nblocks > 0;
    const int nbx = (int)gridDim.x - ta.nblocks;
    const int bx = (int)blockIdx.x - ta.nblocks;
#define GH_STAMP(k) do { if (stamps && threadIdx.x == 0 && blockIdx.y == 0) stamps[(int64_t)bx * 8 + (k)] = wall_clock64(); } while (0)
    GH_STAMP(0);
    __shared__ float4 tile[TILE * LD / 4];                 // fp32 midpoints of the owned edges
    __shared__ gh_h8 qa[GH_SCAN_QGROUP * KB * 2];          // A rows: [query][block of 16][half]
    __shared__ float4 qsh[(GH_SCAN_QGROUP + 1) * (QS / 4)]; // (-2q, t) records: the exact re-check takes q from them
    __shared__ float taush[GH_SCAN_QGROUP];
    __shared__ uint64_t hkey[HITBUF];
    __shared__ int hq[HITBUF];
    __shared__ uint16_t badlist[TILE];
    __shared__ uint32_t ids[TILE];
    __shared__ uint16_t exq[GH_SCAN_QGROUP];               // queries of the group outside the f16 range
    // pairs that passed the filter, (query of the group) << 16 | reference: listed by the lane that found them and decided
    // exactly by ALL threads after the group's last matrix instruction -- inline, each pair is a divergent detour of a
    // D-step chain and two LDS row reads for one lane of a wave, and this form lets ~3x the necessary pairs through
    // (D = 6 at 1 M vertices: 277 -> 266 us per iteration; the split form of D <= 3 keeps the inline check: there the list
    // cost it three VGPRs -- a wave of occupancy -- for a dozen pairs per workgroup: 173 -> 189 us)
    constexpr int PENDCAP = 512;
    __shared__ uint32_t pend[PENDCAP];
    __shared__ int hcount, nbad, nexq, npend;
    float *mids = reinterpret_cast<float *>(tile);

    const int v0 = vblock[bx], v1 = vblock[bx + 1];
    const int fe0 = first_edge[v0];
    const int nedges = first_edge[v1] - fe0;
    if (threadIdx.x == 0) { hcount = 0; nbad = 0; nexq = 0; npend = 0; }
    for (int j = threadIdx.x; j < nedges; j += NT) ids[j] = own_eids ? (uint32_t)own_eids[fe0 + j] : (uint32_t)(fe0 + j);

    // ---- phase A (every query slice redoes it for its tile, slice 0 alone stores its results)
    // (the reduction scratch lives in the hit buffer, idle until the scan: two workgroups of 16-float rows fit the CU's
    // 160 KB of LDS with 3 KB to spare instead of 0.8)
    static_assert(sizeof(double) * (NT / 64) * 2 * LD <= sizeof(uint64_t) * HITBUF, "reduction scratch must fit the hit buffer");
    double *red = reinterpret_cast<double *>(hkey);
    {
        double sx[LD], sxx[LD];
        const bool store = blockIdx.y == 0;
        gh_phase_a<D, LD, NT, LONG>(pos, rowptr, adj, first_edge, v0, v1, fe0, nedges, row_lo, L_min, neg_k, store ? Fs : nullptr,
                                    store ? out_new : nullptr, mids, sx, sxx, la, Fs);
        GH_STAMP(1);
        gh_block_stats<LD, NT>(sx, sxx, red, store ? blockstats : nullptr, bx, nbx, v1 - v0);  // contains the barrier that ends phase A
    }
    GH_STAMP(2);

    // ---- phase B: operands of this wave's column blocks
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = lane & 31, hsel = lane >> 5;
    gh_h8 B[NB][KB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = w * (64 * R) + b * 32 + col;
        float mv[LD];
        gh_load_row<LD>(mids, j < nedges ? j : 0, mv);
        if (!gh_mfw_ref_col<D, KB>(mv, j < nedges, hsel, B[b]) && hsel == 0) badlist[atomicAdd(&nbad, 1)] = (uint16_t)j;
    }

    auto park = [&](int s_lo, int s, int j) {  // exact decision on pair (query s of the group, reference j)
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
            if (ta.cdist) {   // parity mode: the key carries the value the reference ranks
                float q[D];
#pragma unroll
                for (int d = 0; d < D; ++d) q[d] = -0.5f * qv[d];
                d2 = gh_aten_cdist<D>(q, mv);
            }
            const uint32_t id = ids[j];
            const int p = atomicAdd(&hcount, 1);
            if (p < HITBUF) { hkey[p] = gh_key(d2, id); hq[p] = s_lo + s; }
            else gh_append_candidate(cand, cnt, s_lo + s, gh_key(d2, id));
        }
    };
    if (stamps && ta.nblocks > 0 && threadIdx.x == 0 && blockIdx.y == 0) stamps[(int64_t)bx * 8 + 6] = wall_clock64();
    if (ta.nblocks > 0 && threadIdx.x == 0) gh_tau_wait(ta);  // thresholds of this launch: out by now, as a rule
    if (stamps && ta.nblocks > 0 && threadIdx.x == 0 && blockIdx.y == 0) stamps[(int64_t)bx * 8 + 7] = wall_clock64();
    __syncthreads();
    GH_STAMP(3);

    const int per = (S + (int)gridDim.y - 1) / (int)gridDim.y;
    const int s_begin = (int)blockIdx.y * per, s_end = min(S, s_begin + per);
    const gh_f16x zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s_lo = s_begin; s_lo < s_end; s_lo += GH_SCAN_QGROUP) {
        const int nq = min(s_end - s_lo, GH_SCAN_QGROUP);
        if (s_lo > s_begin) {
            __syncthreads();  // the previous group's rows and list are still being read
            const bool flush = hcount >= HITBUF / 4;   // many query groups: the parked hits leave before the buffer fills
            if (flush) gh_flush_hits<HITBUF, NT>(hkey, hq, &hcount, cand, cnt);
            __syncthreads();
            if (threadIdx.x == 0) { nexq = 0; npend = 0; if (flush) hcount = 0; }
            __syncthreads();
        }
        gh_stage_queries<QS, QT, NT>(qscan, qt, s_lo, nq, qsh, taush, coh);
        {
            // one query per thread: its A row is BUILT here from the query record and the threshold (80 instructions per
            // group and thread) -- the threshold kernel then carries no operand-row code for this form, which as a launch of
            // its own cost it 15 - 25 us (tau_core.h), and a query outside the f16 range goes on a list of this workgroup
            const int q = threadIdx.x;
            _Float16 row[16 * KB];
            bool ok = true;
            if (q < nq) {
                const float4 *src = reinterpret_cast<const float4 *>(qscan) + (int64_t)(s_lo + q) * (QS / 4);
                float qv[16];
#pragma unroll
                for (int d = 0; d < 16; ++d) qv[d] = 0.0f;
#pragma unroll
                for (int i = 0; i < LD / 4; ++i) {
                    const float4 v = gh_ld_f4(src + i, coh);   // (-2q_0 ..): the record holds -2q, exact both ways
                    qv[4 * i] = -0.5f * v.x; qv[4 * i + 1] = -0.5f * v.y; qv[4 * i + 2] = -0.5f * v.z; qv[4 * i + 3] = -0.5f * v.w;
                }
                const float tau = gh_ld_f32(qt + (int64_t)(s_lo + q) * QS + QT, coh);
                ok = gh_mfw_query_row<KB>(qv, tau, row);
                if (!ok) exq[atomicAdd(&nexq, 1)] = (uint16_t)q;
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
        __syncthreads();   // staged rows, edge ids, the list of out-of-range references: visible to every thread
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
                if (mn <= 0) {  // rare: result row (i&3) + 8(i>>2) + 4*half, column = this lane's reference
                    uint32_t m = 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) m = __builtin_amdgcn_alignbit(m, __float_as_uint(f[i]), 31);
                    const int j = w * (64 * R) + b * 32 + col;
                    if (j < nedges) {
                        while (m) {
                            const int bit = 31 - __builtin_clz(m);
                            m &= ~(1u << bit);
                            const int i = 15 - bit;
                            const int s = qb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hsel;
                            if (s < nq) {
                                const int p = atomicAdd(&npend, 1);
                                if (p < PENDCAP) pend[p] = ((uint32_t)s << 16) | (uint32_t)j;
                                else park(s_lo, s, j);   // list full: decided on the spot
                            }
                        }
                    }
                }
                f = fn;
            }
        }
        __syncthreads();
        for (int p = threadIdx.x, np = min(npend, PENDCAP); p < np; p += NT) park(s_lo, (int)(pend[p] >> 16), (int)(pend[p] & 0xFFFFu));
        // outside the f16 range: exact scan of this group's listed queries over the whole tile ...
        const int nex = nexq;   // complete: the staging barrier of this group came after the rows were built
        for (int x = 0; x < nex; ++x) {
            const int s = exq[x];
            for (int j = threadIdx.x; j < nedges; j += NT) park(s_lo, s, j);
        }
        // ... and of the tile's out-of-range references against every query of the group
        const int nb_ = nbad;
        for (int p = threadIdx.x; p < nb_ * nq; p += NT) {
            const int s = p % nq;
            const gh_h8 tv = qa[(s * KB + (gh_mfw<KB>::base + 3) / 16) * 2 + ((gh_mfw<KB>::base + 3) % 16) / 8];
            if ((float)tv[(gh_mfw<KB>::base + 3) % 8] == GH_MF_NEVER) continue;  // a listed query: the loop above has done this pair
            park(s_lo, s, badlist[p / nq]);
        }
    }
    __syncthreads();
    GH_STAMP(4);
    gh_flush_hits<HITBUF, NT>(hkey, hq, &hcount, cand, cnt);
    GH_STAMP(5);
#undef GH_STAMP
}

// The thresholds of this iteration: computed by the first workgroups of the fused launch itself (h->tau_embedded), or
// already in place (knn_tau_kernel ran; nblocks = 0).
static unsigned fused_grid(const gh_engine *h, const gh_tau_args &ta) {
    return (unsigned)(h->n_vblocks + ta.nblocks);
}
gh_tau_args fused_tau_args(gh_engine *h, int nt) {
    gh_tau_args ta{};
    ta.cdist = h->cdist ? 1 : 0;
    if (!h->tau_embedded) return ta;
    ta = gh_make_tau_args(h);
    ta.cdist = h->cdist ? 1 : 0;
    ta.flag = h->d_tau_flag;          // zeroed by this iteration's set-up (setup_core.h), S once the producers are through
    ta.target = (unsigned)h->S;
    ta.nblocks = gh_tau_blocks((int)h->S, nt);
    ta.wait_failed = h->d_wait_failed;
    return ta;
}

template <int D, int R, bool DEFER>
void launch_mfma_d(gh_engine *h) {
    const gh_tau_args ta = fused_tau_args(h, 256);
    spring_scan_mfma_kernel<D, R, DEFER><<<dim3(fused_grid(h, ta)), dim3(256), 0, h->stream>>>(
        h->d_pos, h->d_rowptr, h->d_adj, h->d_first_edge, h->d_own_eids, h->d_vblock, h->part.row_lo, h->prm.L_min,
        -h->prm.k_attr, h->d_Fs, h->d_new, h->d_blockstats, h->d_q, reinterpret_cast<const gh_h8 *>(h->d_qA),
        h->d_qexact, (int)h->S, h->d_cand, h->d_cnt, gh_make_long_args(h, true), ta, h->d_stamps);
}
template <int D, int R>
void launch_mfma(gh_engine *h) {
    if (R == 2 && h->n_vblocks <= 2048) launch_mfma_d<D, R, (R == 2)>(h);   // one round of workgroups: occupancy does not matter
    else launch_mfma_d<D, R, false>(h);
}

template <int D, int LD, bool LONG>
void launch_mfmaw_l(gh_engine *h) {
    // few workgroups (a small graph with wide rows): the queries in slices over blockIdx.y, as many as still run all at once
    unsigned ny = 1;
    if (h->n_vblocks < 384) {
        static int resident = 0;  // per instantiation
        if (resident == 0) {
            int occ = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spring_scan_mfmaw_kernel<D, LD, LONG>, 256, 0) != hipSuccess) occ = 1;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) cus = 256;
            resident = (occ > 0 ? occ : 1) * (cus > 0 ? cus : 256);
        }
        ny = (unsigned)(resident / (h->n_vblocks > 0 ? h->n_vblocks : 1));
        if (ny > 4) ny = 4;
        if (ny < 1) ny = 1;
        if ((int64_t)ny * 64 > h->S) ny = 1;
    }
    const gh_tau_args ta = fused_tau_args(h, 256);
    spring_scan_mfmaw_kernel<D, LD, LONG><<<dim3(fused_grid(h, ta), ny), dim3(256), 0, h->stream>>>(
        h->d_pos, h->d_rowptr, h->d_adj, h->d_first_edge, h->d_own_eids, h->d_vblock, h->part.row_lo, h->prm.L_min, -h->prm.k_attr,
        h->d_Fs, h->d_new, h->d_blockstats, h->d_q, h->d_qscan, (int)h->S,
        h->d_cand, h->d_cnt, gh_make_long_args(h, true), ta, h->d_stamps);
}
template <int D, int LD>
void launch_mfmaw(gh_engine *h) {
    if (gh_make_long_args(h).n > 0) launch_mfmaw_l<D, LD, true>(h);
    else launch_mfmaw_l<D, LD, false>(h);
}

}  // namespace

gh_status gh_launch_spring_scan(gh_engine *h) {
    if (h->n_vblocks == 0) return GH_OK;
    GH_TRY_ST(gh_launch_spring_long(h, h->d_Fs, 0));  // hubs first: their rows' forces are read back in phase A
    gh_scope t(h, "spring_scan");
    if (fused_mfma(h->LD, h->D, h->S) && h->D > 3) {
        switch (h->D) {
            case 4: launch_mfmaw<4, 4>(h); break;
            case 5: launch_mfmaw<5, 8>(h); break;
            case 6: launch_mfmaw<6, 8>(h); break;
            case 7: launch_mfmaw<7, 8>(h); break;
            case 8: launch_mfmaw<8, 8>(h); break;
            case 9: launch_mfmaw<9, 16>(h); break;
            case 10: launch_mfmaw<10, 16>(h); break;
            case 11: launch_mfmaw<11, 16>(h); break;
            case 12: launch_mfmaw<12, 16>(h); break;
            case 13: launch_mfmaw<13, 16>(h); break;
            case 14: launch_mfmaw<14, 16>(h); break;
            case 15: launch_mfmaw<15, 16>(h); break;
            default: launch_mfmaw<16, 16>(h); break;
        }
    } else if (fused_mfma(h->LD, h->D, h->S)) {
        if (h->D == 2) launch_mfma<2, 2>(h); else launch_mfma<3, 2>(h);
    } else {
        h->err = "fused spring+scan launched for an unsupported dimension";
        return GH_ERR_RUNTIME;
    }
    GH_LAUNCH_CHECK();
    h->new0_ready = true;  // d_new = pos + Fs and d_blockstats[n_vblocks] are in place
    return GH_OK;
}
