// Fused spring + KNN scan: the dominant kernel of an iteration.
//
// Why fused: the spring pull is bound by the chip's random-line fetch rate (8 neighbour rows
// of 16 bytes per vertex, each a 128-byte line from the Infinity Cache: ~64 G lines/s), the
// KNN scan by the fp32 VALU.  As separate kernels they run back to back (measured 166 + 153 us
// on the 1M-vertex graph) and the midpoints make a round trip through HBM.  Here one workgroup
// owns a range of vertices holding at most TILE owned edges (edge (u, v), u < v, is owned by
// u; the edge list is sorted by first endpoint so the range is contiguous):
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

#include <stdio.h>
#include <stdlib.h>

// Fused-kernel geometry: NT threads x R references per thread = owned edges per workgroup.
// Measured on the 1M-vertex graph (D = 3): 256x8 225 us, 256x4 197 us -- the smaller tile costs
// ~18 % more VALU work per pair but halves the LDS per workgroup, so twice as many workgroups
// are resident and their gather / VALU phases interleave better.
// GRAPHEM_HIP_FUSED_CFG="NT,R" overrides the LD = 4 default for experiments.
// Small graphs get smaller tiles so that there are still >= ~1000 workgroups for 256 CUs.
static void fused_cfg(int LD, int64_t own_edges, int *nt, int *r) {
    *nt = 256;
    *r = LD <= 4 ? 4 : LD <= 8 ? 4 : 2;
    if (LD <= 4 && own_edges < 1500000) *r = 2;
    if (LD <= 4 && own_edges < 400000) *nt = 128;
    const char *e = getenv("GRAPHEM_HIP_FUSED_CFG");
    if (e && LD <= 4) {
        int a = 0, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) == 2 && (a == 128 || a == 256) && (b == 2 || b == 4 || b == 8)) { *nt = a; *r = b; }
    }
}
int gh_fused_tile(int LD, int64_t own_edges) {
    int nt, r;
    fused_cfg(LD, own_edges, &nt, &r);
    return nt * r;
}

namespace {

// Phase A of both fused kernels for one workgroup: spring forces of the vertices v0..v1 -> Fs,
// midpoints of their owned edges -> LDS, and new0 = pos + Fs -> out_new (pt.py:796-799 for every
// vertex the intersection phase does not touch: Fs + 0 == Fs exactly; the few touched vertices are
// redone by stats_fix_kernel).  Returns this thread's column sums of new0 for the fp64 statistics.
template <int D, int LD, int NT>
__device__ __forceinline__ void gh_phase_a(const float *__restrict__ pos, const int32_t *__restrict__ rowptr,
                                           const int32_t *__restrict__ adj, const int32_t *__restrict__ first_edge,
                                           int v0, int v1, int fe0, int64_t row_lo, float L_min, float neg_k,
                                           float *__restrict__ Fs, float *__restrict__ out_new, float *mids,
                                           double (&sx)[LD], double (&sxx)[LD]) {
#pragma unroll
    for (int d = 0; d < LD; ++d) { sx[d] = 0.0; sxx[d] = 0.0; }
    for (int i = v0 + threadIdx.x; i < v1; i += NT) {
        const int64_t x = row_lo + i;
        float px[LD], F[LD], nw[LD];
        gh_load_row<LD>(pos, x, px);
        spring_pull<D, LD, true>(pos, adj, rowptr[i], rowptr[i + 1], x, px, L_min, neg_k, F, mids, first_edge[i] - fe0);
        gh_store_row<LD>(Fs, i, F);
#pragma unroll
        for (int d = 0; d < LD; ++d) {
            nw[d] = px[d] + F[d];
            sx[d] += (double)nw[d];
            sxx[d] += (double)nw[d] * (double)nw[d];
        }
        gh_store_row<LD>(out_new, i, nw);
    }
}

// Workgroup reduction of the per-thread column sums -> blockstats[blockIdx.x][2*LD] (fixed order).
template <int LD, int NT>
__device__ __forceinline__ void gh_block_stats(const double (&sx)[LD], const double (&sxx)[LD], double *red /* [NT/64][2*LD] */,
                                               double *__restrict__ blockstats) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < LD; ++d) {
        const double a = gh_wave_sum(sx[d]), b = gh_wave_sum(sxx[d]);
        if (lane == 0) { red[w * 2 * LD + d] = a; red[w * 2 * LD + LD + d] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * LD) {
        double v = red[threadIdx.x];
#pragma unroll
        for (int ww = 1; ww < NT / 64; ++ww) v += red[ww * 2 * LD + threadIdx.x];
        blockstats[(int64_t)blockIdx.x * 2 * LD + threadIdx.x] = v;
    }
}


template <int D, int LD, int R, int NT>
__global__ __launch_bounds__(NT) void spring_scan_kernel(
    const float *__restrict__ pos, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ adj,
    const int32_t *__restrict__ first_edge, const int32_t *__restrict__ own_eids,
    const int32_t *__restrict__ vblock, int64_t row_lo, float L_min, float neg_k, float *__restrict__ Fs,
    float *__restrict__ out_new, double *__restrict__ blockstats, const float *__restrict__ qt,
    const float *__restrict__ qscan, int S, uint64_t *__restrict__ cand, int32_t *__restrict__ cnt) {
    constexpr int TILE = NT * R;
    constexpr int QS = D <= 3 ? 4 : LD + 4;
    constexpr int HITBUF = TILE * LD * 4 / 16;  // hit records reuse the midpoint tile's LDS
    __shared__ float4 tile[TILE * LD / 4];
    __shared__ float4 qsh[(GH_SCAN_QGROUP + 1) * (QS / 4)];
    __shared__ float taush[GH_SCAN_QGROUP];
    __shared__ int hcount;
    float *mids = reinterpret_cast<float *>(tile);

    const int v0 = vblock[blockIdx.x], v1 = vblock[blockIdx.x + 1];
    const int fe0 = first_edge[v0];
    const int nedges = first_edge[v1] - fe0;
    if (threadIdx.x == 0) hcount = 0;

    // ---- phase A: spring forces, new0 = pos + Fs, midpoints of the owned edges to LDS
    __shared__ double red[(NT / 64) * 2 * LD];
    {
        double sx[LD], sxx[LD];
        gh_phase_a<D, LD, NT>(pos, rowptr, adj, first_edge, v0, v1, fe0, row_lo, L_min, neg_k, Fs, out_new, mids, sx, sxx);
        gh_block_stats<LD, NT>(sx, sxx, red, blockstats);  // contains the barrier that ends phase A
    }

    // ---- phase B: the tile becomes this workgroup's references
    gh_f2 m[R / 2][D], c0[R / 2];
    uint32_t id[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int j = r * NT + threadIdx.x;
        float mv[LD];
        if (j < nedges) {
            gh_load_row<LD>(mids, j, mv);
            id[r] = own_eids ? (uint32_t)own_eids[fe0 + j] : (uint32_t)(fe0 + j);
        } else {
#pragma unroll
            for (int d = 0; d < LD; ++d) mv[d] = 0.0f;  // padding slot: c0 = +inf never passes the filter
            id[r] = 0xFFFFFFFFu;
        }
        const float c = gh_ref_c0<D>(mv, j < nedges);
        if (r & 1) c0[r / 2].y = c; else c0[r / 2].x = c;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (r & 1) m[r / 2][d].y = mv[d];
            else m[r / 2][d].x = mv[d];
        }
    }
    __syncthreads();  // every thread has its references: the tile's LDS becomes the hit buffer
    uint64_t *hkey = reinterpret_cast<uint64_t *>(tile);
    int *hq = reinterpret_cast<int *>(hkey + HITBUF);
    for (int s_lo = 0; s_lo < S; s_lo += GH_SCAN_QGROUP) {
        const int nq = min(S - s_lo, GH_SCAN_QGROUP);
        if (s_lo > 0) __syncthreads();  // the previous group's records are still being read
        gh_stage_queries<QS, (D <= 3 ? 3 : LD), NT>(qscan, qt, s_lo, nq, qsh, taush);
        __syncthreads();
        gh_scan_queries<D, R, HITBUF>(m, c0, id, qsh, nq, s_lo, taush, hkey, hq, &hcount, cand, cnt);
    }
    __syncthreads();
    gh_flush_hits<HITBUF, NT>(hkey, hq, &hcount, cand, cnt);
}

// MFMA form of phase B (D <= 3, 256 threads, 1024-edge tiles): each wave owns 256 references of
// the tile as 16 groups of 16, one float per lane and group (scan_core.h gh_scan_queries_mfma).
template <int D>
__global__ __launch_bounds__(256) void spring_scan_mfma_kernel(
    const float *__restrict__ pos, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ adj,
    const int32_t *__restrict__ first_edge, const int32_t *__restrict__ own_eids,
    const int32_t *__restrict__ vblock, int64_t row_lo, float L_min, float neg_k, float *__restrict__ Fs,
    float *__restrict__ out_new, double *__restrict__ blockstats, const float *__restrict__ qt,
    const float *__restrict__ qscan, int S, uint64_t *__restrict__ cand, int32_t *__restrict__ cnt) {
    constexpr int LD = 4, TILE = 1024, G = 16;
    constexpr int HITBUF = TILE * LD * 4 / 16;
    __shared__ float4 tile[TILE * LD / 4];
    __shared__ __align__(16) float qT[3 * GH_SCAN_QGROUP];
    __shared__ __align__(16) float tneg[GH_SCAN_QGROUP];
    __shared__ int hcount;
    float *mids = reinterpret_cast<float *>(tile);

    const int v0 = vblock[blockIdx.x], v1 = vblock[blockIdx.x + 1];
    const int fe0 = first_edge[v0];
    const int nedges = first_edge[v1] - fe0;
    if (threadIdx.x == 0) hcount = 0;

    __shared__ double red[4 * 2 * LD];
    {
        double sx[LD], sxx[LD];
        gh_phase_a<D, LD, 256>(pos, rowptr, adj, first_edge, v0, v1, fe0, row_lo, L_min, neg_k, Fs, out_new, mids, sx, sxx);
        gh_block_stats<LD, 256>(sx, sxx, red, blockstats);  // contains the barrier that ends phase A
    }

    // B operands: lane (kq = lane>>4, col = lane&15) holds component kq of reference w*256 + g*16 + col
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    float bq[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int j = w * 256 + g * 16 + col;
        float v;
        if (j < nedges) {
            if (kq < 3) {
                v = mids[j * LD + kq];
            } else {
                float mv[LD];
                gh_load_row<LD>(mids, j, mv);
                v = gh_ref_c0<D>(mv, true);
            }
        } else {
            v = kq < 3 ? 0.0f : INFINITY;  // padding slot: F = +inf never passes
        }
        bq[g] = v;
    }
    __syncthreads();  // the tile's LDS becomes the hit buffer
    uint64_t *hkey = reinterpret_cast<uint64_t *>(tile);
    int *hq = reinterpret_cast<int *>(hkey + HITBUF);
    const int nvalid = min(max(nedges - w * 256, 0), 256);
    for (int s_lo = 0; s_lo < S; s_lo += GH_SCAN_QGROUP) {
        const int nq = min(S - s_lo, GH_SCAN_QGROUP);
        if (s_lo > 0) __syncthreads();
        {   // transposed staging: qT[k][q] = -2 q_k, tneg[q] = -t (padding queries: -t = +inf)
            const int q = threadIdx.x;
            const float4 rec = q < nq ? reinterpret_cast<const float4 *>(qscan)[s_lo + q]
                                      : make_float4(0.f, 0.f, 0.f, -INFINITY);
            qT[q] = rec.x;
            qT[GH_SCAN_QGROUP + q] = rec.y;
            qT[2 * GH_SCAN_QGROUP + q] = rec.z;
            tneg[q] = -rec.w;
        }
        __syncthreads();
        gh_scan_queries_mfma<D, G, HITBUF>(bq, (uint32_t)(fe0 + w * 256), nvalid, qT, tneg, nq, s_lo, qt, hkey, hq,
                                           &hcount, cand, cnt);
    }
    __syncthreads();
    gh_flush_hits<HITBUF, 256>(hkey, hq, &hcount, cand, cnt);
}

template <int D>
void launch_mfma(gh_engine *h) {
    spring_scan_mfma_kernel<D><<<dim3((unsigned)h->n_vblocks), dim3(256), 0, h->stream>>>(
        h->d_pos, h->d_rowptr, h->d_adj, h->d_first_edge, h->d_own_eids, h->d_vblock, h->part.row_lo, h->prm.L_min, -h->prm.k_attr,
        h->d_Fs, h->d_new, h->d_blockstats, h->d_q, h->d_qscan, (int)h->S, h->d_cand, h->d_cnt);
}

template <int D, int LD, int R, int NT>
void launch(gh_engine *h) {
    spring_scan_kernel<D, LD, R, NT><<<dim3((unsigned)h->n_vblocks), dim3(NT), 0, h->stream>>>(
        h->d_pos, h->d_rowptr, h->d_adj, h->d_first_edge, h->d_own_eids, h->d_vblock, h->part.row_lo, h->prm.L_min, -h->prm.k_attr,
        h->d_Fs, h->d_new, h->d_blockstats, h->d_q, h->d_qscan, (int)h->S, h->d_cand, h->d_cnt);
}

}  // namespace

gh_status gh_launch_spring_scan(gh_engine *h) {
    if (h->n_vblocks == 0) return GH_OK;
    gh_scope t(h, "spring_scan");
    int nt, r;
    fused_cfg(h->LD, h->own_count, &nt, &r);
#define GH_FUSED_D(NTT, RR)                                   \
    switch (h->D) {                                           \
        case 2: launch<2, 4, RR, NTT>(h); break;              \
        case 3: launch<3, 4, RR, NTT>(h); break;              \
        default: launch<4, 4, RR, NTT>(h); break;             \
    }
    // The MFMA form of the pre-filter is opt-in: measured 214-225 us against 164 us for the packed
    // VALU form on the 1M-vertex graph (AGPR read-back, MFMA->VALU latency, 100 VGPRs).
    if (h->LD == 4 && h->D <= 3 && nt == 256 && r == 4 && !h->d_own_eids && getenv("GRAPHEM_HIP_MFMA")) {
        if (h->D == 2) launch_mfma<2>(h); else launch_mfma<3>(h);
    } else if (h->LD == 4) {
        if (nt == 256 && r == 8) { GH_FUSED_D(256, 8) }
        else if (nt == 256 && r == 4) { GH_FUSED_D(256, 4) }
        else if (nt == 256 && r == 2) { GH_FUSED_D(256, 2) }
        else if (nt == 128 && r == 8) { GH_FUSED_D(128, 8) }
        else if (nt == 128 && r == 4) { GH_FUSED_D(128, 4) }
        else { GH_FUSED_D(128, 2) }
    } else if (h->LD == 8) {
        launch<8, 8, 4, 256>(h);
    } else if (h->LD == 16) {
        launch<16, 16, 2, 256>(h);
    } else {
        h->err = "fused spring+scan launched for an unsupported dimension";
        return GH_ERR_RUNTIME;
    }
#undef GH_FUSED_D
    GH_LAUNCH_CHECK();
    h->new0_ready = true;  // d_new = pos + Fs and d_blockstats[n_vblocks] are in place
    return GH_OK;
}
