// Spring attraction, intersection repulsion, integration and normalisation: the work of
// _compute_spring_forces (reference pt.py:595-636), _compute_intersection_forces +
// _check_line_intersections (pt.py:638-774) and the tail of update_positions
// (pt.py:796-804), plus the sampler that stands in for torch.randperm (pt.py:409).
#include "common.h"
#include "engine.h"
#include "intersect_core.h"
#include "setup_core.h"

namespace {

// ---------------------------------------------------------------------------------
// Spring force on one vertex, pull style: the vertex walks its own neighbour list and
// nobody else writes its row, so there are no atomics and the sum is reproducible.
// The list is stored in the reference's summation order (all edges where the vertex
// is the first endpoint, then all where it is the second, each in edge order:
// the two sequential index_add_ calls of pt.py:633-634).  For a neighbour y of x
//   diff = p_y - p_x,  dist = |diff| + 1e-6,  f = (-k_attr * (dist - L_min)) * (diff / dist)
// which equals +f of pt.py:629 when x is the first endpoint and -f when it is the
// second, bit for bit (negation commutes with every rounding involved).
// Spring forces of the rows [row_lo, row_lo + rows) (pt.py:595-636) and, when WRITE_MID, the
// midpoints of the edges those rows own (edges are sorted by first endpoint, so the edges of
// row i are first_edge[i] .. first_edge[i+1]).  F goes to outF[(i + f_row0) * LD].
template <int D, int LD, bool WRITE_MID, bool LONG>
__global__ __launch_bounds__(256) void spring_kernel(
    const float *__restrict__ pos, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ adj,
    const int32_t *__restrict__ first_edge, int64_t edge_lo, int64_t row_lo, int64_t rows, float L_min,
    float neg_k, float *__restrict__ outF, int64_t f_row0, float *__restrict__ mid, gh_long_args la) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const int64_t x = row_lo + i;
    float px[LD], F[LD];
    gh_load_row<LD>(pos, x, px);
    int64_t mid_row0 = 0;
    if (WRITE_MID) mid_row0 = first_edge[i] - edge_lo;  // d_mid row of the first edge this row owns
    if constexpr (LONG)
        spring_row<D, LD, WRITE_MID>(pos, adj, rowptr[i], rowptr[i + 1], x, px, L_min, neg_k, F, mid, mid_row0, la, (int)i,
                                     outF + (i + f_row0) * LD);
    else
        spring_pull<D, LD, WRITE_MID>(pos, adj, rowptr[i], rowptr[i + 1], x, px, L_min, neg_k, F, mid, mid_row0);
    gh_store_row<LD>(outF, i + f_row0, F);
}

// Spring force of the long rows (common.h GH_LONG_DEG) in two launches.
// (1) long_terms_kernel: one thread per pull-list entry of a long row computes that neighbour's force
//     term with the arithmetic of spring_pull -- all gathers of all hubs in flight at once -- and
//     stores it component-major per row: terms[eptr[r]*D + d*deg_r + idx].
// (2) long_sum_kernel: one wave per long row ADDS THE TERMS IN LIST ORDER, the reference's order
//     (pt.py:633-634): 64 terms per step, lane 0 starts from the running sum, then 63 dependent adds
//     x_l = x_(l-1) + term_l with the neighbour lane read through DPP wave_shr:1, so that lane k
//     holds ((F + t_0) + t_1) + ... + t_k.  The reads are contiguous and fetched two steps ahead.
template <int D, int LD>
__global__ __launch_bounds__(256) void long_terms_kernel(const float *__restrict__ pos, const int32_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ adj,
                                                        const int32_t *__restrict__ long_rows,
                                                        const int32_t *__restrict__ eptr, const int32_t *__restrict__ erow,
                                                        int nentries, int64_t row_lo, float L_min, float neg_k,
                                                        float *__restrict__ terms) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nentries) return;
    const int lo = erow[t];   // (a table: the binary search over eptr was up to 12 dependent loads per entry)
    const int i = long_rows[lo];
    const int idx = t - eptr[lo], deg = eptr[lo + 1] - eptr[lo];
    const int64_t y = (uint32_t)adj[rowptr[i] + idx] & 0x7FFFFFFFu;
    float px[LD], py[LD], diff[D];
    gh_load_row<LD>(pos, row_lo + i, px);
    gh_load_row<LD>(pos, y, py);
#pragma unroll
    for (int d = 0; d < D; ++d) diff[d] = py[d] - px[d];
    const float dist = sqrtf(gh_sumsq<D>(diff)) + 1e-6f;
    const float fm = neg_k * (dist - L_min);
#pragma unroll
    for (int d = 0; d < D; ++d) terms[(int64_t)eptr[lo] * D + (int64_t)d * deg + idx] = fm * (diff[d] / dist);
}

template <int D, int LD>
__global__ __launch_bounds__(256) void long_sum_kernel(const float *__restrict__ terms, const int32_t *__restrict__ long_rows,
                                                      const int32_t *__restrict__ eptr, int nlong,
                                                      float *__restrict__ outF, int64_t f_row0) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nlong) return;
    const int deg = eptr[r + 1] - eptr[r];
    const float *tr = terms + (int64_t)eptr[r] * D;
    float F[D], nxt[2][D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        F[d] = 0.0f;
        nxt[0][d] = lane < deg ? tr[(int64_t)d * deg + lane] : 0.0f;
        nxt[1][d] = 64 + lane < deg ? tr[(int64_t)d * deg + 64 + lane] : 0.0f;
    }
    for (int base = 0; base < deg; base += 64) {
        float x[D], t[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float term = nxt[0][d];
            nxt[0][d] = nxt[1][d];
            nxt[1][d] = base + 128 + lane < deg ? tr[(int64_t)d * deg + base + 128 + lane] : 0.0f;
            x[d] = lane == 0 ? F[d] + term : term;
            t[d] = term;
        }
        const int cnt = deg - base < 64 ? deg - base : 64;
        asm volatile("s_nop 4" ::: "memory");   // (the first DPP read of x must not follow its VALU write directly: the compiler does not see into the asm)
        // 63 steps of ONE instruction per coordinate: v_add_f32 with its first operand taken from the
        // left neighbour lane (DPP wave_shr:1); lane 0 has none and, bound_ctrl being off, keeps its
        // value.  Values only travel upwards, so a lane is final after as many steps as its index and
        // the steps past cnt-1 of a short last batch change nothing below lane cnt.  The s_nop keeps the
        // two wait states a DPP read needs after the VALU write of the same register (D < 3).
#pragma unroll
        for (int step = 1; step < 64; ++step) {
#pragma unroll
            for (int d = 0; d < D; ++d)
                asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[d]) : "v"(t[d]));
            asm volatile("s_nop 1");
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
            F[d] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(x[d]), cnt - 1));
    }
    if (lane == 0) {
        float out[LD];
#pragma unroll
        for (int d = 0; d < LD; ++d) out[d] = d < D ? F[d] : 0.0f;
        gh_store_row<LD>(outF, long_rows[r] + f_row0, out);
    }
}

// Form D of a partitioned step: the finished value of an own row the intersection phase touched, pos + (Fs + Fi), goes
// into the rank's patch list -- (row, LD floats) records behind the statistics, all-gathered with them -- because the row
// itself left as pos + Fs before Fi was known.  One returning atomic per touched own row (a few thousand per iteration).
template <int LD>
__device__ __forceinline__ void gh_patch_append(int32_t *count, float *rows, int cap, int64_t x, const float (&nw)[LD]) {
    const int slot = atomicAdd(count, 1);
    if (slot < cap) {
        float *rec = rows + (int64_t)slot * (1 + LD);
        rec[0] = __int_as_float((int)x);
#pragma unroll
        for (int d = 0; d < LD; ++d) rec[1 + d] = nw[d];
    }
}

// Combine (pt.py:796-799): new = pos + (F_spring + F_inter) for the own rows, plus the
// per-workgroup column sums / sums of squares in fp64.  Streaming, one thread per row.
template <int LD>
__global__ __launch_bounds__(256) void integrate_kernel(
    const float *__restrict__ pos, const float *__restrict__ Fs, int64_t row_lo, int64_t rows,
    const double *__restrict__ acc, const int32_t *__restrict__ tflag, float *__restrict__ out,
    double *__restrict__ blockstats, int32_t *__restrict__ patch_count = nullptr /* form D: the rows are already travelling as
    pos + Fs (new0_kernel); a touched row's value goes into the rank's patch list (gh_patch_append) instead of `out` */,
    float *__restrict__ patch_rows = nullptr, int patch_cap = 0) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    double sx[LD], sxx[LD];
#pragma unroll
    for (int d = 0; d < LD; ++d) { sx[d] = 0.0; sxx[d] = 0.0; }
    if (i < rows) {
        const int64_t x = row_lo + i;
        float px[LD], F[LD], nw[LD];
        gh_load_row<LD>(pos, x, px);
        gh_load_row<LD>(Fs, i, F);
        const bool touched = tflag[x] != 0;
#pragma unroll
        for (int d = 0; d < LD; ++d) {
            const float fi = touched ? (float)acc[x * LD + d] : 0.0f;
            const float tot = F[d] + fi;
            nw[d] = px[d] + tot;
            sx[d] = (double)nw[d];
            sxx[d] = (double)nw[d] * (double)nw[d];
        }
        if (!patch_count) gh_store_row<LD>(out, i, nw);
        else if (touched) gh_patch_append<LD>(patch_count, patch_rows, patch_cap, x, nw);
    }
    __shared__ double red[4][2 * LD];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < LD; ++d) {
        const double a = gh_wave_sum(sx[d]), b = gh_wave_sum(sxx[d]);
        if (lane == 0) { red[w][d] = a; red[w][LD + d] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * LD) {
        const double v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
        blockstats[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = v;  // [entry][workgroup]: the reducers read contiguously
    }
}

// Any LD: new = pos + (Fs + Fi), one thread per element (statistics by column_stats_kernel).
__global__ __launch_bounds__(256) void integrate_generic_kernel(
    const float *__restrict__ pos, const float *__restrict__ Fs, int LD, int64_t row_lo, int64_t rows,
    const double *__restrict__ acc, const int32_t *__restrict__ tflag, float *__restrict__ out) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= rows * LD) return;
    const int64_t x = row_lo + t / LD;
    const int64_t g = row_lo * LD + t;
    const float fi = tflag[x] != 0 ? (float)acc[g] : 0.0f;
    const float tot = Fs[t] + fi;
    out[t] = pos[g] + tot;
}

// Form D on an engine whose part 1 did not run the fused kernel: new0 = pos + Fs of the own rows, element by element.
__global__ __launch_bounds__(256) void new0_kernel(const float *__restrict__ pos, const float *__restrict__ Fs, int64_t row_lo,
                                                  int64_t count /* rows * LD */, int LD, float *__restrict__ out) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < count) out[t] = pos[row_lo * LD + t] + Fs[t];
}

// Midpoints of the edges [e_lo, e_lo + M) by gathering both endpoints (used when the edge
// list is not sorted by first endpoint, or a partition does not follow row ownership).
__global__ __launch_bounds__(256) void mid_gather_kernel(const float *__restrict__ pos,
                                                        const int32_t *__restrict__ edges, int64_t e_lo,
                                                        const int32_t *__restrict__ eids, int64_t M, int D, int LD,
                                                        float *__restrict__ mid) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= M * LD) return;
    const int64_t j = t / LD;
    const int d = (int)(t % LD);
    const int64_t e = eids ? (int64_t)eids[j] : e_lo + j;
    const int64_t u = edges[2 * e], v = edges[2 * e + 1];
    mid[t] = d < D ? (pos[u * LD + d] + pos[v * LD + d]) / 2.0f : 0.0f;
}

// Any D: one thread per vertex, rows in global memory, same arithmetic and order.
__global__ __launch_bounds__(256) void spring_generic_kernel(
    const float *__restrict__ pos, int D, int LD, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ adj, int64_t row_lo, int64_t rows, float L_min, float neg_k,
    float *__restrict__ outF, int64_t f_row0, float *__restrict__ scratch /* (rows, LD) diff scratch */) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const int64_t x = row_lo + i;
    float *diff = scratch + i * LD;
    float *dst = outF + (i + f_row0) * LD;
    for (int d = 0; d < LD; ++d) dst[d] = 0.0f;
    for (int j = rowptr[i]; j < rowptr[i + 1]; ++j) {
        const int64_t y = (uint32_t)adj[j] & 0x7FFFFFFFu;  // bit 31 = ownership flag
        for (int d = 0; d < D; ++d) diff[d] = pos[y * LD + d] - pos[x * LD + d];
        const float dist = sqrtf(gh_sumsq_rt(diff, D)) + 1e-6f;
        const float fm = neg_k * (dist - L_min);
        for (int d = 0; d < D; ++d) dst[d] = dst[d] + fm * (diff[d] / dist);
    }
}

// Column sums for the generic path: one workgroup per column chunk, fixed order.
__global__ __launch_bounds__(256) void column_stats_kernel(const float *__restrict__ x, int64_t rows, int D, int LD,
                                                          double *__restrict__ stats) {
    const int d = blockIdx.x;
    double sx = 0.0, sxx = 0.0;
    for (int64_t i = threadIdx.x; i < rows; i += blockDim.x) {
        const double v = (double)x[i * LD + d];
        sx += v;
        sxx += v * v;
    }
    __shared__ double red[2][4];
    const double a = gh_wave_sum(sx), b = gh_wave_sum(sxx);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        stats[d] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        stats[LD + d] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    }
    if (d == 0 && threadIdx.x == 0)
        for (int p = D; p < LD; ++p) { stats[p] = 0.0; stats[LD + p] = 0.0; }
}

// Fixed-order reduction of the per-workgroup partials -> (2, LD) sums.
__global__ __launch_bounds__(256) void stats_reduce_kernel(const double *__restrict__ blockstats, int nblocks, int LD,
                                                          double *__restrict__ stats,
                                                          uint64_t *__restrict__ iter_bump = nullptr /* replayed iterations on the unfused paths: the device's iteration counter (stats_fix_kernel moves it on the fused path) */) {
    if (iter_bump && blockIdx.x == 0 && threadIdx.x == 0) *iter_bump += 1;
    const int c = blockIdx.x;  // column of the (2*LD) record
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) s += blockstats[(int64_t)c * nblocks + b];
    __shared__ double red[4];
    const double a = gh_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) stats[c] = ((red[0] + red[1]) + red[2]) + red[3];
}

// Column sums after the fused kernel: blockstats holds the sums of new0 = pos + Fs; the vertices the
// intersection phase touched must instead be new = pos + (Fs + Fi) (pt.py:796-799).
// gh_fix_blocks(LD) = 2*LD workgroups: each reduces one entry of the (2, LD) statistics each, in a
// fixed order; every workgroup also takes a slice of the touched list, stores the corrected rows
// and leaves its column sums of (new - new0) and (new^2 - new0^2) in its own row pair of the
// statistics buffer (rows 2.. of d_stats); normalise_kernel adds all rows up.  new0 is recomputed
// from pos and Fs, never read back.
template <int LD>
__global__ __launch_bounds__(256) void stats_fix_kernel(const double *__restrict__ blockstats, int nblocks,
                                                       const float *__restrict__ pos, const float *__restrict__ Fs,
                                                       const double *__restrict__ acc,
                                                       const int32_t *__restrict__ touched,
                                                       const int32_t *__restrict__ tcount, int64_t row_lo,
                                                       int64_t rows, float *__restrict__ out_new,
                                                       double *__restrict__ stats, int skip_reduce,
                                                       uint64_t *__restrict__ iter_bump /* replayed iterations: the device's iteration counter, or null */,
                                                       int32_t *__restrict__ patch_count = nullptr /* form D of a partitioned step
                                                       (gh_overlap_layout): new0 is already travelling to the other ranks; the corrected
                                                       rows go into the rank's patch list instead of out_new */,
                                                       float *__restrict__ patch_rows = nullptr, int patch_cap = 0) {
    __shared__ double red[4][2 * LD];
    if (iter_bump && blockIdx.x == 0 && threadIdx.x == 0) *iter_bump += 1;   // read by the set-up inside the NEXT launch (normalise)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (!skip_reduce) {  // (the select launch of a single-rank step has done this part already)
        const int c = blockIdx.x;  // gridDim.x == 2 * LD
        double s = 0.0;
        // four independent partial sums per thread (loads in flight), combined in a fixed order
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        const double *col = blockstats + (int64_t)c * nblocks;
        int b = threadIdx.x;
        for (; b + 3 * (int)blockDim.x < nblocks; b += 4 * blockDim.x) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s4[u] += col[b + u * blockDim.x];
        }
        for (int u = 0; b < nblocks; b += blockDim.x, ++u) s4[u] += col[b];
        s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        const double a = gh_wave_sum(s);
        if (lane == 0) red[w][0] = a;
        __syncthreads();
        if (threadIdx.x == 0) stats[c] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
        __syncthreads();
    }
    double dx[LD], dxx[LD];
#pragma unroll
    for (int d = 0; d < LD; ++d) { dx[d] = 0.0; dxx[d] = 0.0; }
    const int nt = *tcount;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nt; t += gridDim.x * blockDim.x) {
        const int64_t x = touched[t];
        const int64_t i = x - row_lo;
        if (i < 0 || i >= rows) continue;
        float p[LD], f[LD], nw[LD];
        gh_load_row<LD>(pos, x, p);
        gh_load_row<LD>(Fs, i, f);
#pragma unroll
        for (int d = 0; d < LD; ++d) {
            const float n0 = p[d] + f[d];
            const float tot = f[d] + (float)acc[x * LD + d];
            nw[d] = p[d] + tot;
            dx[d] += (double)nw[d] - (double)n0;
            dxx[d] += (double)nw[d] * (double)nw[d] - (double)n0 * (double)n0;
        }
        if (!patch_count) gh_store_row<LD>(out_new, i, nw);
        else gh_patch_append<LD>(patch_count, patch_rows, patch_cap, x, nw);
    }
#pragma unroll
    for (int d = 0; d < LD; ++d) {
        const double a = gh_wave_sum(dx[d]), b = gh_wave_sum(dxx[d]);
        if (lane == 0) { red[w][d] = a; red[w][LD + d] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * LD)
        stats[(2 + 2 * blockIdx.x) * LD + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// pt.py:802-804: centre by the column mean, divide by (unbiased std + 1e-6).
// stats = global (sum, sum of squares) over all n rows; every thread derives the same
// mean / std from them.  Writes rows [row_lo, row_lo+rows) of pos.
// The workgroups past g_norm (single-rank steps only, rows == n) run the NEXT iteration's KNN set-up
// (setup_core.h) from the un-normalised rows: one launch and its dependent loads off the iteration.
// LDT: the row stride as a compile-time constant for the set-up part (4, 8, 16; 0 = any stride, no tiles): one kernel for
// all strides carried the 16-wide set-up's 127 VGPRs -- 4 workgroups per CU, so that at 1 M vertices half of the 2048
// streaming workgroups queued behind the first 1024 (tools/stamp_probe.py: median start 8 us into the launch).
template <int LDT>
__global__ __launch_bounds__(256) void normalise_kernel(const float *__restrict__ nw, int64_t rows, int64_t row_lo,
                                                       int D, int LD, int64_t n, const double *__restrict__ stats,
                                                       float *__restrict__ pos, double *__restrict__ acc,
                                                       int32_t *__restrict__ tflag,
                                                       const int32_t *__restrict__ touched,
                                                       const int32_t *__restrict__ tcount, int nfix, int g_norm,
                                                       int n_setup, gh_setup_args sa, int32_t *__restrict__ qexact,
                                                       unsigned long long *__restrict__ stamps /* diagnostic, or null */,
                                                       int stat_world = 1,
                                                       float *__restrict__ packed = nullptr /* form C: the same rows without pad columns, (rows, D) */) {
    if (stamps && (blockIdx.x >= GH_STAMP_EXTRA)) stamps = nullptr;
    if (stamps) stamps += (int64_t)blockIdx.x * 8;
#define GH_STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[k] = wall_clock64(); } while (0)
    GH_STAMP(0);
    // the set-up workgroups come FIRST in the grid: their chains of dependent gathers start at once and
    // run under the streaming of the others
    __shared__ __align__(16) unsigned char setup_lds[GH_SETUP_LDS_BYTES];
    const bool setup_block = (int)blockIdx.x < n_setup;
    const int nb = (int)blockIdx.x - n_setup;  // index among the normalising workgroups
    // also zero what the intersection phase touched (acc != nullptr): the integrate kernel that
    // read those accumulators has finished; tcount itself is reset by the next KNN setup
    if (acc && !setup_block) {
        const int64_t nt = (int64_t)(*tcount) * LD;
        for (int64_t t = nb * (int64_t)blockDim.x + threadIdx.x; t < nt; t += (int64_t)g_norm * blockDim.x) {
            const int64_t x = touched[t / LD];
            const int d = (int)(t % LD);
            acc[x * LD + d] = 0.0;
            if (d == 0) tflag[x] = 0;
        }
    }
    // set-up workgroups: the rows they will need are requested before anything else -- two levels of dependent cold loads
    // (edge -> rows), which the statistics' own round trip and arithmetic below then run under instead of in front of
    // (16-wide rows: 64 registers held across the prologue would cost the launch its occupancy; they fetch afterwards)
    constexpr int LDS_ = LDT > 0 ? LDT : 4;
    constexpr bool early_fetch = LDT > 0 && LDT <= 8;
    gh_setup_rows<LDS_> srows;
    auto raw_row = [=](int64_t v, float (&row)[LDS_]) { gh_load_row<LDS_>(nw, v, row); };
    if constexpr (early_fetch)
        if (setup_block && sa.tiles > 0) gh_setup_fetch<LDS_>(sa, (int)blockIdx.x, raw_row, srows);
    extern __shared__ float ms[];  // mean[LD], std[LD], then the (2 + 2 nfix, LD) statistics rows as doubles
    // all statistics rows with one load per thread and round, then summed from LDS in the fixed order: a thread adding
    // its column's 2 nfix corrections straight from memory waited for them one after the other (LD = 16: 64 loads, 7.6 us
    // before any workgroup of this launch knew mean and std -- tools/stamp_probe.py)
    // (stat_world > 1: form C of a partitioned step -- the statistics rows of every rank, all-gathered in rank order; each
    // rank's rows in their fixed order, then the next rank's, so every rank derives the same mean / std bits)
    double *srow = reinterpret_cast<double *>(ms + 2 * LD);
    const int R = 2 + 2 * nfix;
    for (int t = threadIdx.x; t < stat_world * R * LD; t += blockDim.x) srow[t] = stats[t];
    __syncthreads();
    for (int d = threadIdx.x; d < LD; d += blockDim.x) {
        float mean = 0.0f, sd = 1.0f;
        if (d < D) {
            double sum = 0.0, sq = 0.0;
            for (int r = 0; r < stat_world; ++r) {
                const double *rr = srow + (int64_t)r * R * LD;
                sum += rr[d];
                sq += rr[LD + d];
#pragma unroll 8
                for (int b = 0; b < nfix; ++b) {  // corrections of the touched rows (zero when unused)
                    sum += rr[(2 + 2 * b) * LD + d];
                    sq += rr[(3 + 2 * b) * LD + d];
                }
            }
            const double m = sum / (double)n;
            double var = (sq - sum * m) / (double)(n - 1);
            if (var < 0.0) var = 0.0;
            mean = (float)m;
            sd = (float)sqrt(var) + 1e-6f;
        }
        ms[d] = mean;
        ms[LD + d] = sd;
    }
    __syncthreads();
    GH_STAMP(1);
    if (stamps && threadIdx.x == 0) stamps[6] = setup_block ? 1 : 2;
    if (setup_block) {
        const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (t == 0) qexact[0] = 0;
        // position of vertex v, component d < D, exactly as the normalising threads below compute it
        auto getp = [=](int64_t v, int d) { return (nw[v * LD + d] - ms[d]) / ms[LD + d]; };
        if constexpr (LDT > 0) {
            if (sa.tiles > 0) {
                auto norm = [=](float x, int d) { return (x - ms[d]) / ms[LD + d]; };
                if constexpr (!early_fetch) gh_setup_fetch<LDS_>(sa, (int)blockIdx.x, raw_row, srows);
                gh_setup_finish<LDS_>(sa, (int)blockIdx.x, raw_row, norm, srows, setup_lds, stamps);
            } else {
                gh_setup_item(sa, t, getp);
            }
        } else {
            gh_setup_item(sa, t, getp);
        }
        GH_STAMP(3);
        return;
    }
    // 16 bytes per thread and step (LD is a multiple of 4, rows are 16-byte aligned)
    const int64_t total4 = rows * LD / 4;
    const float4 *src = reinterpret_cast<const float4 *>(nw);
    float4 *dst = reinterpret_cast<float4 *>(pos + row_lo * LD);
    // four elements per thread and round, all four loads issued before the first division.  (Measured neutral for the
    // launch as a whole: at 1 M vertices the streaming workgroups are done after 13 us, the set-up workgroups after 17 --
    // 10 us of that are their two levels of cold random loads, edge -> rows, and holding the streaming back by 3.4 us
    // did not shorten them: tools/stamp_probe.py.)
    const int64_t step = (int64_t)g_norm * blockDim.x;
    auto norm4 = [&](int64_t t, const float4 &v) {
        const int d0 = (int)((t * 4) % LD);
        float4 o;
        o.x = d0 + 0 < D ? (v.x - ms[d0 + 0]) / ms[LD + d0 + 0] : 0.0f;
        o.y = d0 + 1 < D ? (v.y - ms[d0 + 1]) / ms[LD + d0 + 1] : 0.0f;
        o.z = d0 + 2 < D ? (v.z - ms[d0 + 2]) / ms[LD + d0 + 2] : 0.0f;
        o.w = d0 + 3 < D ? (v.w - ms[d0 + 3]) / ms[LD + d0 + 3] : 0.0f;
        dst[t] = o;
        if (packed) {   // the copy that travels (gh_step_unpack_rows on the receiving ranks)
            float *pk = packed + (t * 4 / LD) * D + d0;
            if (d0 + 0 < D) pk[0] = o.x;
            if (d0 + 1 < D) pk[1] = o.y;
            if (d0 + 2 < D) pk[2] = o.z;
            if (d0 + 3 < D) pk[3] = o.w;
        }
    };
    int64_t t = nb * (int64_t)blockDim.x + threadIdx.x;
    for (; t + 3 * step < total4; t += 4 * step) {
        const float4 v0 = src[t], v1 = src[t + step], v2 = src[t + 2 * step], v3 = src[t + 3 * step];
        norm4(t, v0);
        norm4(t + step, v1);
        norm4(t + 2 * step, v2);
        norm4(t + 3 * step, v3);
    }
    for (; t < total4; t += step) norm4(t, src[t]);
    GH_STAMP(3);
#undef GH_STAMP
}

// The same normalisation for ALL n rows from the gathered slots of every rank (one-collective
// finish, include/graphem_hip.h): slot r = [chunk rows of new positions | that rank's statistics].
// The per-rank statistics are added in rank order, so every rank derives the same mean / std.
// Since round 5 rows and statistics are two strided arrays: form B passes both halves of its slots, form D
// (gh_overlap_layout) the early all-gathered rows -- RS floats per row: LD, or D when they travelled without pad columns
// -- and the late all-gathered statistics; skip_cleanup: form D's patch launch has zeroed the accumulators already.
template <int LDT>
__global__ __launch_bounds__(256) void normalise_gathered_kernel(const float *__restrict__ rows_base, int64_t rows_block /* floats between two ranks' blocks */,
                                                                int RS, const double *__restrict__ stats_base, int64_t stats_block /* doubles */,
                                                                int skip_cleanup,
                                                                int64_t chunk, int world, int D, int LD, int64_t n,
                                                                int nfix, float *__restrict__ pos,
                                                                double *__restrict__ acc, int32_t *__restrict__ tflag,
                                                                const int32_t *__restrict__ touched,
                                                                const int32_t *__restrict__ tcount, int g_norm,
                                                                int n_setup, gh_setup_args sa,
                                                                int32_t *__restrict__ qexact) {
    __shared__ __align__(16) unsigned char setup_lds[GH_SETUP_LDS_BYTES];
    const bool setup_block = (int)blockIdx.x < n_setup;  // the next iteration's KNN set-up, as in normalise_kernel
    const int nb = (int)blockIdx.x - n_setup;
    if (!setup_block && !skip_cleanup) {
        const int64_t nt = (int64_t)(*tcount) * LD;
        for (int64_t t = nb * (int64_t)blockDim.x + threadIdx.x; t < nt; t += (int64_t)g_norm * blockDim.x) {
            const int64_t x = touched[t / LD];
            const int d = (int)(t % LD);
            acc[x * LD + d] = 0.0;
            if (d == 0) tflag[x] = 0;
        }
    }
    extern __shared__ float ms[];  // mean[LD], std[LD], then world * 2 * LD doubles of per-rank totals
    double *part = reinterpret_cast<double *>(ms + 2 * LD);
    // per-rank totals (sum and sum of squares incl. the correction rows), one thread per (rank, entry) ...
    for (int t = threadIdx.x; t < world * 2 * LD; t += blockDim.x) {
        const int r = t / (2 * LD), c = t % (2 * LD), base = c / LD, col = c % LD;
        const double *st = stats_base + (int64_t)r * stats_block;
        double v = st[base * LD + col];
        for (int b = 0; b < nfix; ++b) v += st[(2 + 2 * b + base) * LD + col];
        part[t] = v;
    }
    __syncthreads();
    // ... added in rank order
    for (int d = threadIdx.x; d < LD; d += blockDim.x) {
        float mean = 0.0f, sd = 1.0f;
        if (d < D) {
            double sum = 0.0, sq = 0.0;
            for (int r = 0; r < world; ++r) {
                sum += part[r * 2 * LD + d];
                sq += part[r * 2 * LD + LD + d];
            }
            const double m = sum / (double)n;
            double var = (sq - sum * m) / (double)(n - 1);
            if (var < 0.0) var = 0.0;
            mean = (float)m;
            sd = (float)sqrt(var) + 1e-6f;
        }
        ms[d] = mean;
        ms[LD + d] = sd;
    }
    __syncthreads();
    if (setup_block) {
        const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (t == 0) qexact[0] = 0;
        auto getp = [=](int64_t v, int d) {
            const int64_t r = v / chunk;
            const float *rowsrc = rows_base + r * rows_block;
            return (rowsrc[(v - r * chunk) * RS + d] - ms[d]) / ms[LD + d];
        };
        if constexpr (LDT > 0) {
            if (sa.tiles > 0) gh_setup_block<LDT>(sa, (int)blockIdx.x, getp, setup_lds);
            else gh_setup_item(sa, t, getp);
        } else {
            gh_setup_item(sa, t, getp);
        }
        return;
    }
    const int64_t total4 = n * LD / 4;  // 16 bytes per thread and step
    float4 *dst = reinterpret_cast<float4 *>(pos);
    for (int64_t t = nb * (int64_t)blockDim.x + threadIdx.x; t < total4; t += (int64_t)g_norm * blockDim.x) {
        const int64_t i = t * 4 / LD;
        const int d0 = (int)((t * 4) % LD);
        const int64_t r = i / chunk;
        const float *rp = rows_base + r * rows_block + (i - r * chunk) * RS + d0;
        float4 v;
        if (RS == LD) v = *reinterpret_cast<const float4 *>(rp);
        else {   // rows without pad columns: what exists of this quarter row
            v.x = d0 + 0 < D ? rp[0] : 0.0f;
            v.y = d0 + 1 < D ? rp[1] : 0.0f;
            v.z = d0 + 2 < D ? rp[2] : 0.0f;
            v.w = d0 + 3 < D ? rp[3] : 0.0f;
        }
        float4 o;
        o.x = d0 + 0 < D ? (v.x - ms[d0 + 0]) / ms[LD + d0 + 0] : 0.0f;
        o.y = d0 + 1 < D ? (v.y - ms[d0 + 1]) / ms[LD + d0 + 1] : 0.0f;
        o.z = d0 + 2 < D ? (v.z - ms[d0 + 2]) / ms[LD + d0 + 2] : 0.0f;
        o.w = d0 + 3 < D ? (v.w - ms[d0 + 3]) / ms[LD + d0 + 3] : 0.0f;
        dst[t] = o;
    }
}

// Form C of a partitioned step: every rank's statistics rows (sum, sum of squares, then its correction row pairs),
// all-gathered in rank order -> one (2, LD) record: per rank its rows in their fixed order, then the ranks in rank order,
// so every rank derives the same mean / std bits.  One workgroup, thread = (base, column).
__global__ __launch_bounds__(64) void stats_combine_kernel(const double *__restrict__ all, int world, int R /* rows per rank */,
                                                          int LD, double *__restrict__ out) {
    const int t = threadIdx.x;
    if (t >= 2 * LD) return;
    const int base = t / LD, col = t % LD;
    double tot = 0.0;
    for (int r = 0; r < world; ++r) {
        const double *st = all + (int64_t)r * R * LD;
        double v = st[base * LD + col];
        for (int b = 2 + base; b < R; b += 2) v += st[b * LD + col];
        tot += v;
    }
    out[t] = tot;
}

// pt.py:796-799 with given force arrays (per-phase entry point gh_integrate_normalise).
__global__ __launch_bounds__(256) void integrate_given_kernel(const float *__restrict__ pos,
                                                             const float *__restrict__ Fs,
                                                             const float *__restrict__ Fi, int64_t total,
                                                             float *__restrict__ nw) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    const float tot = Fs[t] + Fi[t];
    nw[t] = pos[t] + tot;
}

// ---------------------------------------------------------------------------------
// Intersection repulsion (pt.py:638-774).  One thread per candidate pair
// (i = sampled[r], j = knn[r][c]): keep i < j (pt.py:672), drop pairs sharing a vertex
// (pt.py:685-692), keep pairs whose projections on coordinates 0,1 strictly cross
// (pt.py:760-772); then each of the four endpoints x gets
//   k_inter * (x - c) / (|x - c| + 1e-6)^2,   c = (((p1 + p2) + q1) + q2) / 4
// (pt.py:722-734).  Contributions are summed in fp64 atomics: with the handful of terms
// a vertex receives the fp64 sum is exact, hence independent of arrival order.
// DT: the dimension at compile time (2..16) or 0 = any (the scratch form); one small kernel per dimension instead of one
// with all fifteen inlined behind a switch (14 000 instructions fetched through a cold instruction cache).
template <int DT>
__global__ __launch_bounds__(256) void intersect_kernel(const float *__restrict__ pos, int D, int LD,
                                                       const int32_t *__restrict__ edges,
                                                       const int32_t *__restrict__ sampled,
                                                       const uint64_t *__restrict__ keys, int64_t S, int k,
                                                       float k_inter, double *__restrict__ acc,
                                                       int32_t *__restrict__ tflag, int32_t *__restrict__ touched,
                                                       int32_t *__restrict__ tcount, float *__restrict__ scratch, int32_t own_lo,
                                                       int32_t own_hi) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= S * k) return;
    const int64_t r = t / k;
    // neighbour c of query r is key column c+1: column 0 is dropped blindly (pt.py:421)
    const int32_t i = sampled[r], j = (int32_t)gh_key_id(keys[r * (k + 1) + (t - r * k) + 1]);
    if constexpr (DT >= 2)
        gh_intersect_pair_t<DT, (DT <= 4 ? 4 : DT <= 8 ? 8 : 16)>(pos, edges, i, j, k_inter, acc, tflag, touched, tcount, own_lo, own_hi);
    else
        gh_intersect_pair(pos, D, LD, edges, i, j, k_inter, acc, tflag, touched, tcount, scratch + t * LD, own_lo, own_hi);
}

// acc (double) -> dense fp32 F for the touched vertices (per-phase entry point).
__global__ void inter_to_dense_kernel(const double *__restrict__ acc, const int32_t *__restrict__ touched,
                                      const int32_t *__restrict__ tcount, int LD, float *__restrict__ F) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= (int64_t)(*tcount) * LD) return;
    const int64_t x = touched[t / LD];
    const int d = (int)(t % LD);
    F[x * LD + d] = (float)acc[x * LD + d];
}

// Zero what the intersection phase touched (keeps acc / tflag all-zero between iterations).
__global__ void inter_cleanup_kernel(double *__restrict__ acc, int32_t *__restrict__ tflag,
                                     const int32_t *__restrict__ touched, const int32_t *__restrict__ tcount, int LD) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= (int64_t)(*tcount) * LD) return;
    const int64_t x = touched[t / LD];
    const int d = (int)(t % LD);
    acc[x * LD + d] = 0.0;
    if (d == 0) tflag[x] = 0;
}

__global__ void reset_counter_kernel(int32_t *c) { *c = 0; }

// ---------------------------------------------------------------------------------
// (n, D) <-> (n, LD) copies for the host boundary.
// Vertex arrays cross the API as (n, D) in the caller's vertex order; on the device they are
// (n, LD) rows in the internal order (order[v] = row of vertex v; null = identity).
__global__ void pad_kernel(const float *__restrict__ src, int64_t n, int D, int LD, const int32_t *__restrict__ order,
                           float *__restrict__ dst) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n * LD) return;
    const int64_t i = t / LD;
    const int d = (int)(t % LD);
    const int64_t row = order ? order[i] : i;
    dst[row * LD + d] = d < D ? src[i * D + d] : 0.0f;
}
__global__ void unpad_kernel(const float *__restrict__ src, int64_t n, int D, int LD, const int32_t *__restrict__ order,
                             float *__restrict__ dst) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n * D) return;
    const int64_t i = t / D;
    const int d = (int)(t % D);
    const int64_t row = order ? order[i] : i;
    dst[t] = src[row * LD + d];
}

// ---------------------------------------------------------------------------------
// Stand-alone sampler launch (the hot path samples inside knn_setup_kernel instead).
__global__ void sample_kernel(int64_t E, int64_t S, uint64_t seed, uint64_t iter, int32_t *__restrict__ sampled) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < S) sampled[t] = gh_sample_id(E, seed, iter, t);
}
__global__ void arange_kernel(int64_t S, int32_t *__restrict__ sampled) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < S) sampled[t] = (int32_t)t;
}

// Self-test of common.h's lean square root and division against sqrtf and '/': pseudo-random operands over and beyond
// their fast domain (exponents, zeros, values one ulp around exact squares, |n| up to d), mismatching bit patterns counted.
__global__ __launch_bounds__(256) void arith_selftest_kernel(uint64_t seed, int64_t per_thread, unsigned long long *__restrict__ bad) {
    uint64_t x = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    auto rnd_float = [&](int emin, int emax) {   // random sign-less float with biased exponent in [emin, emax]
        const uint64_t r = next();
        const uint32_t e = (uint32_t)(emin + (int)((r >> 40) % (uint64_t)(emax - emin + 1)));
        return __uint_as_float((e << 23) | (uint32_t)(r & 0x7FFFFFu));
    };
    unsigned long long bs = 0, bd = 0;
    for (int64_t it = 0; it < per_thread; ++it) {
        // square root: any magnitude; every 4th sample sits within 2 ulp of an exact square
        float v = rnd_float(1, 254);
        if ((it & 3) == 0) { const float t = rnd_float(64, 190); v = __uint_as_float(__float_as_uint(t * t) + (uint32_t)(next() % 5) - 2u); }
        if ((it & 63) == 0) v = (it & 64) ? 0.0f : __uint_as_float((uint32_t)(next() & 0x7FFFFFu));   // zero, denormals
        if (__float_as_uint(gh_sqrt_ieee(v)) != __float_as_uint(sqrtf(v))) ++bs;
        // division: d around and beyond [2^-40, 2^20], numerators from 2^-110 d up to d, sometimes zero / denormal
        const float d = rnd_float(127 - 44, 127 + 24);
        float n[3], q[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint64_t r = next();
            float m = rnd_float(1, 127) * d;                            // |n| <= d, down to denormal
            if ((r & 7) == 0) m = d * __uint_as_float(0x3F000000u + (uint32_t)((r >> 8) & 0x7FFFFFu));   // same magnitude as d
            if ((r & 255) == 1) m = 0.0f;
            n[i] = (r >> 63) ? -m : m;
        }
        gh_div_by<3>(n, d, q);
#pragma unroll
        for (int i = 0; i < 3; ++i) if (__float_as_uint(q[i]) != __float_as_uint(n[i] / d)) ++bd;
    }
    if (bs) atomicAdd(&bad[0], bs);
    if (bd) atomicAdd(&bad[1], bd);
}

inline unsigned grid_for(int64_t total, int bs) { return (unsigned)((total + bs - 1) / bs); }

bool spring_is_templated(int D) { return gh_dim_templated(D); }

// WRITE_MID only when the engine's edge range follows row ownership (h->fused_mid).
template <bool WRITE_MID>
gh_status launch_spring(gh_engine *h, float *outF, int64_t f_row0) {
    const unsigned grid = grid_for(h->rows, 256);
    const float neg_k = -h->prm.k_attr;
    const gh_long_args la = gh_make_long_args(h);
#define GH_SPRING_ARGS h->d_pos, h->d_rowptr, h->d_adj, h->d_first_edge, h->mid_base, h->part.row_lo, h->rows, \
                       h->prm.L_min, neg_k, outF, f_row0, h->d_mid, la
#define GH_SPRING_CASE(DD, LL)                                                                              \
    if (la.n > 0) spring_kernel<DD, LL, WRITE_MID, true><<<dim3(grid), dim3(256), 0, h->stream>>>(GH_SPRING_ARGS); \
    else spring_kernel<DD, LL, WRITE_MID, false><<<dim3(grid), dim3(256), 0, h->stream>>>(GH_SPRING_ARGS)
    GH_TRY_ST(gh_launch_spring_long(h, outF, f_row0));  // hubs first: spring_row reads their forces back
#define GH_SPRING_ONE(DD, LL) case DD: GH_SPRING_CASE(DD, LL); break;
    switch (h->D) {
        GH_FOR_EACH_DIM(GH_SPRING_ONE)
        default:
            spring_generic_kernel<<<dim3(grid), dim3(256), 0, h->stream>>>(
                h->d_pos, h->D, h->LD, h->d_rowptr, h->d_adj, h->part.row_lo, h->rows, h->prm.L_min, neg_k, outF,
                f_row0, h->d_tmpF2);
    }
#undef GH_SPRING_ONE
#undef GH_SPRING_CASE
#undef GH_SPRING_ARGS
    GH_LAUNCH_CHECK();
    return GH_OK;
}

}  // namespace

static bool spring_is_templated_d(int D) { return gh_dim_templated(D); }

gh_long_args gh_make_long_args(const gh_engine *h, bool coop_mid) {
    if (h->nlong == 0 || !spring_is_templated_d(h->D)) return gh_long_args{nullptr, nullptr, nullptr, 0, h->long_deg, nullptr, nullptr, nullptr};
    const bool coop = coop_mid && h->d_own_long && h->d_own_eids;
    return gh_long_args{h->d_long_rows, h->d_long_ownptr, h->d_long_ownadj, h->nlong, h->long_deg,
                        coop ? h->d_own_long : nullptr, h->d_own_eids, h->d_edges};
}

// Spring forces of the long own rows -> outF rows (i + f_row0); no-op for graphs without hubs.
// Both steps in one launch for rows of moderate length (every row of a small dense graph, common.h GH_LONG_DEG_DENSE): one
// wave per long row computes the terms of 64 list entries in registers -- lane = entry, the arithmetic of spring_pull -- and
// adds them in list order with the same DPP chain; the neighbour rows of the next 64 entries are in flight during the
// chain.  No terms array, one launch less (the SNAP shape at 16 components: 27.3 -> 17.4 us, 112 -> 102 us per iteration).  A hub of thousands of
// neighbours has only one wave's gathers in flight this way, so graphs with such rows keep the two launches.
#define GH_LONG_ONE_LAUNCH_MAX_DEG 1024
template <int D, int LD>
__global__ __launch_bounds__(256) void long_rows_kernel(const float *__restrict__ pos, const int32_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ adj, const int32_t *__restrict__ long_rows,
                                                       const int32_t *__restrict__ eptr, int nlong, int64_t row_lo, float L_min,
                                                       float neg_k, float *__restrict__ outF, int64_t f_row0) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nlong) return;
    const int i = long_rows[r];
    const int deg = eptr[r + 1] - eptr[r];
    const int32_t *list = adj + rowptr[i];
    float px[LD], py[LD], F[D];
    gh_load_row<LD>(pos, row_lo + i, px);
    auto fetch = [&](int idx) {   // neighbour row of list entry idx (own row past the end: a zero term)
        const int64_t y = idx < deg ? (int64_t)((uint32_t)list[idx] & 0x7FFFFFFFu) : row_lo + i;
        gh_load_row<LD>(pos, y, py);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) F[d] = 0.0f;
    fetch(lane);
    for (int base = 0; base < deg; base += 64) {
        float x[D], t[D], diff[D];
#pragma unroll
        for (int d = 0; d < D; ++d) diff[d] = py[d] - px[d];
        const bool live = base + lane < deg;
        const float dist = sqrtf(gh_sumsq<D>(diff)) + 1e-6f;
        const float fm = neg_k * (dist - L_min);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            t[d] = live ? fm * (diff[d] / dist) : 0.0f;
            x[d] = lane == 0 ? F[d] + t[d] : t[d];
        }
        if (base + 64 < deg) fetch(base + 64 + lane);   // in flight during the chain below
        const int cnt = deg - base < 64 ? deg - base : 64;
        asm volatile("s_nop 4" ::: "memory");   // the chain's first DPP read of x comes right behind the VALU writes above when the fetch is skipped
#pragma unroll
        for (int step = 1; step < 64; ++step) {
#pragma unroll
            for (int d = 0; d < D; ++d)
                asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[d]) : "v"(t[d]));
            asm volatile("s_nop 1");
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
            F[d] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(x[d]), cnt - 1));
    }
    if (lane == 0) {
        float out[LD];
#pragma unroll
        for (int d = 0; d < LD; ++d) out[d] = d < D ? F[d] : 0.0f;
        gh_store_row<LD>(outF, long_rows[r] + f_row0, out);
    }
}

gh_status gh_launch_spring_long(gh_engine *h, float *outF, int64_t f_row0) {
    const gh_long_args la = gh_make_long_args(h);
    if (la.n == 0) return GH_OK;
    gh_scope t(h, "spring_long");
    const float neg_k = -h->prm.k_attr;
    if (h->long_max_deg <= GH_LONG_ONE_LAUNCH_MAX_DEG) {
#define GH_LONG_FUSED(DD, LL)                                                                                                 \
    case DD:                                                                                                                  \
        long_rows_kernel<DD, LL><<<dim3((unsigned)((la.n + 3) / 4)), dim3(256), 0, h->stream>>>(                                \
            h->d_pos, h->d_rowptr, h->d_adj, la.rows, h->d_long_eptr, la.n, h->part.row_lo, h->prm.L_min, neg_k, outF, f_row0); \
        break;
        switch (h->D) {
            GH_FOR_EACH_DIM(GH_LONG_FUSED)
            default: break;
        }
#undef GH_LONG_FUSED
        GH_LAUNCH_CHECK();
        return GH_OK;
    }
#define GH_LONG_CASE(DD, LL)                                                                                          \
    long_terms_kernel<DD, LL><<<dim3(grid_for(h->long_entries, 256)), dim3(256), 0, h->stream>>>(                       \
        h->d_pos, h->d_rowptr, h->d_adj, la.rows, h->d_long_eptr, h->d_long_erow, (int)h->long_entries, h->part.row_lo,     \
        h->prm.L_min, neg_k, h->d_long_terms);                                                                          \
    long_sum_kernel<DD, LL><<<dim3((unsigned)((la.n + 3) / 4)), dim3(256), 0, h->stream>>>(h->d_long_terms, la.rows,   \
                                                                                          h->d_long_eptr, la.n, outF, f_row0)
#define GH_LONG_ONE(DD, LL) case DD: GH_LONG_CASE(DD, LL); break;
    switch (h->D) {
        GH_FOR_EACH_DIM(GH_LONG_ONE)
        default: break;  // gh_make_long_args returns none for other dimensions
    }
#undef GH_LONG_ONE
#undef GH_LONG_CASE
    GH_LAUNCH_CHECK();
    return GH_OK;
}

namespace {

gh_status launch_mid_gather(gh_engine *h) {
    const int64_t M = h->own_count;
    if (M == 0) return GH_OK;
    gh_scope t(h, "mid_gather");
    mid_gather_kernel<<<dim3(grid_for(M * h->LD, 256)), dim3(256), 0, h->stream>>>(
        h->d_pos, h->d_edges, h->part.edge_lo, h->d_own_eids, M, h->D, h->LD, h->d_mid);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

}  // namespace

// Spring forces of the own rows -> d_Fs, and the midpoints of the own edges -> d_mid.
gh_status gh_launch_spring_mid(gh_engine *h) {
    const bool fused = h->fused_mid && spring_is_templated(h->D);
    if (h->rows > 0) {
        gh_scope t(h, fused ? "spring_mid" : "spring");
        gh_status st = fused ? launch_spring<true>(h, h->d_Fs, 0) : launch_spring<false>(h, h->d_Fs, 0);
        if (st) return st;
    }
    if (!fused) return launch_mid_gather(h);
    return GH_OK;
}

// Midpoints only (per-phase KNN entry point).
gh_status gh_launch_mid_only(gh_engine *h) { return launch_mid_gather(h); }

// new = pos + (Fs + Fi) for the own rows -> d_new, column statistics -> d_stats.
gh_status gh_launch_integrate(gh_engine *h) {
    if (h->new0_ready && h->rows > 0) {  // the fused kernel already wrote pos + Fs and its partial sums
        h->new0_ready = false;
        gh_scope t(h, "stats_fix");
        const struct reset_flag { gh_engine *e; ~reset_flag() { e->stats_reduced = false; } } reset{h};
#define GH_FIX_CASE(LL)                                                                                      \
    stats_fix_kernel<LL><<<dim3(gh_fix_blocks(LL)), dim3(256), 0, h->stream>>>(                                  \
        h->d_blockstats, h->n_vblocks, h->d_pos, h->d_Fs, h->d_acc, h->d_touched, h->d_tcount, h->part.row_lo, \
        h->rows, h->d_new, h->d_stats, h->stats_reduced ? 1 : 0, h->graph_capturing ? h->d_iter : nullptr, h->overlap ? gh_patch_count(h) + (h->iter & 1) : nullptr, \
        h->overlap ? gh_patch_records(h) : nullptr, (int)h->patch_cap)
        if (h->LD == 4) GH_FIX_CASE(4);
        else if (h->LD == 8) GH_FIX_CASE(8);
        else GH_FIX_CASE(16);
#undef GH_FIX_CASE
        GH_LAUNCH_CHECK();
        return GH_OK;
    }
    h->new0_ready = false;
    // unfused path: no correction rows
    GH_HIP(hipMemsetAsync(h->d_stats + 2 * h->LD, 0, sizeof(double) * 2 * gh_fix_blocks(h->LD) * h->LD, h->stream));
    if (h->rows == 0) {
        GH_HIP(hipMemsetAsync(h->d_stats, 0, sizeof(double) * 2 * h->LD, h->stream));
        return GH_OK;
    }
    const unsigned grid = grid_for(h->rows, 256);
    // form D: the own block (d_new) holds new0 and is on its way to the other ranks: statistics as always, the touched rows'
    // values into the patch list (row strides of 4, 8, 16 floats: gh_overlap_layout refuses the others)
    int32_t *pc = h->overlap ? gh_patch_count(h) + (h->iter & 1) : nullptr;   // (two counters, alternate iterations: patch_rows_kernel)
    float *pr = h->overlap ? gh_patch_records(h) : nullptr;
    const int pcap = (int)h->patch_cap;
    float *wide_out = h->d_new;
    {
        gh_scope t(h, "integrate");
        switch (h->LD) {
            case 4:
                integrate_kernel<4><<<dim3(grid), dim3(256), 0, h->stream>>>(h->d_pos, h->d_Fs, h->part.row_lo, h->rows,
                                                                            h->d_acc, h->d_tflag, h->d_new, h->d_blockstats, pc, pr, pcap);
                break;
            case 8:
                integrate_kernel<8><<<dim3(grid), dim3(256), 0, h->stream>>>(h->d_pos, h->d_Fs, h->part.row_lo, h->rows,
                                                                            h->d_acc, h->d_tflag, h->d_new, h->d_blockstats, pc, pr, pcap);
                break;
            case 16:
                integrate_kernel<16><<<dim3(grid), dim3(256), 0, h->stream>>>(h->d_pos, h->d_Fs, h->part.row_lo, h->rows,
                                                                             h->d_acc, h->d_tflag, h->d_new, h->d_blockstats, pc, pr, pcap);
                break;
            default:
                integrate_generic_kernel<<<dim3(grid_for(h->rows * h->LD, 256)), dim3(256), 0, h->stream>>>(
                    h->d_pos, h->d_Fs, h->LD, h->part.row_lo, h->rows, h->d_acc, h->d_tflag, wide_out);
        }
        GH_LAUNCH_CHECK();
    }
    gh_scope t(h, "stats_reduce");
    if (h->LD <= 16) {
        stats_reduce_kernel<<<dim3(2 * h->LD), dim3(256), 0, h->stream>>>(h->d_blockstats, h->nblocks_update, h->LD,
                                                                          h->d_stats, h->graph_capturing ? h->d_iter : nullptr);
    } else {
        column_stats_kernel<<<dim3(h->D), dim3(256), 0, h->stream>>>(wide_out, h->rows, h->D, h->LD, h->d_stats);
    }
    GH_LAUNCH_CHECK();
    return GH_OK;
}

gh_status gh_launch_spring_only(gh_engine *h, float *d_F) {
    GH_HIP(hipMemsetAsync(d_F, 0, sizeof(float) * h->n * h->LD, h->stream));
    if (h->rows == 0) return GH_OK;
    gh_scope t(h, "spring_only");
    return launch_spring<false>(h, d_F, h->part.row_lo);
}

gh_status gh_launch_intersect(gh_engine *h) {
    const int64_t P = h->S * h->k;
    if (P == 0) return GH_OK;
    // a row partition accumulates only what lands on its own rows
    const bool own_only = h->rows != h->n;
    const int32_t own_lo = own_only ? (int32_t)h->part.row_lo : 0, own_hi = own_only ? (int32_t)h->part.row_hi : 0x7FFFFFFF;
    gh_scope t(h, "intersect");
#define GH_INTER_ONE(DD, LL)                                                                                       \
    case DD:                                                                                                       \
        intersect_kernel<DD><<<dim3(grid_for(P, 256)), dim3(256), 0, h->stream>>>(                                   \
            h->d_pos, h->D, h->LD, h->d_edges, h->d_sampled_cur, h->d_keys_cur, h->S, h->k, h->prm.k_inter, h->d_acc, \
            h->d_tflag, h->d_touched, h->d_tcount, h->d_iscratch, own_lo, own_hi);                                 \
        break;
    switch (h->D) {
        GH_FOR_EACH_DIM(GH_INTER_ONE)
        default:
            intersect_kernel<0><<<dim3(grid_for(P, 256)), dim3(256), 0, h->stream>>>(
                h->d_pos, h->D, h->LD, h->d_edges, h->d_sampled_cur, h->d_keys_cur, h->S, h->k, h->prm.k_inter, h->d_acc,
                h->d_tflag, h->d_touched, h->d_tcount, h->d_iscratch, own_lo, own_hi);
    }
#undef GH_INTER_ONE
    GH_LAUNCH_CHECK();
    return GH_OK;
}

gh_status gh_launch_inter_to_dense(gh_engine *h, float *d_F) {
    GH_HIP(hipMemsetAsync(d_F, 0, sizeof(float) * h->n * h->LD, h->stream));
    const int64_t maxT = 4 * h->S * h->k * h->LD;
    if (maxT == 0) return GH_OK;
    inter_to_dense_kernel<<<dim3(grid_for(maxT, 256)), dim3(256), 0, h->stream>>>(h->d_acc, h->d_touched, h->d_tcount,
                                                                                  h->LD, d_F);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

gh_status gh_launch_inter_cleanup(gh_engine *h) {
    const int64_t maxT = 4 * h->S * h->k * h->LD;
    if (maxT == 0) return GH_OK;
    gh_scope t(h, "inter_cleanup");
    inter_cleanup_kernel<<<dim3(grid_for(maxT, 256)), dim3(256), 0, h->stream>>>(h->d_acc, h->d_tflag, h->d_touched,
                                                                                 h->d_tcount, h->LD);
    reset_counter_kernel<<<dim3(1), dim3(1), 0, h->stream>>>(h->d_tcount);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

gh_status gh_launch_integrate_given(gh_engine *h, const float *d_Fs, const float *d_Fi) {
    GH_HIP(hipMemsetAsync(h->d_stats, 0, sizeof(double) * (2 + 2 * gh_fix_blocks(h->LD)) * h->LD, h->stream));
    // whole graph on one rank only (per-phase entry point)
    const int64_t total = h->n * h->LD;
    integrate_given_kernel<<<dim3(grid_for(total, 256)), dim3(256), 0, h->stream>>>(h->d_pos, d_Fs, d_Fi, total,
                                                                                    h->d_new);
    column_stats_kernel<<<dim3(h->D), dim3(256), 0, h->stream>>>(h->d_new, h->n, h->D, h->LD, h->d_stats);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

// presetup: also run the next iteration's KNN set-up in this launch (single-rank fused steps), for the
// sample source the caller expects then: mode 0 = the ids at next_ids, 1 = device sampler, 2 = arange.
gh_status gh_launch_normalise(gh_engine *h, bool with_cleanup, bool presetup, int next_mode, int32_t *next_ids) {
    h->presetup_valid = false;  // positions change: whatever set-up was done ahead is stale
    if (h->rows == 0) return with_cleanup ? gh_launch_inter_cleanup(h) : GH_OK;
    gh_scope t(h, "normalise");
    const int64_t total = h->rows * h->LD / 4;  // float4 elements
    unsigned grid = grid_for(total, 1024);  // four elements per thread and round (normalise_kernel)
    if (grid > 2048) grid = 2048;
    gh_setup_args sa{};
    unsigned extra = 0;
    if (presetup) {
        sa = gh_make_setup_args(h, next_mode, next_mode == 0 ? next_ids : h->d_sampled, h->iter + 1);
        extra = gh_setup_blocks(sa);
    }
    const size_t smem = sizeof(float) * 2 * h->LD + sizeof(double) * (size_t)(2 + 2 * gh_fix_blocks(h->LD)) * h->LD;
#define GH_NORM(LL)                                                                                         \
    normalise_kernel<LL><<<dim3(grid + extra), dim3(256), smem, h->stream>>>(                                    \
        h->d_new, h->rows, h->part.row_lo, h->D, h->LD, h->n, h->d_stats, h->d_pos,                              \
        with_cleanup ? h->d_acc : nullptr, h->d_tflag, h->d_touched, h->d_tcount, gh_fix_blocks(h->LD), (int)grid, \
        (int)extra, sa, h->d_qexact, h->d_stamps ? h->d_stamps + (int64_t)std::max(h->n_vblocks, 1) * 8 : nullptr)
    if (h->LD == 4) GH_NORM(4);
    else if (h->LD == 8) GH_NORM(8);
    else if (h->LD == 16) GH_NORM(16);
    else {
        if (sa.tiles > 0) { h->err = "KNN set-up tiles need a row stride of 4, 8 or 16"; return GH_ERR_RUNTIME; }
        GH_NORM(0);
    }
#undef GH_NORM
    GH_LAUNCH_CHECK();
    if (presetup) {
        h->presetup_valid = true;
        h->presetup_mode = next_mode;
        h->presetup_ids = next_mode == 0 ? next_ids : h->d_sampled;
        h->presetup_iter = h->iter + 1;
    }
    return GH_OK;
}

// Form C: normalise the own rows into their block of d_pos from the ranks' gathered statistics; also zeroes what the
// intersection phase touched.  No set-up for the next iteration here: the other ranks' rows arrive with the caller's
// all-gather of the position blocks.
gh_status gh_launch_normalise_own(gh_engine *h, const double *stats_all, int world) {
    h->presetup_valid = false;
    const int R = 2 + 2 * gh_fix_blocks(h->LD);
    int nfix = gh_fix_blocks(h->LD), sworld = world;
    size_t smem = sizeof(float) * 2 * h->LD + sizeof(double) * (size_t)world * R * h->LD;
    if (smem > 48 * 1024) {   // wide rows on many ranks: the rows do not fit the workgroup's LDS -- added up by a launch of their own
        if (2 * h->LD > 64) { h->err = "row stride too large for the partitioned finish"; return GH_ERR_INVALID; }
        gh_scope t(h, "stats_combine");
        stats_combine_kernel<<<dim3(1), dim3(64), 0, h->stream>>>(stats_all, world, R, h->LD, h->d_stats_comb);
        GH_LAUNCH_CHECK();
        stats_all = h->d_stats_comb;
        nfix = 0;
        sworld = 1;
        smem = sizeof(float) * 2 * h->LD + sizeof(double) * 2 * (size_t)h->LD;
    }
    if (h->rows == 0) return gh_launch_inter_cleanup(h);
    gh_scope t(h, "normalise_own");
    const int64_t total = h->rows * h->LD / 4;
    unsigned grid = grid_for(total, 1024);
    if (grid > 2048) grid = 2048;
    gh_setup_args sa{};
    normalise_kernel<0><<<dim3(grid), dim3(256), smem, h->stream>>>(
        h->d_new, h->rows, h->part.row_lo, h->D, h->LD, h->n, stats_all, h->d_pos, h->d_acc, h->d_tflag, h->d_touched,
        h->d_tcount, nfix, (int)grid, 0, sa, h->d_qexact, nullptr, sworld,
        h->packed_exchange ? h->d_rows_packed + (size_t)h->g_rank * h->g_chunk * h->D : nullptr);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

// Form C with fewer components than the row stride: the blocks travel without their pad columns.  After the all-gather of
// d_rows_packed (world, chunk, D): the other ranks' rows -> d_pos (the own block was written in place by the normalise launch).
__global__ __launch_bounds__(256) void unpack_rows_kernel(const float *__restrict__ packed, int64_t n, int D, int LD, int64_t own_lo,
                                                         int64_t own_hi, float *__restrict__ pos) {
    const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;   // element (row, d) of the padded array
    if (t >= n * LD) return;
    const int64_t row = t / LD;
    const int d = (int)(t % LD);
    if (row >= own_lo && row < own_hi) return;
    pos[t] = d < D ? packed[row * D + d] : 0.0f;
}
gh_status gh_launch_unpack_rows(gh_engine *h) {
    if (!h->packed_exchange) return GH_OK;
    gh_scope t(h, "unpack_rows");
    unpack_rows_kernel<<<dim3(grid_for(h->n * h->LD, 256)), dim3(256), 0, h->stream>>>(h->d_rows_packed, h->n, h->D, h->LD, h->part.row_lo,
                                                                                       h->part.row_hi, h->d_pos);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

// next_mode >= 0: also the next iteration's KNN set-up (as gh_launch_normalise).
gh_status gh_launch_normalise_gathered(gh_engine *h, int next_mode) {
    h->presetup_valid = false;
    gh_scope t(h, "normalise_gathered");
    unsigned grid = grid_for(h->n * h->LD / 4, 256);
    if (grid > 2048) grid = 2048;
    const bool presetup = next_mode >= 0 && h->fused_scan && gh_knn_scan_path(h) && h->S > 0 && h->k > 0 && !h->opt_no_presetup;
    gh_setup_args sa{};
    unsigned extra = 0;
    if (presetup) {
        sa = gh_make_setup_args(h, next_mode, h->d_sampled, h->iter + 1);
        extra = gh_setup_blocks(sa);
    }
    const size_t smem = sizeof(float) * 2 * h->LD + sizeof(double) * 2 * h->LD * (size_t)h->g_world;
    // form B: both halves of the gathered slots; form D: the early-gathered rows (packed or not) and the late-gathered statistics
    const float *rows_base = h->overlap ? (h->d_rows_pk ? h->d_rows_pk : h->d_rows_all) : reinterpret_cast<const float *>(h->d_gbuf);
    const int RS = h->overlap && h->d_rows_pk ? h->D : h->LD;
    const int64_t rows_block = h->overlap ? h->g_chunk * RS : h->g_slot / (int64_t)sizeof(float);
    const double *stats_base = h->overlap ? h->d_stats_all : reinterpret_cast<const double *>(h->d_gbuf + h->g_chunk * h->LD * (int64_t)sizeof(float));
    const int64_t stats_block = h->overlap ? h->stats_block : h->g_slot / (int64_t)sizeof(double);
    const int skip_cleanup = h->overlap && h->rows_early ? 1 : 0;
#define GH_NORMG(LL)                                                                                              \
    normalise_gathered_kernel<LL><<<dim3(grid + extra), dim3(256), smem, h->stream>>>(                                   \
        rows_base, rows_block, RS, stats_base, stats_block, skip_cleanup,                                                  \
        h->g_chunk, h->g_world, h->D, h->LD, h->n, gh_fix_blocks(h->LD), h->d_pos, h->d_acc,         \
        h->d_tflag, h->d_touched, h->d_tcount, (int)grid, (int)extra, sa, h->d_qexact)
    if (h->LD == 4) GH_NORMG(4);
    else if (h->LD == 8) GH_NORMG(8);
    else if (h->LD == 16) GH_NORMG(16);
    else {
        if (sa.tiles > 0) { h->err = "KNN set-up tiles need a row stride of 4, 8 or 16"; return GH_ERR_RUNTIME; }
        GH_NORMG(0);
    }
#undef GH_NORMG
    GH_LAUNCH_CHECK();
    if (presetup) {
        h->presetup_valid = true;
        h->presetup_mode = next_mode;
        h->presetup_ids = h->d_sampled;
        h->presetup_iter = h->iter + 1;
    }
    return GH_OK;
}

// ---- form D of a partitioned step (gh_overlap_layout) ----------------------------------------------------------------
// The own block of new0 = pos + Fs without its pad columns -> its slot of the (world, chunk, D) array that travels.
__global__ __launch_bounds__(256) void pack_rows_kernel(const float *__restrict__ src /* (rows, LD) */, int64_t rows, int D, int LD,
                                                       float *__restrict__ dst /* (rows, D) */) {
    const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (t >= rows * D) return;
    dst[t] = src[(t / D) * LD + t % D];
}
gh_status gh_launch_new0(gh_engine *h) {
    if (h->rows == 0) return GH_OK;
    gh_scope t(h, "new0");
    new0_kernel<<<dim3(grid_for(h->rows * h->LD, 256)), dim3(256), 0, h->stream>>>(h->d_pos, h->d_Fs, h->part.row_lo, h->rows * h->LD, h->LD, h->d_new);
    GH_LAUNCH_CHECK();
    return GH_OK;
}
gh_status gh_launch_pack_rows(gh_engine *h, hipStream_t stream) {
    if (!h->d_rows_pk || h->rows == 0) return GH_OK;
    gh_scope t(h, "pack_rows", stream);
    // (chunk rows, not only the real ones: the block of a rank with fewer rows must not travel with stale bytes)
    pack_rows_kernel<<<dim3(grid_for(h->g_chunk * h->D, 256)), dim3(256), 0, stream>>>(h->d_new, h->g_chunk, h->D, h->LD,
                                                                                        h->d_rows_pk + (size_t)h->g_rank * h->g_chunk * h->D);
    GH_LAUNCH_CHECK();
    return GH_OK;
}
// After both all-gathers: every rank's patch list -- the rows its intersection phase touched, as their owner finished them
// (pos + (Fs + Fi), the single engine's expression) -- is written over those rows of the gathered new0 array (blockIdx.y =
// the rank whose list); the own accumulators are zeroed.  The list has TWO counters, used by alternate iterations: this
// launch reads counter `par` of every rank and zeroes the own OTHER one for the next iteration (no launch of its own, no
// counter that is read and reset in one launch).
__global__ __launch_bounds__(256) void patch_rows_kernel(float *__restrict__ rows /* (world * chunk, RS) */, int RS, int D, int LD,
                                                        double *__restrict__ stats_all, int64_t stats_block, int stat_doubles,
                                                        int rank, int cap, int par,
                                                        double *__restrict__ acc, int32_t *__restrict__ tflag,
                                                        const int32_t *__restrict__ touched, const int32_t *__restrict__ tcount) {
    const int64_t T = (int64_t)gridDim.x * blockDim.x, t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    int32_t *hdr = reinterpret_cast<int32_t *>(stats_all + (int64_t)r * stats_block + stat_doubles);
    const float *rec = reinterpret_cast<const float *>(hdr + 4);
    const int64_t cnt = min(hdr[par], cap);
    for (int64_t t = t0; t < cnt * D; t += T) {
        const int64_t j = t / D;
        const int d = (int)(t % D);
        const int64_t x = __float_as_int(rec[j * (1 + LD)]);
        rows[x * RS + d] = rec[j * (1 + LD) + 1 + d];
    }
    if (r != rank) return;
    if (t0 == 0) hdr[par ^ 1] = 0;
    const int64_t nt = (int64_t)(*tcount) * LD;
    for (int64_t t = t0; t < nt; t += T) {
        const int64_t x = touched[t / LD];
        const int d = (int)(t % LD);
        acc[x * LD + d] = 0.0;
        if (d == 0) tflag[x] = 0;
    }
}
gh_status gh_launch_patch_rows(gh_engine *h) {
    gh_scope t(h, "patch_rows");
    const int64_t maxT = std::max<int64_t>(4 * h->S * h->k * h->LD, 256);
    unsigned grid = grid_for(maxT, 256);
    if (grid > 64) grid = 64;
    const int stat_doubles = (2 + 2 * gh_fix_blocks(h->LD)) * h->LD;
    patch_rows_kernel<<<dim3(grid, (unsigned)h->g_world), dim3(256), 0, h->stream>>>(
        h->d_rows_pk ? h->d_rows_pk : h->d_rows_all, h->d_rows_pk ? h->D : h->LD, h->D, h->LD, h->d_stats_all, h->stats_block, stat_doubles,
        h->g_rank, (int)h->patch_cap, (int)(h->iter & 1), h->d_acc, h->d_tflag, h->d_touched, h->d_tcount);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

gh_status gh_launch_pad(gh_engine *h, const float *d_src_nD, float *d_dst_nLD) {
    pad_kernel<<<dim3(grid_for(h->n * h->LD, 256)), dim3(256), 0, h->stream>>>(d_src_nD, h->n, h->D, h->LD, h->d_order, d_dst_nLD);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

gh_status gh_launch_unpad(gh_engine *h, const float *d_src_nLD, float *d_dst_nD) {
    unpad_kernel<<<dim3(grid_for(h->n * h->D, 256)), dim3(256), 0, h->stream>>>(d_src_nLD, h->n, h->D, h->LD, h->d_order, d_dst_nD);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

gh_status gh_launch_sample(gh_engine *h) {
    gh_scope t(h, "sample");
    sample_kernel<<<dim3(grid_for(h->S, 256)), dim3(256), 0, h->stream>>>(h->E, h->S, h->prm.seed, h->iter, h->d_sampled);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

gh_status gh_launch_arange(gh_engine *h) {
    arange_kernel<<<dim3(grid_for(h->S, 256)), dim3(256), 0, h->stream>>>(h->S, h->d_sampled);
    GH_LAUNCH_CHECK();
    return GH_OK;
}

gh_status gh_ensure_sample(gh_engine *h) {
    if (!h->sample_pending) return GH_OK;
    h->sample_pending = false;
    return h->sample_mode == 2 ? gh_launch_arange(h) : gh_launch_sample(h);
}

// include/graphem_hip.h: self-test of the lean IEEE square root / division of the spring phase.
extern "C" gh_status gh_selftest_arith(int device_id, uint64_t seed, int64_t samples, int64_t *bad_sqrt, int64_t *bad_div) {
    if (!bad_sqrt || !bad_div || samples < 0) return GH_ERR_INVALID;
    if (hipSetDevice(device_id) != hipSuccess) return GH_ERR_HIP;
    unsigned long long *d_bad = nullptr, host[2] = {0, 0};
    if (hipMalloc(reinterpret_cast<void **>(&d_bad), sizeof(host)) != hipSuccess) return GH_ERR_NOMEM;
    gh_status st = GH_OK;
    const unsigned blocks = 4096;
    const int64_t per_thread = (samples + (int64_t)blocks * 256 - 1) / ((int64_t)blocks * 256);
    if (hipMemset(d_bad, 0, sizeof(host)) != hipSuccess) st = GH_ERR_HIP;
    if (st == GH_OK) {
        arith_selftest_kernel<<<dim3(blocks), dim3(256)>>>(seed, per_thread, d_bad);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(host, d_bad, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) st = GH_ERR_HIP;
    }
    (void)hipFree(d_bad);
    *bad_sqrt = (int64_t)host[0];
    *bad_div = (int64_t)host[1];
    return st;
}
