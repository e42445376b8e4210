#pragma once
// The per-iteration KNN set-up as work items, shared by knn_setup_kernel (knn.hip: reads the
// positions) and normalise_kernel (forces.hip: runs the NEXT iteration's set-up in the same launch,
// taking every position it needs as normalise(new row) -- the same arithmetic the normalising
// threads apply, hence the same bits -- so it does not wait for them).
//   item t <  S           : sample id t of the iteration (given / device sampler / arange, pt.py:403-413),
//                           query record = midpoint of that edge (pt.py:785, pt.py:410) + tau = inf,
//                           candidate-list and overflow reset
//   item t >= S           : element (t - S) of the compact (M1, LD) midpoint subset the threshold kernel
//                           streams: every stride-th own edge
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
    int64_t M1, stride;
    float *midsub;
};

template <class P /* float(int64_t vertex, int d) */>
__device__ __forceinline__ void gh_setup_item(const gh_setup_args &a, int64_t t, P getp) {
    if (t < a.S) {
        int32_t e32;
        if (a.mode == 1) { e32 = gh_sample_id(a.E, a.seed, a.iter, t); a.sampled[t] = e32; }
        else if (a.mode == 2) { e32 = (int32_t)t; a.sampled[t] = e32; }
        else e32 = a.sampled[t];
        const int QS = gh_qs(a.D, a.LD);
        const int64_t e = e32;
        const int64_t u = a.edges[2 * e], v = a.edges[2 * e + 1];
        for (int d = 0; d < QS; ++d) a.qt[t * QS + d] = d < a.D ? (getp(u, d) + getp(v, d)) / 2.0f : 0.0f;
        a.qt[t * QS + gh_qtau(a.D, a.LD)] = INFINITY;
        a.cnt[t * GH_CNT_STRIDE] = 0;
        a.ovf[t] = 0;
        return;
    }
    const int64_t g = t - a.S;
    if (g >= a.M1 * a.LD) return;
    const int64_t j = g / a.LD;
    const int d = (int)(g % a.LD);
    const int64_t e = a.own_eids ? (int64_t)a.own_eids[j * a.stride] : a.e_lo + j * a.stride;
    const int64_t u = a.edges[2 * e], v = a.edges[2 * e + 1];
    a.midsub[g] = d < a.D ? (getp(u, d) + getp(v, d)) / 2.0f : 0.0f;
}
