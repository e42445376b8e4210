/*
 * aten_cdist_topk.cpp -- the reference's KNN as ATen computes it on the CPU:
 * torch.cdist(q, ref, p=2) followed by torch.topk(dist, k+1, largest=False)
 * (reference graphem_rapids/backends/embedder_pytorch.py:580-583, "pt.py").
 *
 * TEST INFRASTRUCTURE ONLY (see graphem_oracle.c).  The product computes the KNN
 * with exact-difference distances (what the reference's own KeOps path computes,
 * pt.py:531); this restatement exists to QUANTIFY what the reference's cdist
 * rounding does to the neighbour lists at full size and to keep the oracle's
 * trajectory on the reference's, bit for bit, across several steps.
 *
 * The algorithm lives in a third-party dependency of the reference, PyTorch
 * (here 2.10.0, CPU build, AVX-512, MKL 2024.2); restated from its published source:
 *   - ATen/native/Distance.cpp, _euclidean_dist (taken by cdist for p = 2 when either
 *     side has more than 25 rows): x1_ = [-2 x, |x|^2, 1], x2_ = [y, 1, |y|^2],
 *     result = x1_ @ x2_^T, clamp_min(0), sqrt.  |x|^2 = x.pow(2).sum(-1): products
 *     rounded, added left to right.  The sgemm with inner dimension D + 2 accumulates
 *     acc = fma(a_k, b_k, acc) for k = 0 .. D+1 from acc = 0 -- pinned empirically:
 *     equal to torch's result BIT FOR BIT on 256 x 100000 random rows (D = 3), and
 *     tests/test_oracle_reference_fullsize.py reproduces the reference's neighbour
 *     ids, ties included, on every captured step of the 100 K and 1 M fixtures;
 *   - ATen/native/TopKImpl.h, topk_impl_loop: a vector of (value, index) pairs in
 *     index order; k * 64 <= n: std::partial_sort of the first k with the comparator
 *     "x.first < y.first" (NaN last); else std::nth_element(k - 1) + std::sort of
 *     the first k - 1.  Ties therefore come out in libstdc++'s heap order, which is
 *     what this file reproduces by calling the same std:: algorithms.
 * Smaller inputs (both sides <= 25 rows) take ATen's direct kernel instead; they are
 * restated as sqrt of the exact-difference sum (no golden case depends on them).
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <utility>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" int go_knn_midpoints_aten(const float *pos, int D, const int32_t *edges, int64_t E,
                                     const int32_t *sampled, int64_t S, int k, int32_t *knn_out,
                                     int32_t *col0_out /* (S) dropped column 0, or NULL */) {
    const int64_t K = (int64_t)k + 1;
    if (K > E) return 1; /* GO_ERR_K_TOO_LARGE */
    std::vector<float> mid((size_t)E * D), nrm((size_t)E);
    for (int64_t e = 0; e < E; ++e) {
        const float *p1 = pos + (size_t)edges[2 * e] * D, *p2 = pos + (size_t)edges[2 * e + 1] * D;
        float s = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float m = (p1[d] + p2[d]) / 2.0f; /* pt.py:785 */
            mid[(size_t)e * D + d] = m;
            const float sq = m * m;                 /* pow(2): rounded on its own */
            s = s + sq;                             /* sum(-1): left to right */
        }
        nrm[(size_t)e] = s;
    }
    const bool mm_form = S > 25 || E > 25;
    int threads = 1;
#ifdef _OPENMP
    threads = omp_get_max_threads();
    if (threads > 16) threads = 16; /* every thread holds an (E) pair queue: 16 bytes per edge */
#endif
    auto less = [](const std::pair<float, int64_t> &x, const std::pair<float, int64_t> &y) -> bool {
        return ((!std::isnan(x.first) && std::isnan(y.first)) || (x.first < y.first));
    };
#pragma omp parallel num_threads(threads)
    {
        std::vector<std::pair<float, int64_t>> queue((size_t)E);
        std::vector<float> a((size_t)D);
#pragma omp for schedule(dynamic, 1)
        for (int64_t r = 0; r < S; ++r) {
            const float *q = mid.data() + (size_t)sampled[r] * D;
            const float qn = nrm[(size_t)sampled[r]];
            for (int d = 0; d < D; ++d) a[(size_t)d] = q[d] * -2.0f; /* x1.mul(-2) */
            for (int64_t e = 0; e < E; ++e) {
                const float *m = mid.data() + (size_t)e * D;
                float v;
                if (mm_form) {
                    float acc = 0.0f;
                    for (int d = 0; d < D; ++d) acc = std::fmaf(a[(size_t)d], m[d], acc);
                    acc = std::fmaf(qn, 1.0f, acc);
                    acc = std::fmaf(1.0f, nrm[(size_t)e], acc);
                    v = std::sqrt(acc < 0.0f ? 0.0f : acc);
                } else {
                    float acc = 0.0f;
                    for (int d = 0; d < D; ++d) {
                        const float t = q[d] - m[d];
                        acc = std::fmaf(t, t, acc);
                    }
                    v = std::sqrt(acc);
                }
                queue[(size_t)e].first = v;
                queue[(size_t)e].second = e;
            }
            if (K * 64 <= E) {
                std::partial_sort(queue.begin(), queue.begin() + K, queue.end(), less);
            } else {
                std::nth_element(queue.begin(), queue.begin() + K - 1, queue.end(), less);
                std::sort(queue.begin(), queue.begin() + K - 1, less);
            }
            if (col0_out) col0_out[r] = (int32_t)queue[0].second;
            for (int64_t c = 1; c < K; ++c) knn_out[(size_t)r * k + (c - 1)] = (int32_t)queue[(size_t)c].second; /* pt.py:421 */
        }
    }
    return 0;
}

/* The (S, E) cdist values themselves, as ATen computes them (the formula of go_knn_midpoints_aten above): what a rank of a
 * row-partitioned run ranks its own edges by (tests/cpu_shard_engine.py, the CPU stand-in of a rank's engine). */
extern "C" int go_cdist_values_aten(const float *pos, int D, const int32_t *edges, int64_t E, const int32_t *sampled, int64_t S,
                                    float *out /* (S, E) */) {
    std::vector<float> mid((size_t)E * D), nrm((size_t)E);
    for (int64_t e = 0; e < E; ++e) {
        const float *p1 = pos + (size_t)edges[2 * e] * D, *p2 = pos + (size_t)edges[2 * e + 1] * D;
        float s = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float m = (p1[d] + p2[d]) / 2.0f;
            mid[(size_t)e * D + d] = m;
            const float sq = m * m;
            s = s + sq;
        }
        nrm[(size_t)e] = s;
    }
    const bool mm_form = S > 25 || E > 25;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < S; ++r) {
        const float *q = mid.data() + (size_t)sampled[r] * D;
        const float qn = nrm[(size_t)sampled[r]];
        for (int64_t e = 0; e < E; ++e) {
            const float *m = mid.data() + (size_t)e * D;
            float acc = 0.0f;
            if (mm_form) {
                for (int d = 0; d < D; ++d) acc = std::fmaf(q[d] * -2.0f, m[d], acc);
                acc = std::fmaf(qn, 1.0f, acc);
                acc = std::fmaf(1.0f, nrm[(size_t)e], acc);
                out[(size_t)r * E + e] = std::sqrt(acc < 0.0f ? 0.0f : acc);
            } else {
                for (int d = 0; d < D; ++d) { const float t = q[d] - m[d]; acc = std::fmaf(t, t, acc); }
                out[(size_t)r * E + e] = std::sqrt(acc);
            }
        }
    }
    return 0;
}
