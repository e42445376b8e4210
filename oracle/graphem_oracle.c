/*
 * graphem_oracle.c -- CPU restatement of the reference's force-directed layout
 * iteration (the hot path behind GraphEmbedderPyTorch.run_layout()).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (graphem-rapids_amd/)
 * may import, link or execute this file; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, as the checker / reported baseline.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against the .npz files under tests/golden/, which tests/golden/make_golden.py produced
 * by running the reference's PyTorch-CPU backend in the build container.
 *
 * All citations are to the reference file graphem_rapids/backends/embedder_pytorch.py
 * ("pt.py").  Arithmetic is fp32 with one rounding per torch op (the reference
 * issues every elementwise op as its own ATen kernel); build with
 * -ffp-contract=off so the compiler cannot fuse what torch does not fuse.
 * Reductions (norm / mean / std) have an ATen-internal order; the orders used
 * here were chosen by matching the golden vectors (see go_norm2 below).
 *
 * Layout: positions (n, D) row-major f32; edges (E, 2) row-major int32 with
 * u < v in CSR row order (pt.py:220-245); sampled (S,) int32; knn (S, k) int32.
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define GO_OK 0
#define GO_ERR_K_TOO_LARGE 1 /* torch.topk raises when k+1 > E */
#define GO_ERR_NOMEM 2

/* torch.norm(x, dim=1) for one row of D floats (pt.py:623, pt.py:731).
 * ATen's CPU kernel (AVX2 build, torch 2.10) reduces a contiguous row as:
 *   full groups of 8: eight lane accumulators, acc[l] = fma(x, x, acc[l]);
 *   the 8 lanes are then added left to right;
 *   then groups of 4: s = s + x*x (product rounded first), left to right;
 *   then the last D%4 elements: s = fma(x, x, s).
 * This order was pinned empirically: it equals torch.norm BIT FOR BIT for every
 * D in 1..69, 127, 128, 250, 300 on 20 000 random rows each, and reproduces the
 * golden spring / intersection forces bit for bit
 * (tests/test_oracle_golden.py).  For D <= 3 it is a plain fma chain. */
static inline float go_norm2(const float *x, int D) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int d = 0;
    for (; d + 8 <= D; d += 8)
        for (int l = 0; l < 8; ++l) acc[l] = fmaf(x[d + l], x[d + l], acc[l]);
    float s = acc[0];
    for (int l = 1; l < 8; ++l) s = s + acc[l];
    for (; d + 4 <= D; d += 4)
        for (int l = 0; l < 4; ++l) s = s + x[d + l] * x[d + l];
    for (; d < D; ++d) s = fmaf(x[d], x[d], s);
    return sqrtf(s);
}

/* pt.py:595-636  _compute_spring_forces
 * diff = p[v]-p[u]; dist = |diff| + 1e-6; fm = -k_attr*(dist-L_min);
 * f = fm*(diff/dist); F[u] += f for all edges, THEN F[v] -= f for all edges
 * (two sequential index_add_ calls, pt.py:633-634). */
void go_spring_forces(const float *pos, int64_t n, int D, const int32_t *edges, int64_t E,
                      float L_min, float k_attr, float *F) {
    memset(F, 0, (size_t)n * D * sizeof(float));
    const float neg_k = -k_attr; /* python: -self.k_attr * tensor */
    float *diff = (float *)malloc(sizeof(float) * (size_t)D);
    for (int pass = 0; pass < 2; ++pass) {
        for (int64_t e = 0; e < E; ++e) {
            const int32_t u = edges[2 * e], v = edges[2 * e + 1];
            const float *p1 = pos + (size_t)u * D, *p2 = pos + (size_t)v * D;
            for (int d = 0; d < D; ++d) diff[d] = p2[d] - p1[d];
            const float dist = go_norm2(diff, D) + 1e-6f;
            const float fm = neg_k * (dist - L_min);
            float *dst = F + (size_t)(pass == 0 ? u : v) * D;
            for (int d = 0; d < D; ++d) {
                const float f = fm * (diff[d] / dist);
                if (pass == 0) dst[d] = dst[d] + f;
                else dst[d] = dst[d] + (-f);
            }
        }
    }
    free(diff);
}

/* pt.py:785  midpoints = (pos[e0] + pos[e1]) / 2.0 */
void go_midpoints(const float *pos, int D, const int32_t *edges, int64_t E, float *mid) {
    for (int64_t e = 0; e < E; ++e) {
        const float *p1 = pos + (size_t)edges[2 * e] * D, *p2 = pos + (size_t)edges[2 * e + 1] * D;
        for (int d = 0; d < D; ++d) mid[(size_t)e * D + d] = (p1[d] + p2[d]) / 2.0f;
    }
}

/* Squared distance between two midpoints in exact-difference form
 * (what pt.py:531 / KeOps computes; torch.cdist's matmul form pt.py:580 agrees
 * with it up to its own cancellation error).  fma chain in
 * coordinate order: the HIP kernel uses the same chain, so d2 is bit-identical. */
static inline float go_d2(const float *a, const float *b, int D) {
    float s = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float t = a[d] - b[d];
        s = fmaf(t, t, s);
    }
    return s;
}

/* pt.py:381-424 + pt.py:543-593: for each sampled midpoint the k+1 nearest
 * midpoints (ascending distance, ties by smaller edge id), column 0 dropped
 * blindly (pt.py:421).  knn_out is (S, k).  If dist_out != NULL it
 * receives the (S, k+1) squared distances including column 0. */
int go_knn_midpoints(const float *pos, int D, const int32_t *edges, int64_t E,
                     const int32_t *sampled, int64_t S, int k, int32_t *knn_out, float *dist_out) {
    const int K = k + 1;
    if ((int64_t)K > E) return GO_ERR_K_TOO_LARGE;
    float *mid = (float *)malloc(sizeof(float) * (size_t)E * D);
    if (!mid) return GO_ERR_NOMEM;
    go_midpoints(pos, D, edges, E, mid);
    int err = GO_OK;
#pragma omp parallel
    {
        float *bd = (float *)malloc(sizeof(float) * (size_t)K);
        int32_t *bi = (int32_t *)malloc(sizeof(int32_t) * (size_t)K);
#pragma omp for schedule(static)
        for (int64_t r = 0; r < S; ++r) {
            const float *q = mid + (size_t)sampled[r] * D;
            int cnt = 0;
            for (int64_t e = 0; e < E; ++e) {
                const float d2 = go_d2(q, mid + (size_t)e * D, D);
                if (cnt == K && !(d2 < bd[K - 1])) continue; /* ties keep the smaller id (earlier e) */
                int j = cnt < K ? cnt : K - 1;
                while (j > 0 && bd[j - 1] > d2) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; --j; }
                bd[j] = d2; bi[j] = (int32_t)e;
                if (cnt < K) ++cnt;
            }
            for (int c = 1; c < K; ++c) knn_out[(size_t)r * k + (c - 1)] = bi[c];
            if (dist_out) for (int c = 0; c < K; ++c) dist_out[(size_t)r * K + c] = bd[c];
        }
        free(bd); free(bi);
    }
    free(mid);
    return err;
}

/* The same search with torch.cdist's matmul formulation (pt.py:580; ATen
 * _euclidean_dist: x1_=[-2x, |x|^2, 1], x2_=[y, 1, |y|^2], clamp_min(0), sqrt).
 * Used only by the oracle-vs-golden test to show how far the reference's own
 * rounding moves the neighbour sets; the summation order of the real sgemm is
 * not knowable, so this is an approximation of an approximation. */
int go_knn_midpoints_cdist_mm(const float *pos, int D, const int32_t *edges, int64_t E,
                              const int32_t *sampled, int64_t S, int k, int32_t *knn_out) {
    const int K = k + 1;
    if ((int64_t)K > E) return GO_ERR_K_TOO_LARGE;
    float *mid = (float *)malloc(sizeof(float) * (size_t)E * D);
    float *nrm = (float *)malloc(sizeof(float) * (size_t)E);
    float *bd = (float *)malloc(sizeof(float) * (size_t)K);
    int32_t *bi = (int32_t *)malloc(sizeof(int32_t) * (size_t)K);
    go_midpoints(pos, D, edges, E, mid);
    for (int64_t e = 0; e < E; ++e) {
        float s = 0.0f;
        for (int d = 0; d < D; ++d) s = s + mid[(size_t)e * D + d] * mid[(size_t)e * D + d];
        nrm[e] = s;
    }
    for (int64_t r = 0; r < S; ++r) {
        const float *q = mid + (size_t)sampled[r] * D;
        const float qn = nrm[sampled[r]];
        int cnt = 0;
        for (int64_t e = 0; e < E; ++e) {
            float s = 0.0f;
            for (int d = 0; d < D; ++d) s = fmaf(-2.0f * q[d], mid[(size_t)e * D + d], s);
            s = s + qn;
            s = s + nrm[e];
            const float dist = sqrtf(s < 0.0f ? 0.0f : s);
            if (cnt == K && !(dist < bd[K - 1])) continue;
            int j = cnt < K ? cnt : K - 1;
            while (j > 0 && bd[j - 1] > dist) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; --j; }
            bd[j] = dist; bi[j] = (int32_t)e;
            if (cnt < K) ++cnt;
        }
        for (int c = 1; c < K; ++c) knn_out[(size_t)r * k + (c - 1)] = bi[c];
    }
    free(mid); free(nrm); free(bd); free(bi);
    return GO_OK;
}

/* pt.py:760-763  orientation of (a, b, c) on coordinates 0 and 1 only. */
static inline float go_orient(const float *a, const float *b, const float *c) {
    return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]);
}

/* pt.py:638-736  _compute_intersection_forces (+ pt.py:738-774).
 * Candidates (i=sampled[r], j=knn[r][c]) in row-major order; keep i<j
 * (pt.py:672), drop pairs sharing a vertex (pt.py:685-692), keep pairs whose
 * 2-D projections strictly cross (pt.py:772).  Then for the four endpoints in
 * the order p1, p2, q1, q2, one sequential index_add_ each (pt.py:727-734):
 * F[x] += k_inter*(x-c)/(|x-c|+1e-6)^2 with c = (p1+p2+q1+q2)/4. */
void go_intersection_forces(const float *pos, int64_t n, int D, const int32_t *edges,
                            const int32_t *sampled, int64_t S, const int32_t *knn, int k,
                            float k_inter, float *F, int64_t *n_pairs_out) {
    memset(F, 0, (size_t)n * D * sizeof(float));
    const int64_t P = S * k;
    int32_t *pi = (int32_t *)malloc(sizeof(int32_t) * (size_t)(P > 0 ? P : 1));
    int32_t *pj = (int32_t *)malloc(sizeof(int32_t) * (size_t)(P > 0 ? P : 1));
    int64_t m = 0;
    for (int64_t r = 0; r < S; ++r) {
        for (int c = 0; c < k; ++c) {
            const int32_t i = sampled[r], j = knn[(size_t)r * k + c];
            if (!(i < j)) continue;
            const int32_t a0 = edges[2 * i], a1 = edges[2 * i + 1], b0 = edges[2 * j], b1 = edges[2 * j + 1];
            if (a0 == b0 || a0 == b1 || a1 == b0 || a1 == b1) continue;
            const float *p1 = pos + (size_t)a0 * D, *p2 = pos + (size_t)a1 * D;
            const float *q1 = pos + (size_t)b0 * D, *q2 = pos + (size_t)b1 * D;
            if (D < 2) continue; /* the reference indexes coordinate 1; D==1 raises there */
            const float o1 = go_orient(p1, p2, q1), o2 = go_orient(p1, p2, q2);
            const float o3 = go_orient(q1, q2, p1), o4 = go_orient(q1, q2, p2);
            if (o1 * o2 < 0.0f && o3 * o4 < 0.0f) { pi[m] = i; pj[m] = j; ++m; }
        }
    }
    if (n_pairs_out) *n_pairs_out = m;
    float *cen = (float *)malloc(sizeof(float) * (size_t)D);
    float *diff = (float *)malloc(sizeof(float) * (size_t)D);
    for (int role = 0; role < 4; ++role) {
        for (int64_t t = 0; t < m; ++t) {
            const int32_t a0 = edges[2 * pi[t]], a1 = edges[2 * pi[t] + 1];
            const int32_t b0 = edges[2 * pj[t]], b1 = edges[2 * pj[t] + 1];
            const float *p1 = pos + (size_t)a0 * D, *p2 = pos + (size_t)a1 * D;
            const float *q1 = pos + (size_t)b0 * D, *q2 = pos + (size_t)b1 * D;
            for (int d = 0; d < D; ++d) cen[d] = (((p1[d] + p2[d]) + q1[d]) + q2[d]) / 4.0f;
            const int32_t vid = role == 0 ? a0 : role == 1 ? a1 : role == 2 ? b0 : b1;
            const float *x = pos + (size_t)vid * D;
            for (int d = 0; d < D; ++d) diff[d] = x[d] - cen[d];
            const float dist = go_norm2(diff, D) + 1e-6f;
            const float dd = dist * dist; /* dist ** 2 */
            float *dst = F + (size_t)vid * D;
            for (int d = 0; d < D; ++d) dst[d] = dst[d] + (k_inter * diff[d]) / dd;
        }
    }
    free(cen); free(diff); free(pi); free(pj);
}

/* torch.sum(x, dim=0) of a contiguous (n, D) f32 tensor, as ATen's CPU
 * cascade_sum does it (AVX2 build, torch 2.10; pinned empirically against
 * torch.sum bit for bit, n up to 1e5, D in {2,3,4,5,8,16,20,33,40,70,250,300}):
 * multi_row_sum = 4-level cascade of fp32 partial sums over the rows;
 * row_sum = the column viewed as (n/4, 4), four interleaved cascades, the n%4
 * leftover rows added to partial 0, then partials 1..3 added in order.
 * Columns take multi_row_sum directly in groups of 32 (D >= 8) or 4 (D < 8);
 * the remaining columns take row_sum.  (For n*D >= 32768 ATen may split the
 * COLUMNS over threads, which can regroup them when D >= 4; D <= 3 is immune.) */
static int go_ceil_log2(int64_t x) {
    int l = 0;
    while (((int64_t)1 << l) < x) ++l;
    return l;
}

static void go_multi_row_sum(const float *base, int64_t row_stride, int64_t col_stride, int64_t n, int K,
                             float *out /* K */) {
    enum { L = 4 };
    int lp = go_ceil_log2(n) / L;
    if (lp < 4) lp = 4;
    const int64_t step = (int64_t)1 << lp, mask = step - 1;
    float *acc = (float *)calloc((size_t)L * K, sizeof(float));
    int64_t i = 0;
    while (i + step <= n) {
        for (int64_t j = 0; j < step; ++j, ++i)
            for (int c = 0; c < K; ++c) acc[c] = acc[c] + base[i * row_stride + c * col_stride];
        for (int j = 1; j < L; ++j) {
            for (int c = 0; c < K; ++c) { acc[j * K + c] = acc[j * K + c] + acc[(j - 1) * K + c]; acc[(j - 1) * K + c] = 0.0f; }
            if ((i & (mask << (j * lp))) != 0) break;
        }
    }
    for (; i < n; ++i)
        for (int c = 0; c < K; ++c) acc[c] = acc[c] + base[i * row_stride + c * col_stride];
    for (int j = 1; j < L; ++j)
        for (int c = 0; c < K; ++c) acc[c] = acc[c] + acc[j * K + c];
    memcpy(out, acc, sizeof(float) * (size_t)K);
    free(acc);
}

static float go_row_sum(const float *base, int64_t stride, int64_t n) {
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t q = n / 4;
    if (q > 0) go_multi_row_sum(base, stride * 4, stride, q, 4, ps);
    for (int64_t i = q * 4; i < n; ++i) ps[0] = ps[0] + base[i * stride];
    for (int k = 1; k < 4; ++k) ps[0] = ps[0] + ps[k];
    return ps[0];
}

void go_column_sums(const float *x, int64_t n, int D, float *out /* D */) {
    int j = 0;
    const int grp = D >= 8 ? 32 : 4;
    for (; j + grp <= D; j += grp) go_multi_row_sum(x + j, D, 1, n, grp, out + j);
    for (; j < D; ++j) out[j] = go_row_sum(x + j, D, n);
}

/* The same sum when the (n, D) tensor is COLUMN-major in memory, which is what
 * the reference's trajectory uses after its Laplacian initialisation
 * (torch.tensor(eigenvectors[:, 1:k]) keeps the Fortran strides, and every
 * later (n, D) tensor inherits them; pt.py:365-376).  ATen then reduces each
 * column as a contiguous row (vectorized_inner_sum): the column viewed as
 * (n/8, 8) lanes, lane sums by row_sum over the 8-wide vectors, the n%8
 * leftover elements summed first, then lanes 0..7 added in order.
 * col(i) is read from the caller's row-major array at x[i*D + j]. */
static float go_inner_sum(const float *x, int64_t stride, int64_t n) {
    const int64_t nv = n / 8;
    float lanes[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (nv > 0) {
        /* row_sum over vectors: vectors viewed as (nv/4, 4) */
        float ps[4][8];
        memset(ps, 0, sizeof ps);
        const int64_t q = nv / 4;
        for (int k = 0; k < 4 && q > 0; ++k)
            go_multi_row_sum(x + (int64_t)k * 8 * stride, stride * 32, stride, q, 8, ps[k]);
        for (int64_t v = q * 4; v < nv; ++v)
            for (int l = 0; l < 8; ++l) ps[0][l] = ps[0][l] + x[(v * 8 + l) * stride];
        for (int k = 1; k < 4; ++k)
            for (int l = 0; l < 8; ++l) ps[0][l] = ps[0][l] + ps[k][l];
        memcpy(lanes, ps[0], sizeof lanes);
    }
    float acc = 0.0f;
    for (int64_t i = nv * 8; i < n; ++i) acc = acc + x[i * stride];
    for (int l = 0; l < 8; ++l) acc = acc + lanes[l];
    return acc;
}

void go_column_sums_colmajor(const float *x, int64_t n, int D, float *out /* D */) {
    if (n < 8) { /* ATen falls to the scalar path when the reduced size is below one vector */
        for (int j = 0; j < D; ++j) out[j] = go_row_sum(x + j, D, n);
        return;
    }
    for (int j = 0; j < D; ++j) out[j] = go_inner_sum(x + j, D, n);
}

/* pt.py:796-804: new = pos + (F_s + F_i); new -= mean(new, 0);
 * pos' = new / (std_unbiased(new, 0) + 1e-6).  torch.mean = torch.sum / n in
 * fp32 (go_column_sums / go_column_sums_colmajor above, by the memory layout
 * the reference's tensors have: column-major after a Laplacian start, row-major
 * after a positions-setter or random start).  torch.std on CPU is Welford with
 * double accumulators over the rows in order, rounded to fp32 once. */
void go_integrate_normalise(const float *pos, const float *Fs, const float *Fi, int64_t n, int D,
                            int colmajor, float *out, float *mean_out, float *std_out) {
    float *sum = (float *)malloc(sizeof(float) * (size_t)D);
    for (size_t o = 0; o < (size_t)n * D; ++o) {
        const float tot = Fs[o] + Fi[o];
        out[o] = pos[o] + tot;
    }
    if (colmajor) go_column_sums_colmajor(out, n, D, sum);
    else go_column_sums(out, n, D, sum);
    for (int d = 0; d < D; ++d) {
        const float mean = sum[d] / (float)n;
        double wm = 0.0, m2 = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            const size_t o = (size_t)i * D + d;
            out[o] = out[o] - mean;
            const double x = (double)out[o], delta = x - wm;
            wm += delta / (double)(i + 1);
            m2 += delta * (x - wm);
        }
        const float sd = (float)sqrt(m2 / (double)(n - 1)) + 1e-6f;
        for (int64_t i = 0; i < n; ++i) out[(size_t)i * D + d] = out[(size_t)i * D + d] / sd;
        if (mean_out) mean_out[d] = mean;
        if (std_out) std_out[d] = sd;
    }
    free(sum);
}

/* pt.py:776-806  update_positions: one iteration on pos (in place).
 * sampled: the S edge ids drawn by torch.randperm(E)[:S] (pt.py:409) or
 * arange(E) when S >= E (pt.py:412); drawing them is the caller's job. */
int go_step(float *pos, int64_t n, int D, const int32_t *edges, int64_t E,
            const int32_t *sampled, int64_t S, int k, float L_min, float k_attr, float k_inter, int colmajor) {
    float *Fs = (float *)malloc(sizeof(float) * (size_t)n * D);
    float *Fi = (float *)malloc(sizeof(float) * (size_t)n * D);
    float *nw = (float *)malloc(sizeof(float) * (size_t)n * D);
    int32_t *knn = (int32_t *)malloc(sizeof(int32_t) * (size_t)(S * k > 0 ? S * k : 1));
    int err = GO_ERR_NOMEM;
    if (Fs && Fi && nw && knn) {
        go_spring_forces(pos, n, D, edges, E, L_min, k_attr, Fs);
        err = go_knn_midpoints(pos, D, edges, E, sampled, S, k, knn, NULL);
        if (err == GO_OK) {
            go_intersection_forces(pos, n, D, edges, sampled, S, knn, k, k_inter, Fi, NULL);
            go_integrate_normalise(pos, Fs, Fi, n, D, colmajor, nw, NULL, NULL);
            memcpy(pos, nw, sizeof(float) * (size_t)n * D);
        }
    }
    free(Fs); free(Fi); free(nw); free(knn);
    return err;
}

/* pt.py:808-833  run_layout: iters x go_step with a caller-supplied sample
 * stream (iters, S) int32. */
int go_run_layout(float *pos, int64_t n, int D, const int32_t *edges, int64_t E,
                  const int32_t *sample_stream, int64_t S, int iters, int k,
                  float L_min, float k_attr, float k_inter, int colmajor) {
    for (int t = 0; t < iters; ++t) {
        int err = go_step(pos, n, D, edges, E, sample_stream + (size_t)t * S, S, k, L_min, k_attr, k_inter, colmajor);
        if (err) return err;
    }
    return GO_OK;
}

/* ---- bench-only multi-threaded variant (bench.py cpu_baseline "port_omp") --------------------------------
 * The strongest honest CPU number for the same algorithm: every phase over all cores.  Spring forces as a
 * per-vertex PULL over lists in the reference's summation order (edges where the vertex is endpoint 0, then
 * endpoint 1, each in edge order: the two index_add_ calls of pt.py:633-634), so each row is one thread's
 * sequential sum and the result is BIT-IDENTICAL to go_spring_forces (tests/test_oracle_golden.py).  The KNN
 * is go_knn_midpoints (already parallel over the queries); combine + normalise with per-thread fp64 column
 * sums (not ATen's fp32 cascade: agrees with go_integrate_normalise to ~1e-7, checked in the same test). */
void go_pull_lists(const int32_t *edges, int64_t E, int64_t n, int64_t *rowptr /* n+1 */, int32_t *adj /* 2E */,
                   int8_t *sign /* 2E: +1 endpoint 0, -1 endpoint 1 */) {
    memset(rowptr, 0, sizeof(int64_t) * (size_t)(n + 1));
    for (int64_t e = 0; e < E; ++e) { rowptr[edges[2 * e] + 1]++; rowptr[edges[2 * e + 1] + 1]++; }
    for (int64_t i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    memcpy(cur, rowptr, sizeof(int64_t) * (size_t)n);
    for (int64_t e = 0; e < E; ++e) { const int64_t at = cur[edges[2 * e]]++; adj[at] = edges[2 * e + 1]; sign[at] = 1; }
    for (int64_t e = 0; e < E; ++e) { const int64_t at = cur[edges[2 * e + 1]]++; adj[at] = edges[2 * e]; sign[at] = -1; }
    free(cur);
}

void go_spring_forces_pull(const float *pos, int64_t n, int D, const int64_t *rowptr, const int32_t *adj,
                           const int8_t *sign, float L_min, float k_attr, float *F) {
    const float neg_k = -k_attr;
#pragma omp parallel
    {
        float *diff = (float *)malloc(sizeof(float) * (size_t)D);
#pragma omp for schedule(static)
        for (int64_t x = 0; x < n; ++x) {
            float *dst = F + (size_t)x * D;
            for (int d = 0; d < D; ++d) dst[d] = 0.0f;
            for (int64_t j = rowptr[x]; j < rowptr[x + 1]; ++j) {
                /* diff = p[v] - p[u] of the edge (u < v): the neighbour minus x when x is endpoint 0, x minus the
                 * neighbour when it is endpoint 1; F[u] += f, F[v] += -f */
                const float *px = pos + (size_t)x * D, *py = pos + (size_t)adj[j] * D;
                if (sign[j] > 0) for (int d = 0; d < D; ++d) diff[d] = py[d] - px[d];
                else for (int d = 0; d < D; ++d) diff[d] = px[d] - py[d];
                const float dist = go_norm2(diff, D) + 1e-6f;
                const float fm = neg_k * (dist - L_min);
                for (int d = 0; d < D; ++d) {
                    const float f = fm * (diff[d] / dist);
                    dst[d] = dst[d] + (sign[j] > 0 ? f : -f);
                }
            }
        }
        free(diff);
    }
}

/* The same exact search with the loops the other way round (bench.py's all-cores baseline; the per-query scan above
 * streams all E midpoints once per query, S * E * D * 4 bytes per iteration).  Here a thread takes a block of
 * GO_KNN_BLOCK consecutive edges, computes its midpoints once, coordinate-major (they stay in its L1) and runs ALL S
 * queries over the block, keeping per query its own K best (d2, id) pairs; the threads' lists are merged at the end.
 * Blocks are contiguous id ranges and a thread's blocks ascend, so "ties keep the smaller id" holds inside a list; the
 * merge orders by (d2, id).  Same result as go_knn_midpoints (tests/test_oracle_golden.py). */
#define GO_KNN_BLOCK 2048   /* a multiple of 16 */
int go_knn_midpoints_tiled(const float *pos, int D, const int32_t *edges, int64_t E, const int32_t *sampled, int64_t S, int k,
                           int32_t *knn_out) {
    const int K = k + 1;
    if ((int64_t)K > E) return GO_ERR_K_TOO_LARGE;
    float *q = (float *)malloc(sizeof(float) * (size_t)S * D);
    if (!q) return GO_ERR_NOMEM;
    for (int64_t r = 0; r < S; ++r)
        for (int d = 0; d < D; ++d)
            q[(size_t)r * D + d] = (pos[(size_t)edges[2 * (size_t)sampled[r]] * D + d] + pos[(size_t)edges[2 * (size_t)sampled[r] + 1] * D + d]) / 2.0f;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    float *bd_all = (float *)malloc(sizeof(float) * (size_t)nthreads * S * K);
    int32_t *bi_all = (int32_t *)malloc(sizeof(int32_t) * (size_t)nthreads * S * K);
    int *cnt_all = (int *)calloc((size_t)nthreads * S, sizeof(int));
    if (!bd_all || !bi_all || !cnt_all) { free(q); free(bd_all); free(bi_all); free(cnt_all); return GO_ERR_NOMEM; }
    const int64_t nblocks = (E + GO_KNN_BLOCK - 1) / GO_KNN_BLOCK;
#pragma omp parallel num_threads(nthreads)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        float *bd = bd_all + (size_t)tid * S * K;
        int32_t *bi = bi_all + (size_t)tid * S * K;
        int *cnt = cnt_all + (size_t)tid * S;
        float *mid = (float *)malloc(sizeof(float) * (size_t)GO_KNN_BLOCK * D);
        float d2[16];
#pragma omp for schedule(static)   /* contiguous, ascending block ranges per thread */
        for (int64_t b = 0; b < nblocks; ++b) {
            const int64_t e0 = b * GO_KNN_BLOCK, m = (E - e0 < GO_KNN_BLOCK) ? E - e0 : GO_KNN_BLOCK, mpad = (m + 15) & ~(int64_t)15;
            for (int64_t j = 0; j < m; ++j)   /* coordinate-major inside the block: the distance loop below is unit-stride */
                for (int d = 0; d < D; ++d)
                    mid[(size_t)d * GO_KNN_BLOCK + j] = (pos[(size_t)edges[2 * (e0 + j)] * D + d] + pos[(size_t)edges[2 * (e0 + j) + 1] * D + d]) / 2.0f;
            for (int64_t j = m; j < mpad; ++j)   /* padding lanes: far away in coordinate 0, never selected (d2 = inf) */
                for (int d = 0; d < D; ++d) mid[(size_t)d * GO_KNN_BLOCK + j] = d == 0 ? INFINITY : 0.0f;
            for (int64_t r = 0; r < S; ++r) {
                const float *qq = q + (size_t)r * D;
                float *rd = bd + (size_t)r * K;
                int32_t *ri = bi + (size_t)r * K;
                int c = cnt[r];
                float worst = c == K ? rd[K - 1] : INFINITY;
                __m256 vworst = _mm256_set1_ps(worst);
                for (int64_t j0 = 0; j0 < mpad; j0 += 16) {
                    /* go_d2's chain per lane: t = q_d - m_d (rounded), acc = fma(t, t, acc), coordinates in order */
                    __m256 a0 = _mm256_setzero_ps(), a1 = _mm256_setzero_ps();
                    for (int d = 0; d < D; ++d) {
                        const __m256 qd = _mm256_broadcast_ss(qq + d);
                        const float *md = mid + (size_t)d * GO_KNN_BLOCK + j0;
                        const __m256 t0 = _mm256_sub_ps(qd, _mm256_loadu_ps(md)), t1 = _mm256_sub_ps(qd, _mm256_loadu_ps(md + 8));
                        a0 = _mm256_fmadd_ps(t0, t0, a0);
                        a1 = _mm256_fmadd_ps(t1, t1, a1);
                    }
                    const int hit = _mm256_movemask_ps(_mm256_or_ps(_mm256_cmp_ps(a0, vworst, _CMP_LT_OQ), _mm256_cmp_ps(a1, vworst, _CMP_LT_OQ)));
                    if (!hit) continue;
                    _mm256_storeu_ps(d2, a0);
                    _mm256_storeu_ps(d2 + 8, a1);
                    for (int jj = 0; jj < 16; ++jj) {
                        const float v = d2[jj];
                        if (!(v < worst)) continue;   /* ties keep the smaller id (earlier edge) */
                        int t = c < K ? c : K - 1;
                        while (t > 0 && rd[t - 1] > v) { rd[t] = rd[t - 1]; ri[t] = ri[t - 1]; --t; }
                        rd[t] = v; ri[t] = (int32_t)(e0 + j0 + jj);
                        if (c < K) ++c;
                        worst = c == K ? rd[K - 1] : INFINITY;
                    }
                    vworst = _mm256_set1_ps(worst);
                }
                cnt[r] = c;
            }
        }
        free(mid);
    }
    /* merge: per query the K smallest (d2, id) over the threads' lists */
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < S; ++r) {
        int head[1024];
        int *hp = nthreads <= 1024 ? head : (int *)malloc(sizeof(int) * (size_t)nthreads);
        for (int t = 0; t < nthreads; ++t) hp[t] = 0;
        for (int c = 0; c < K; ++c) {
            int best = -1;
            for (int t = 0; t < nthreads; ++t) {
                if (hp[t] >= cnt_all[(size_t)t * S + r]) continue;
                if (best < 0) { best = t; continue; }
                const float dv = bd_all[((size_t)t * S + r) * K + hp[t]], db = bd_all[((size_t)best * S + r) * K + hp[best]];
                const int32_t iv = bi_all[((size_t)t * S + r) * K + hp[t]], ib = bi_all[((size_t)best * S + r) * K + hp[best]];
                if (dv < db || (dv == db && iv < ib)) best = t;
            }
            const int32_t id = bi_all[((size_t)best * S + r) * K + hp[best]];
            hp[best]++;
            if (c >= 1) knn_out[(size_t)r * k + (c - 1)] = id;   /* column 0 dropped (pt.py:421) */
        }
        if (hp != head) free(hp);
    }
    free(q); free(bd_all); free(bi_all); free(cnt_all);
    return GO_OK;
}

int go_step_omp(float *pos, int64_t n, int D, const int32_t *edges, int64_t E, const int64_t *rowptr,
                const int32_t *adj, const int8_t *sign, const int32_t *sampled, int64_t S, int k, float L_min,
                float k_attr, float k_inter) {
    float *Fs = (float *)malloc(sizeof(float) * (size_t)n * D);
    float *Fi = (float *)malloc(sizeof(float) * (size_t)n * D);
    int32_t *knn = (int32_t *)malloc(sizeof(int32_t) * (size_t)(S * k > 0 ? S * k : 1));
    if (!Fs || !Fi || !knn) { free(Fs); free(Fi); free(knn); return GO_ERR_NOMEM; }
    go_spring_forces_pull(pos, n, D, rowptr, adj, sign, L_min, k_attr, Fs);
    int err = go_knn_midpoints_tiled(pos, D, edges, E, sampled, S, k, knn);
    if (err == GO_OK) {
        go_intersection_forces(pos, n, D, edges, sampled, S, knn, k, k_inter, Fi, NULL);   /* O(S k): serial */
        double *sum = (double *)calloc((size_t)D, sizeof(double)), *sq = (double *)calloc((size_t)D, sizeof(double));
#pragma omp parallel
        {
            double *ls = (double *)calloc((size_t)2 * D, sizeof(double));
#pragma omp for schedule(static)
            for (int64_t i = 0; i < n; ++i)
                for (int d = 0; d < D; ++d) {
                    const size_t o = (size_t)i * D + d;
                    const float tot = Fs[o] + Fi[o];
                    const float nw = pos[o] + tot;
                    Fs[o] = nw;
                    ls[d] += (double)nw;
                    ls[D + d] += (double)nw * (double)nw;
                }
#pragma omp critical
            for (int d = 0; d < D; ++d) { sum[d] += ls[d]; sq[d] += ls[D + d]; }
            free(ls);
        }
        for (int d = 0; d < D; ++d) {
            const double m = sum[d] / (double)n;
            double var = (sq[d] - sum[d] * m) / (double)(n - 1);
            if (var < 0.0) var = 0.0;
            sum[d] = (double)(float)m;
            sq[d] = (double)((float)sqrt(var) + 1e-6f);
        }
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i)
            for (int d = 0; d < D; ++d) {
                const size_t o = (size_t)i * D + d;
                pos[o] = (Fs[o] - (float)sum[d]) / (float)sq[d];
            }
        free(sum); free(sq);
    }
    free(Fs); free(Fi); free(knn);
    return err;
}

/* bench.py tries a few team sizes for the all-cores leg: on a two-socket host fewer threads than cores can be faster */
void go_set_threads(int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
}

int go_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
