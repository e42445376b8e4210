// Micro-benchmark for VERDICT r3 item 3: "evaluate each edge once".
//
// Today the spring phase PULLS: every vertex walks its list of 8 neighbours, gathers their rows (2E random 16-byte
// gathers per iteration) and evaluates 2E sqrt / division chains.  The compute-once form evaluates an undirected edge
// ONCE, at the endpoint that owns it (E random gathers, E chains), keeps +f in its own slot and stores -f into the
// partner's pull-list SLOT (E random 16-byte stores); a streaming pass then adds each vertex's 8 slots -- one 128-byte
// line -- in the reference's order and writes Fs and new0.  Same sums bit for bit (negation is exact); the question is
// whether E gathers + E scattered stores + the streamed sum beat 2E gathers on this memory system.
//
//   pull        8 gathers + 8 terms per vertex -> Fs, new0                      (phase A of the fused kernel, without the scan)
//   pull_mem    the same without the arithmetic (sum of the gathered rows)      (its memory side)
//   once        4 gathers + 4 terms per vertex -> 4 own slots (one 64-byte store) + 4 scattered 16-byte stores
//   once_mem    the same without the arithmetic
//   slotsum     8 slots per vertex (128 contiguous bytes) added in order -> Fs, new0
//
// Graph: 4 random permutations p_j; vertex x owns the edges (x, p_j(x)), j < 4, and is the partner of the edges
// (q_j(x), x), q_j = p_j^-1: 8-regular, no locality (a random-regular graph's worst case, like bench.py's rr1m / rr16m).
// Slot s of vertex x: s < 4 the owned edge j = s, s >= 4 the incoming edge j = s - 4.
//
// hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -I graphem-rapids_amd/csrc tools/micro/edge_once.hip -o gpurun_out/edge_once
// usage: edge_once [n = 1000000]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#include "common.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// the spring term of common.h's spring_pull for one neighbour row
template <bool ARITH>
__device__ __forceinline__ void term(const float *px, const float *py, float L_min, float neg_k, float *t) {
    if (ARITH) {
        float diff[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) diff[d] = py[d] - px[d];
        const float dist = gh_sqrt_ieee(gh_sumsq<3>(diff)) + 1e-6f;
        const float fm = neg_k * (dist - L_min);
        float quot[3];
        gh_div_by<3>(diff, dist, quot);
#pragma unroll
        for (int d = 0; d < 3; ++d) t[d] = fm * quot[d];
    } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) t[d] = py[d];
    }
    t[3] = 0.0f;
}

// (round 4, second half) two more shapes of the same pull: the neighbour rows through non-temporal loads (global_load ... nt:
// does a gather that is not going to hit again cost the fabric a whole 128-byte line?), and rows without their pad column
// (12-byte stride: a 12 MB table, 10.7 rows per line instead of 8)
typedef float gh_v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void pull_variant_kernel(const float *__restrict__ pos, const float *__restrict__ pos12,
                                                           const int *__restrict__ adj8, int64_t n, float *__restrict__ Fs,
                                                           float *__restrict__ new0) {
    const int64_t x = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (x >= n) return;
    float px[4], F[4] = {0.f, 0.f, 0.f, 0.f};
    gh_load_row<4>(pos, x, px);
    const int4 a = reinterpret_cast<const int4 *>(adj8)[2 * x], b = reinterpret_cast<const int4 *>(adj8)[2 * x + 1];
    const int ys[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float py[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (MODE == 1) {
            const gh_v4f r = __builtin_nontemporal_load(reinterpret_cast<const gh_v4f *>(pos) + ys[j]);
            py[j][0] = r.x; py[j][1] = r.y; py[j][2] = r.z; py[j][3] = r.w;
        } else {
            const float *r = pos12 + (int64_t)ys[j] * 3;
            py[j][0] = r[0]; py[j][1] = r[1]; py[j][2] = r[2]; py[j][3] = 0.0f;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t[4];
        term<true>(px, py[j], 1.0f, -0.2f, t);
#pragma unroll
        for (int d = 0; d < 3; ++d) F[d] = F[d] + t[d];
    }
    float nw[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) nw[d] = px[d] + F[d];
    gh_store_row<4>(Fs, x, F);
    gh_store_row<4>(new0, x, nw);
}

template <bool ARITH>
__global__ __launch_bounds__(256) void pull_kernel(const float *__restrict__ pos, const int *__restrict__ adj8, int64_t n,
                                                   float *__restrict__ Fs, float *__restrict__ new0) {
    const int64_t x = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (x >= n) return;
    float px[4], F[4] = {0.f, 0.f, 0.f, 0.f};
    gh_load_row<4>(pos, x, px);
    const int4 a = reinterpret_cast<const int4 *>(adj8)[2 * x], b = reinterpret_cast<const int4 *>(adj8)[2 * x + 1];
    const int ys[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float py[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) gh_load_row<4>(pos, ys[j], py[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t[4];
        term<ARITH>(px, py[j], 1.0f, -0.2f, t);
#pragma unroll
        for (int d = 0; d < 3; ++d) F[d] = F[d] + t[d];
    }
    float nw[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) nw[d] = px[d] + F[d];
    gh_store_row<4>(Fs, x, F);
    gh_store_row<4>(new0, x, nw);
}

template <bool ARITH>
__global__ __launch_bounds__(256) void once_kernel(const float *__restrict__ pos, const int *__restrict__ adj4, int64_t n,
                                                   float *__restrict__ slots) {
    const int64_t x = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (x >= n) return;
    float px[4];
    gh_load_row<4>(pos, x, px);
    const int4 a = reinterpret_cast<const int4 *>(adj4)[x];
    const int ys[4] = {a.x, a.y, a.z, a.w};
    float py[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) gh_load_row<4>(pos, ys[j], py[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float t[4], mt[4];
        term<ARITH>(px, py[j], 1.0f, -0.2f, t);
#pragma unroll
        for (int d = 0; d < 4; ++d) mt[d] = -t[d];
        gh_store_row<4>(slots, x * 8 + j, t);                       // own slot j (the four of them: one 64-byte run)
        gh_store_row<4>(slots, (int64_t)ys[j] * 8 + 4 + j, mt);     // the partner's slot of this edge: a random 16-byte store
    }
}

__global__ __launch_bounds__(256) void slotsum_kernel(const float *__restrict__ pos, const float *__restrict__ slots, int64_t n,
                                                      float *__restrict__ Fs, float *__restrict__ new0) {
    const int64_t x = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (x >= n) return;
    float px[4], F[4] = {0.f, 0.f, 0.f, 0.f}, s[8][4];
    gh_load_row<4>(pos, x, px);
#pragma unroll
    for (int j = 0; j < 8; ++j) gh_load_row<4>(slots, x * 8 + j, s[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int d = 0; d < 3; ++d) F[d] = F[d] + s[j][d];
    float nw[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) nw[d] = px[d] + F[d];
    gh_store_row<4>(Fs, x, F);
    gh_store_row<4>(new0, x, nw);
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
    std::mt19937_64 rng(1);
    std::vector<int> adj8((size_t)n * 8), adj4((size_t)n * 4), perm((size_t)n);
    for (int j = 0; j < 4; ++j) {
        std::iota(perm.begin(), perm.end(), 0);
        std::shuffle(perm.begin(), perm.end(), rng);
        for (int64_t x = 0; x < n; ++x) {
            adj4[(size_t)x * 4 + j] = perm[(size_t)x];
            adj8[(size_t)x * 8 + j] = perm[(size_t)x];
            adj8[(size_t)perm[(size_t)x] * 8 + 4 + j] = (int)x;
        }
    }
    std::vector<float> hp((size_t)n * 4);
    std::normal_distribution<float> g(0.f, 1.f);
    for (int64_t i = 0; i < n; ++i) { for (int d = 0; d < 3; ++d) hp[(size_t)i * 4 + d] = g(rng); hp[(size_t)i * 4 + 3] = 0.f; }
    float *pos, *Fs, *new0, *Fs2, *new02, *slots;
    int *d8, *d4;
    CK(hipMalloc(&pos, n * 16)); CK(hipMalloc(&Fs, n * 16)); CK(hipMalloc(&new0, n * 16)); CK(hipMalloc(&Fs2, n * 16)); CK(hipMalloc(&new02, n * 16));
    CK(hipMalloc(&slots, n * 128)); CK(hipMalloc(&d8, n * 32)); CK(hipMalloc(&d4, n * 16));
    CK(hipMemcpy(pos, hp.data(), n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(d8, adj8.data(), n * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(d4, adj4.data(), n * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    auto timeit = [&](const char *name, auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 8; ++rep) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        printf("%-10s %9.1f us\n", name, best * 1e3f);
        return best * 1e3f;
    };
    printf("n = %lld vertices, E = %lld edges, position table %.0f MB, slot array %.0f MB\n", (long long)n, (long long)(4 * n), n * 16 / 1e6, n * 128 / 1e6);
    const float t_pull = timeit("pull", [&] { pull_kernel<true><<<grid, blk>>>(pos, d8, n, Fs, new0); });
    const float t_pull_mem = timeit("pull_mem", [&] { pull_kernel<false><<<grid, blk>>>(pos, d8, n, Fs, new0); });
    {
        std::vector<float> h12((size_t)n * 3);
        for (int64_t i = 0; i < n; ++i) for (int d = 0; d < 3; ++d) h12[(size_t)i * 3 + d] = hp[(size_t)i * 4 + d];
        float *pos12;
        CK(hipMalloc(&pos12, n * 12));
        CK(hipMemcpy(pos12, h12.data(), n * 12, hipMemcpyHostToDevice));
        timeit("pull_nt", [&] { pull_variant_kernel<1><<<grid, blk>>>(pos, pos12, d8, n, Fs2, new02); });
        timeit("pull_12B", [&] { pull_variant_kernel<2><<<grid, blk>>>(pos, pos12, d8, n, Fs2, new02); });
        std::vector<float> a((size_t)n * 4), b((size_t)n * 4);
        pull_kernel<true><<<grid, blk>>>(pos, d8, n, Fs, new0);
        CK(hipMemcpy(a.data(), Fs, n * 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), Fs2, n * 16, hipMemcpyDeviceToHost));
        int64_t bad = 0;
        for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i] ? 1 : 0;
        printf("12-byte rows: forces identical: %s\n", bad ? "NO" : "yes");
        CK(hipFree(pos12));
    }
    pull_kernel<true><<<grid, blk>>>(pos, d8, n, Fs, new0);
    const float t_once = timeit("once", [&] { once_kernel<true><<<grid, blk>>>(pos, d4, n, slots); });
    const float t_once_mem = timeit("once_mem", [&] { once_kernel<false><<<grid, blk>>>(pos, d4, n, slots); });
    once_kernel<true><<<grid, blk>>>(pos, d4, n, slots);
    const float t_sum = timeit("slotsum", [&] { slotsum_kernel<<<grid, blk>>>(pos, slots, n, Fs2, new02); });
    // the two forms must give the same forces bit for bit
    std::vector<float> a((size_t)n * 4), b((size_t)n * 4);
    CK(hipMemcpy(a.data(), Fs, n * 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), Fs2, n * 16, hipMemcpyDeviceToHost));
    int64_t bad = 0;
    for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i] ? 1 : 0;
    printf("forces identical: %s (%lld differing floats)\n", bad ? "NO" : "yes", (long long)bad);
    printf("pull %.1f us | once + slotsum %.1f us (%.1f + %.1f) | memory sides: pull %.1f, once %.1f\n", t_pull, t_once + t_sum, t_once, t_sum,
           t_pull_mem, t_once_mem);
    printf("rates: pull %.1f G gathers/s; once %.1f G (gathers + scattered stores)/s\n", 8.0 * n / t_pull / 1e3, 8.0 * n / t_once / 1e3);
    return 0;
}
