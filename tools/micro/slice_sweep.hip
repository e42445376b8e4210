// Micro-benchmark: neighbour-row gathers of a pull-style sweep over a random 8-regular "graph",
// (a) plain: one thread per row, 8 gathers from the whole 16 MB table;
// (b) slice sweep: workgroups with blockIdx%8 == g own the rows of slice g (1/8 of the table) and walk
//     the neighbour lists slice by slice in cyclic order, so that at any time the workgroups of one
//     XCD gather from ONE 2 MB slice (L2-resident).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void plain(const float4 *__restrict__ pos, const int *__restrict__ adj, int n, int deg,
                                             float4 *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int j = 0; j < deg; ++j) {
        const float4 v = pos[adj[(int64_t)i * deg + j]];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    out[i] = acc;
}

template <int T /* tiles per workgroup */>
__global__ __launch_bounds__(256) void sweep(const float4 *__restrict__ pos, const int *__restrict__ adj, int n, int deg,
                                             int slice_rows, int G, float4 *__restrict__ out) {
    const int g = blockIdx.x % 8, local = blockIdx.x / 8;
    const int r0 = g * slice_rows;
    int cur[T], end[T], row[T];
    float4 acc[T];
#pragma unroll
    for (int k = 0; k < T; ++k) {
        const int i = r0 + (local + k * G) * 256 + threadIdx.x;
        const bool ok = i < r0 + slice_rows && i < n;
        row[k] = ok ? i : -1;
        cur[k] = ok ? i * deg : 0;
        end[k] = ok ? i * deg + deg : 0;
        acc[k] = make_float4(0, 0, 0, 0);
    }
    for (int t = 0; t < 8; ++t) {
        const int s = (g + t) & 7;
        const int lo = s * slice_rows, hi = lo + slice_rows;
        bool more = true;
        while (more) {
            more = false;
            int y[T];
#pragma unroll
            for (int k = 0; k < T; ++k) {
                y[k] = -1;
                if (cur[k] < end[k]) {
                    const int c = adj[cur[k]];
                    if (c >= lo && c < hi) { y[k] = c; ++cur[k]; more = true; }
                }
            }
            float4 v[T];
#pragma unroll
            for (int k = 0; k < T; ++k) v[k] = y[k] >= 0 ? pos[y[k]] : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < T; ++k) { acc[k].x += v[k].x; acc[k].y += v[k].y; acc[k].z += v[k].z; acc[k].w += v[k].w; }
            more = __any(more);
        }
    }
#pragma unroll
    for (int k = 0; k < T; ++k) if (row[k] >= 0) out[row[k]] = acc[k];
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1000000, deg = 8;
    const int slice_rows = (n + 7) / 8;
    std::vector<int> adj((size_t)n * deg), adj2((size_t)n * deg);
    std::mt19937_64 rng(1);
    for (auto &x : adj) x = (int)(rng() % n);
    for (int i = 0; i < n; ++i) {  // cyclic slice order starting at the row's own slice
        const int r = i / slice_rows;
        std::vector<int> l(adj.begin() + (size_t)i * deg, adj.begin() + (size_t)(i + 1) * deg);
        std::sort(l.begin(), l.end(), [&](int a, int b) {
            const int sa = (a / slice_rows - r + 8) % 8, sb = (b / slice_rows - r + 8) % 8;
            return sa != sb ? sa < sb : a < b;
        });
        std::copy(l.begin(), l.end(), adj2.begin() + (size_t)i * deg);
    }
    float4 *pos, *out; int *dadj, *dadj2;
    CK(hipMalloc(&pos, (size_t)n * 16)); CK(hipMalloc(&out, (size_t)n * 16));
    CK(hipMalloc(&dadj, adj.size() * 4)); CK(hipMalloc(&dadj2, adj.size() * 4));
    std::vector<float> hp((size_t)n * 4, 1.0f);
    CK(hipMemcpy(pos, hp.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(dadj, adj.data(), adj.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dadj2, adj2.data(), adj.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        std::vector<float> ho((size_t)n * 4);
        CK(hipMemcpy(ho.data(), out, (size_t)n * 16, hipMemcpyDeviceToHost));
        double sum = 0; for (size_t i = 0; i < ho.size(); i += 4) sum += ho[i];
        printf("n=%d %-28s %8.1f us   checksum %.0f\n", n, name, best * 1e3, sum);
        CK(hipMemset(out, 0, (size_t)n * 16));
    };
    timeit("plain", [&] { plain<<<dim3((n + 255) / 256), dim3(256)>>>(pos, dadj, n, deg, out); });
    timeit("plain (sorted lists)", [&] { plain<<<dim3((n + 255) / 256), dim3(256)>>>(pos, dadj2, n, deg, out); });
    const int tiles = (slice_rows + 255) / 256;
    for (int T : {1, 2, 4}) {
        const int G = (tiles + T - 1) / T;
        char name[64]; snprintf(name, sizeof name, "sweep T=%d (G=%d per XCD)", T, G);
        if (T == 1) timeit(name, [&] { sweep<1><<<dim3(8 * G), dim3(256)>>>(pos, dadj2, n, deg, slice_rows, G, out); });
        if (T == 2) timeit(name, [&] { sweep<2><<<dim3(8 * G), dim3(256)>>>(pos, dadj2, n, deg, slice_rows, G, out); });
        if (T == 4) timeit(name, [&] { sweep<4><<<dim3(8 * G), dim3(256)>>>(pos, dadj2, n, deg, slice_rows, G, out); });
    }
    return 0;
}
