// Micro-benchmark: N random 16-byte row gathers from a table of T rows, with different cache policies.
// hipcc -O3 --offload-arch=gfx950 tools/micro/gather_bench.hip -o gpurun_out/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__device__ __forceinline__ float4 ld(const float4 *p) {
    float4 v;
    if (MODE == 0) v = *p;
    else if (MODE == 1) {
        asm volatile("global_load_dwordx4 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    } else if (MODE == 2) {
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    } else if (MODE == 3) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    } else if (MODE == 4) {
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    } else {
        asm volatile("global_load_dwordx4 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    }
    return v;
}

// each thread: C gathers per step (issued together for MODE 0; the asm forms wait per load, so they
// are compared at equal occupancy with C independent threads instead)
template <int MODE>
__global__ __launch_bounds__(256) void gather(const float4 *__restrict__ table, const int *__restrict__ idx, int64_t n,
                                              float4 *__restrict__ out) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    float4 acc = make_float4(0, 0, 0, 0);
    const int4 *ip = reinterpret_cast<const int4 *>(idx) + 2 * t;
    const int4 a = ip[0], b = ip[1];
    const int ids[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float4 v = table[ids[j]]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    } else {
        float4 v[8];
        // issue all eight, then wait once
#define LD(j, FLAGS) asm volatile("global_load_dwordx4 %0, %1, off " FLAGS : "=v"(v[j]) : "v"(table + ids[j]) : "memory")
#define LD8(FLAGS) LD(0, FLAGS); LD(1, FLAGS); LD(2, FLAGS); LD(3, FLAGS); LD(4, FLAGS); LD(5, FLAGS); LD(6, FLAGS); LD(7, FLAGS)
        if (MODE == 1) { LD8("nt"); }
        else if (MODE == 2) { LD8("sc0 sc1"); }
        else if (MODE == 3) { LD8("sc1"); }
        else if (MODE == 4) { LD8("sc0 sc1 nt"); }
        else { LD8("sc0"); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
    out[t] = acc;
}

int main(int argc, char **argv) {
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 1000000;   // table rows (16 B each)
    const int64_t n = argc > 2 ? atoll(argv[2]) : 1000000;      // gathering threads (8 gathers each)
    const int only = argc > 3 ? atoi(argv[3]) : -1;             // one cache policy only (counter passes: 0 = plain)
    std::vector<int> h((size_t)n * 8);
    std::mt19937_64 rng(1);
    for (auto &x : h) x = (int)(rng() % rows);
    float4 *table, *out; int *idx;
    CK(hipMalloc(&table, rows * 16)); CK(hipMalloc(&out, n * 16)); CK(hipMalloc(&idx, n * 32));
    CK(hipMemset(table, 0, rows * 16));
    CK(hipMemcpy(idx, h.data(), n * 32, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"plain", "nt", "sc0 sc1", "sc1", "sc0 sc1 nt", "sc0"};
    for (int mode = 0; mode < 6; ++mode) {
        if (only >= 0 && mode != only) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            const dim3 g((unsigned)((n + 255) / 256)), b(256);
            switch (mode) {
                case 0: gather<0><<<g, b>>>(table, idx, n, out); break;
                case 1: gather<1><<<g, b>>>(table, idx, n, out); break;
                case 2: gather<2><<<g, b>>>(table, idx, n, out); break;
                case 3: gather<3><<<g, b>>>(table, idx, n, out); break;
                case 4: gather<4><<<g, b>>>(table, idx, n, out); break;
                default: gather<5><<<g, b>>>(table, idx, n, out); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("rows=%lld gathers=%lld %-12s %8.1f us  %6.1f G gathers/s\n", (long long)rows, (long long)n * 8, names[mode],
               best * 1e3, n * 8 / (best * 1e-3) / 1e9);
    }
    return 0;
}
