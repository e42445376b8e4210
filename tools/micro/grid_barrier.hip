// Micro-benchmark (round 2): what one grid-wide barrier costs inside a resident (cooperatively launched) kernel on
// MI355X, against the launch boundary it would replace.  Decides whether a persistent kernel for small graphs
// (five dependent launches per iteration, 60+ us for ~15 us of work) is worth building.
//
//   barrier   G workgroups x 256 threads; per round every workgroup writes `bytes` of its own slot, passes the
//             barrier (thread 0: agent-scope release add on one counter, relaxed spin, acquire fence), then reads
//             the slot of workgroup (b + 97) % G and checks it holds this round's value (so the measured barrier
//             really publishes across XCDs).
//   launches  the same write / check as two kernels per round on one stream; "graph": that chain captured and replayed.
//   flag      the protocol of the thresholds inside the fused launch (fused.hip): the FIRST P workgroups produce (write-through
//             sc1 stores, s_waitcnt vmcnt(0), one add on a counter) and leave; every other workgroup first reads
//             the producers' slots (stale copies now sit in its L1 / L2), does ~5 us of other work, then polls the
//             counter with relaxed agent-scope loads -- NO acquire fence, no cache invalidate -- and reads the slots
//             with agent-scope atomic loads (sc1).  "stale" counts values that were not this round's.
// Usage: grid_barrier [rounds]
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned *ctr, unsigned target, unsigned *fail) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = false; break; }   // never hang the box: give up after ~0.1 s
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);   // agent scope by default for device code
        if (!ok) atomicAdd(fail, 1u);
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void barrier_kernel(unsigned *ctr, unsigned *fail, float *buf, int words, int rounds,
                                                      unsigned *bad) {
    const int G = gridDim.x, b = blockIdx.x;
    unsigned wrong = 0;
    for (int r = 0; r < rounds; ++r) {
        float *mine = buf + (size_t)b * words;
        for (int i = threadIdx.x; i < words; i += 256) mine[i] = (float)(r + 1);
        if (!grid_barrier(ctr, (unsigned)(2 * r + 1) * G, fail)) return;
        const float *other = buf + (size_t)((b + 97) % G) * words;
        for (int i = threadIdx.x; i < words; i += 256) wrong += (other[i] != (float)(r + 1));
        if (!grid_barrier(ctr, (unsigned)(2 * r + 2) * G, fail)) return;
    }
    if (wrong) atomicAdd(bad, wrong);
}

__global__ __launch_bounds__(256) void flag_kernel(unsigned *ctr, unsigned target, unsigned *fail, float *buf, int words, int P,
                                                   int round, unsigned *bad, float *sink) {
    const int b = blockIdx.x;
    if (b < P) {   // producer
        unsigned *mine = reinterpret_cast<unsigned *>(buf + (size_t)b * words);
        for (int i = threadIdx.x; i < words; i += 256)
            __hip_atomic_store(mine + i, __float_as_uint((float)(round + 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const float *other = buf + (size_t)(b % P) * words;
    float acc = 0.f;
    for (int i = threadIdx.x; i < words; i += 256) acc += other[i];   // plain read: last round's values get cached here
    for (int i = threadIdx.x; i < 1024; i += 256) sink[1024 + (size_t)b * 1024 + i] = acc;   // and this XCD's L2 gets dirty lines
    for (int k = 0; k < 400; ++k) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);   // stand-in for the spring phase
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 20)) { atomicAdd(fail, 1u); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    __syncthreads();
    unsigned wrong = 0;
    for (int i = threadIdx.x; i < words; i += 256) {
        const unsigned u = __hip_atomic_load(reinterpret_cast<const unsigned *>(other) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wrong += (__uint_as_float(u) != (float)(round + 1));
    }
    if (wrong) atomicAdd(bad, wrong);
    if (acc == 12345.f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void write_kernel(float *buf, int words, int r) {
    float *mine = buf + (size_t)blockIdx.x * words;
    for (int i = threadIdx.x; i < words; i += 256) mine[i] = (float)(r + 1);
}
__global__ __launch_bounds__(256) void check_kernel(const float *buf, int words, int r, unsigned *bad) {
    const float *other = buf + (size_t)((blockIdx.x + 97) % gridDim.x) * words;
    unsigned wrong = 0;
    for (int i = threadIdx.x; i < words; i += 256) wrong += (other[i] != (float)(r + 1));
    if (wrong) atomicAdd(bad, wrong);
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("%s: %d CUs\n", prop.name, prop.multiProcessorCount);
    unsigned *d_ctr, *d_fail, *d_bad;
    float *d_buf;
    CK(hipMalloc(&d_ctr, 4)); CK(hipMalloc(&d_fail, 4)); CK(hipMalloc(&d_bad, 4));
    CK(hipMalloc(&d_buf, 64u << 20));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, barrier_kernel, 256, 0));
    printf("barrier_kernel: %d workgroups per CU resident\n", occ);
    for (int G : {256, 512, 1024}) {
        if (G > occ * prop.multiProcessorCount) continue;
        for (int words : {64, 1024, 4096}) {     // 256 B, 4 KB, 16 KB per workgroup and phase
            CK(hipMemsetAsync(d_ctr, 0, 4, st)); CK(hipMemsetAsync(d_fail, 0, 4, st)); CK(hipMemsetAsync(d_bad, 0, 4, st));
            float ms[2];
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemsetAsync(d_ctr, 0, 4, st));
                int w = words, rr = rounds;
                void *args[] = {&d_ctr, &d_fail, &d_buf, &w, &rr, &d_bad};
                CK(hipEventRecord(e0, st));
                CK(hipLaunchCooperativeKernel(reinterpret_cast<void *>(barrier_kernel), dim3(G), dim3(256), args, 0, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                CK(hipEventElapsedTime(&ms[rep], e0, e1));
            }
            unsigned fail = 0, bad = 0;
            CK(hipMemcpy(&fail, d_fail, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
            // the same as launches
            CK(hipMemsetAsync(d_bad, 0, 4, st));
            float msl = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < rounds; ++r) {
                    write_kernel<<<G, 256, 0, st>>>(d_buf, words, r);
                    check_kernel<<<G, 256, 0, st>>>(d_buf, words, r, d_bad);
                }
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                CK(hipEventElapsedTime(&msl, e0, e1));
            }
            unsigned bad2 = 0;
            CK(hipMemcpy(&bad2, d_bad, 4, hipMemcpyDeviceToHost));
            printf("G=%4d  %5d B/wg/phase: barrier %.2f us per phase (timeouts %u, stale reads %u)   launch %.2f us per phase (stale %u)\n",
                   G, words * 4, 1e3 * ms[1] / (2 * rounds), fail, bad, 1e3 * msl / (2 * rounds), bad2);
        }
    }
    // ---- the same chain of dependent launches replayed from a hipGraph
    for (int G : {256, 1024}) {
        const int words = 1024;
        hipGraph_t graph;
        hipGraphExec_t exec;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int r = 0; r < rounds; ++r) {
            write_kernel<<<G, 256, 0, st>>>(d_buf, words, r);
            check_kernel<<<G, 256, 0, st>>>(d_buf, words, r, d_bad);
        }
        CK(hipStreamEndCapture(st, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(exec, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("graph: G=%d, %d launches replayed: %.2f us per launch\n", G, 2 * rounds, 1e3 * ms / (2 * rounds));
        CK(hipGraphExecDestroy(exec));
        CK(hipGraphDestroy(graph));
    }
    // ---- producer / consumer flag inside one launch
    for (int G : {1024, 8192}) {
        const int P = 64, words = 1024;
        CK(hipMemsetAsync(d_ctr, 0, 4, st)); CK(hipMemsetAsync(d_fail, 0, 4, st)); CK(hipMemsetAsync(d_bad, 0, 4, st));
        CK(hipMemsetAsync(d_buf, 0, (size_t)P * words * 4, st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < rounds; ++r)
            flag_kernel<<<G + P, 256, 0, st>>>(d_ctr, (unsigned)(r + 1) * P, d_fail, d_buf, words, P, r, d_bad, d_buf + (4u << 20) / 4);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned fail = 0, bad = 0;
        CK(hipMemcpy(&fail, d_fail, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
        printf("flag: %d consumers + %d producers, %d rounds: %.2f us per launch, timeouts %u, stale values %u\n", G, P, rounds,
               1e3 * ms / rounds, fail, bad);
    }
    return 0;
}
