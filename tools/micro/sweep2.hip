// Micro-benchmark (round 2): time-sliced neighbour gathers.  Workgroups with blockIdx%8 == g (one XCD under
// round-robin placement) own the rows of slice g; a thread owns T rows and walks their neighbour lists -- sorted
// beforehand by slice in the cyclic order g, g+1, ... with the per-slice counts packed in one word -- window by
// window, so that at any time the resident workgroups of an XCD gather from ONE 1/8 slice of the table (2 MB at
// 1M rows: L2-resident).  No dependent index->slice test in the loop (slice_sweep.hip had one), all rows resident.
//   plain      : one thread per row, 8 gathers from the whole table
//   sweep2<T>  : as above, sums in arrival order (gather rate only)
//   sweep2o<T> : the same with the rows kept in registers and summed in LIST order at the end (what bit-exact spring
//                forces need): slot selection by v_cndmask chains
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void plain(const float4 *__restrict__ pos, const int *__restrict__ adj, int n,
                                             float4 *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = pos[adj[(int64_t)i * 8 + j]];
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    out[i] = acc;
}

// The fused kernel's whole memory side without any arithmetic: own row, 8 neighbour gathers through a pull list with
// ownership flags, two 16-byte rows written (Fs, new0), row pointer + first-edge reads, one 16-byte midpoint per owned
// edge into LDS.  The fair ceiling for spring_scan's phase A.
__global__ __launch_bounds__(256) void plain_full(const float4 *__restrict__ pos, const int *__restrict__ rowptr,
                                                  const int *__restrict__ first_edge, const int *__restrict__ adj, int n,
                                                  float4 *__restrict__ out1, float4 *__restrict__ out2) {
    __shared__ float4 tile[1024];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 px = pos[i];
    const int beg = rowptr[i], fe = first_edge[i];
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = pos[adj[beg + j] & 0x7FFFFFFF];
    float4 acc = make_float4(0, 0, 0, 0);
    int slot = (fe & 255) * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc.x += v[j].x - px.x; acc.y += v[j].y - px.y; acc.z += v[j].z - px.z;
        if (j & 1) tile[(slot++) & 1023] = make_float4(0.5f * (v[j].x + px.x), 0.5f * (v[j].y + px.y), 0.5f * (v[j].z + px.z), 0.f);
    }
    out1[i] = acc;
    out2[i] = make_float4(px.x + acc.x, px.y + acc.y, px.z + acc.z, tile[threadIdx.x].x);
}

// rows of XCD g: slice g; thread (local block b, k) -> row r0 + (b + k*G)*256 + tid
template <int T, bool ORDERED>
__global__ __launch_bounds__(256) void sweep2(const float4 *__restrict__ pos, const int *__restrict__ adj,
                                              const unsigned *__restrict__ cnt, const unsigned *__restrict__ perm, int n,
                                              int slice_rows, int G, float4 *__restrict__ out) {
    const int g = blockIdx.x % 8, local = blockIdx.x / 8;
    const int r0 = g * slice_rows;
    int row[T], cur[T];
    unsigned c[T];
    float4 acc[T];
    float rx[T][8], ry[T][8], rz[T][8];
#pragma unroll
    for (int k = 0; k < T; ++k) {
        const int i = r0 + (local + k * G) * 256 + threadIdx.x;
        const bool ok = i < r0 + slice_rows && i < n;
        row[k] = ok ? i : -1;
        cur[k] = ok ? i * 8 : 0;
        c[k] = ok ? cnt[i] : 0u;
        acc[k] = make_float4(0, 0, 0, 0);
        if (ORDERED) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { rx[k][j] = 0; ry[k][j] = 0; rz[k][j] = 0; }
        }
    }
    int slot[T];
#pragma unroll
    for (int k = 0; k < T; ++k) slot[k] = 0;
    for (int t = 0; t < 8; ++t) {
        int left[T];
        int mx = 0;
#pragma unroll
        for (int k = 0; k < T; ++k) { left[k] = (c[k] >> (4 * t)) & 15u; mx = max(mx, left[k]); }
        // wave-uniform trip count
        mx = __reduce_max_sync(0xFFFFFFFFFFFFFFFFull, mx);
        for (int j = 0; j < mx; ++j) {
            float4 v[T];
#pragma unroll
            for (int k = 0; k < T; ++k) {
                const bool on = j < left[k];
                v[k] = on ? pos[adj[cur[k]]] : make_float4(0, 0, 0, 0);
                if (on) ++cur[k];
            }
#pragma unroll
            for (int k = 0; k < T; ++k) {
                if (ORDERED) {
                    const bool on = j < left[k];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const bool here = on && slot[k] == q;
                        rx[k][q] = here ? v[k].x : rx[k][q];
                        ry[k][q] = here ? v[k].y : ry[k][q];
                        rz[k][q] = here ? v[k].z : rz[k][q];
                    }
                    if (on) ++slot[k];
                } else {
                    acc[k].x += v[k].x; acc[k].y += v[k].y; acc[k].z += v[k].z; acc[k].w += v[k].w;
                }
            }
        }
    }
    if (ORDERED) {
        // rows sit in slice order in slots 0..7; list position j lives in slot perm[j] (3 bits each)
#pragma unroll
        for (int k = 0; k < T; ++k) {
            const unsigned p = row[k] >= 0 ? perm[row[k]] : 0u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int sl = (p >> (3 * j)) & 7;
                float x = rx[k][0], y = ry[k][0], z = rz[k][0];
#pragma unroll
                for (int q = 1; q < 8; ++q) { x = sl == q ? rx[k][q] : x; y = sl == q ? ry[k][q] : y; z = sl == q ? rz[k][q] : z; }
                acc[k].x += x; acc[k].y += y; acc[k].z += z;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < T; ++k) if (row[k] >= 0) out[row[k]] = acc[k];
}

// sweep2 with one row per thread and the gathered rows staged through LDS at their LIST position (per-lane slot
// address), summed in list order afterwards: the reference's summation order at the price of 128 bytes of LDS per thread.
__global__ __launch_bounds__(256) void sweep2_lds(const float4 *__restrict__ pos, const int *__restrict__ adj,
                                                  const unsigned *__restrict__ cnt, const unsigned *__restrict__ perm, int n,
                                                  int slice_rows, int G, float4 *__restrict__ out) {
    __shared__ float4 rows[8][256];
    const int g = blockIdx.x % 8, local = blockIdx.x / 8;
    const int r0 = g * slice_rows;
    const int i = r0 + local * 256 + threadIdx.x;
    const bool ok = i < r0 + slice_rows && i < n;
    int cur = ok ? i * 8 : 0;
    const unsigned c = ok ? cnt[i] : 0u;
    const unsigned p = ok ? perm[i] : 0u;
    // inverse of perm: slot s (arrival order) holds list position inv[s]
    unsigned inv = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) inv |= (unsigned)j << (3 * ((p >> (3 * j)) & 7));
    int slot = 0;
    for (int t = 0; t < 8; ++t) {
        const int left = (c >> (4 * t)) & 15u;
        const int mx = __reduce_max_sync(0xFFFFFFFFFFFFFFFFull, left);
        for (int j = 0; j < mx; ++j) {
            if (j < left) {
                const float4 v = pos[adj[cur]];
                ++cur;
                rows[(inv >> (3 * slot)) & 7][threadIdx.x] = v;
                ++slot;
            }
        }
    }
    float4 acc = make_float4(0, 0, 0, 0);
    if (ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float4 v = rows[j][threadIdx.x]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        out[i] = acc;
    }
}

// The fused kernel's phase-A geometry: a 256-thread workgroup holds only 128 rows (512 owned edges per tile, 4 per row),
// threads 128..255 idle in this phase; own row, two outputs, midpoints to LDS.  Variant 0: all 8 gathers at once in list
// order (today's kernel); variant 1: slice-sorted list walked window by window, XCD g owning slice g, sums in arrival
// order (what a relaxed summation order would allow).
template <int SWEEP>
__global__ __launch_bounds__(256) void phase_a_like(const float4 *__restrict__ pos, const int *__restrict__ adj,
                                                    const unsigned *__restrict__ cnt, int n, int slice_rows, int G,
                                                    float4 *__restrict__ out1, float4 *__restrict__ out2) {
    __shared__ float4 tile[512];
    const int g = blockIdx.x % 8, local = blockIdx.x / 8;
    const int i = SWEEP ? g * slice_rows + local * 128 + (int)threadIdx.x : (int)blockIdx.x * 128 + (int)threadIdx.x;
    const bool ok = threadIdx.x < 128 && i < n && (!SWEEP || i < (g + 1) * slice_rows);
    if (!ok) return;
    const float4 px = pos[i];
    float4 acc = make_float4(0, 0, 0, 0);
    int slot = threadIdx.x * 4;
    if (SWEEP) {
        const unsigned c = cnt[i];
        int cur = i * 8, seen = 0;
        for (int t = 0; t < 8; ++t) {
            const int left = (c >> (4 * t)) & 15u;
            for (int j = 0; j < left; ++j) {
                const float4 v = pos[adj[cur++]];
                acc.x += v.x - px.x; acc.y += v.y - px.y; acc.z += v.z - px.z;
                if (seen++ & 1) tile[(slot++) & 511] = make_float4(0.5f * (v.x + px.x), 0.5f * (v.y + px.y), 0.5f * (v.z + px.z), 0.f);
            }
        }
    } else {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = pos[adj[i * 8 + j]];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc.x += v[j].x - px.x; acc.y += v[j].y - px.y; acc.z += v[j].z - px.z;
            if (j & 1) tile[(slot++) & 511] = make_float4(0.5f * (v[j].x + px.x), 0.5f * (v[j].y + px.y), 0.5f * (v[j].z + px.z), 0.f);
        }
    }
    out1[i] = acc;
    out2[i] = make_float4(px.x + acc.x, px.y + acc.y, px.z + acc.z, tile[threadIdx.x].x);
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1000000, deg = 8;
    const int slice_rows = (n + 7) / 8;
    std::vector<int> adj((size_t)n * deg), adj2((size_t)n * deg);
    std::vector<unsigned> cnt((size_t)n), perm((size_t)n);
    std::mt19937_64 rng(1);
    for (auto &x : adj) x = (int)(rng() % n);
    for (int i = 0; i < n; ++i) {  // cyclic slice order starting at the row's own slice
        const int r = i / slice_rows;
        std::vector<int> idx(deg);
        for (int j = 0; j < deg; ++j) idx[j] = j;
        auto key = [&](int j) { return (adj[(size_t)i * deg + j] / slice_rows - r + 8) % 8; };
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key(a) < key(b); });
        unsigned c = 0, p = 0;
        for (int s = 0; s < deg; ++s) {
            adj2[(size_t)i * deg + s] = adj[(size_t)i * deg + idx[s]];
            c += 1u << (4 * key(idx[s]));
            p |= (unsigned)s << (3 * idx[s]);   // list position idx[s] sits in slot s
        }
        cnt[i] = c; perm[i] = p;
    }
    float4 *pos, *out; int *dadj, *dadj2; unsigned *dcnt, *dperm;
    CK(hipMalloc(&pos, (size_t)n * 16)); CK(hipMalloc(&out, (size_t)n * 16));
    CK(hipMalloc(&dadj, adj.size() * 4)); CK(hipMalloc(&dadj2, adj.size() * 4));
    CK(hipMalloc(&dcnt, (size_t)n * 4)); CK(hipMalloc(&dperm, (size_t)n * 4));
    std::vector<float> hp((size_t)n * 4);
    for (size_t i = 0; i < hp.size(); ++i) hp[i] = (float)((i * 2654435761u) % 1000) * 0.001f;
    CK(hipMemcpy(pos, hp.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(dadj, adj.data(), adj.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dadj2, adj2.data(), adj.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcnt, cnt.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dperm, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 8; ++rep) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 1 && ms < best) best = ms;
        }
        std::vector<float> ho((size_t)n * 4);
        CK(hipMemcpy(ho.data(), out, (size_t)n * 16, hipMemcpyDeviceToHost));
        double sum = 0; for (size_t i = 0; i < ho.size(); i += 4) sum += ho[i];
        printf("n=%d %-34s %8.1f us   checksum %.3f\n", n, name, best * 1e3, sum);
        CK(hipMemset(out, 0, (size_t)n * 16));
    };
    timeit("plain", [&] { plain<<<dim3((n + 255) / 256), dim3(256)>>>(pos, dadj, n, out); });
    {
        std::vector<int> rp((size_t)n + 1), fe((size_t)n + 1);
        for (int i = 0; i <= n; ++i) { rp[i] = i * 8; fe[i] = i * 4; }
        int *drp, *dfe; float4 *out2;
        CK(hipMalloc(&drp, rp.size() * 4)); CK(hipMalloc(&dfe, fe.size() * 4)); CK(hipMalloc(&out2, (size_t)n * 16));
        CK(hipMemcpy(drp, rp.data(), rp.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dfe, fe.data(), fe.size() * 4, hipMemcpyHostToDevice));
        timeit("plain_full (fused kernel's memory side)", [&] { plain_full<<<dim3((n + 255) / 256), dim3(256)>>>(pos, drp, dfe, dadj, n, out, out2); });
    }
    {
        float4 *out2; CK(hipMalloc(&out2, (size_t)n * 16));
        const int G0 = (n + 127) / 128;
        timeit("phase-A-like, list order, 8 at once", [&] { phase_a_like<0><<<dim3(G0), dim3(256)>>>(pos, dadj, dcnt, n, slice_rows, 0, out, out2); });
        const int G1 = (slice_rows + 127) / 128;
        timeit("phase-A-like, slice sweep, arrival order", [&] { phase_a_like<1><<<dim3(8 * G1), dim3(256)>>>(pos, dadj2, dcnt, n, slice_rows, G1, out, out2); });
    }
    timeit("plain (slice-sorted lists)", [&] { plain<<<dim3((n + 255) / 256), dim3(256)>>>(pos, dadj2, n, out); });
    const int tiles = (slice_rows + 255) / 256;
#define RUN(T, ORD)                                                                                                  \
    {                                                                                                                \
        const int G = (tiles + T - 1) / T;                                                                           \
        char name[64]; snprintf(name, sizeof name, "sweep2 T=%d %s (%d WGs)", T, ORD ? "ordered" : "unordered", 8 * G); \
        timeit(name, [&] { sweep2<T, ORD><<<dim3(8 * G), dim3(256)>>>(pos, dadj2, dcnt, dperm, n, slice_rows, G, out); }); \
    }
    {
        const int G = tiles;
        timeit("sweep2 T=1 LDS-staged, list order", [&] { sweep2_lds<<<dim3(8 * G), dim3(256)>>>(pos, dadj2, dcnt, dperm, n, slice_rows, G, out); });
    }
    RUN(1, false) RUN(2, false) RUN(4, false) RUN(8, false)
    RUN(1, true) RUN(2, true) RUN(4, true)
    return 0;
}
