// Selectivity and conservativeness of the split-f16 MFMA pre-filter (scan_core.h helpers) on random data:
// passes vs true candidates vs MISSED candidates (must be 0), also with all inputs scaled by powers of two.
#include "../../graphem-rapids_amd/csrc/scan_core.h"
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const float *q /*32x4: x y z tau*/, const float *m /*n x 4*/, int n, float sc, int *pass, int *truec, int *missed) {
    const int lane = threadIdx.x & 63, col = lane & 31, hsel = lane >> 5;
    __shared__ _Float16 rows[32][16];
    if (threadIdx.x < 32) {
        float qs[3] = {q[threadIdx.x * 4] * sc, q[threadIdx.x * 4 + 1] * sc, q[threadIdx.x * 4 + 2] * sc};
        gh_mf_query_row(qs, 3, q[threadIdx.x * 4 + 3] * sc * sc, rows[threadIdx.x]);
    }
    __syncthreads();
    gh_h8 a;
    for (int e = 0; e < 8; ++e) a[e] = rows[col][8 * hsel + e];
    for (int base = blockIdx.x * 32; base < n; base += gridDim.x * 32) {
        const int j = base + col;
        float mv[3] = {m[j * 4] * sc, m[j * 4 + 1] * sc, m[j * 4 + 2] * sc};
        gh_h8 b;
        gh_mf_ref_col(mv, true, hsel, b);
        const gh_f16x zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const gh_f16x f = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0);
        for (int i = 0; i < 16; ++i) {
            const int s = (i & 3) + 8 * (i >> 2) + 4 * hsel;
            float d2 = 0.f;
            for (int d = 0; d < 3; ++d) { const float df = q[s * 4 + d] - m[j * 4 + d]; d2 = fmaf(df, df, d2); }
            const bool t = d2 <= q[s * 4 + 3], p = f[i] < 0.0f;
            if (p) atomicAdd(pass, 1);
            if (t) atomicAdd(truec, 1);
            if (t && !p) atomicAdd(missed, 1);
        }
    }
}
int main() {
    const int n = 1 << 20;
    std::mt19937 g(1); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> q(32 * 4), m((size_t)n * 4);
    for (int i = 0; i < 32; ++i) { q[i * 4] = nd(g); q[i * 4 + 1] = nd(g); q[i * 4 + 2] = nd(g); q[i * 4 + 3] = 0.0226f; }
    for (int i = 0; i < n; ++i) { m[(size_t)i * 4] = nd(g); m[(size_t)i * 4 + 1] = nd(g); m[(size_t)i * 4 + 2] = nd(g); m[(size_t)i * 4 + 3] = 0; }
    float *dq, *dm; int *dc;
    hipMalloc(&dq, q.size() * 4); hipMalloc(&dm, m.size() * 4); hipMalloc(&dc, 12);
    hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dm, m.data(), m.size() * 4, hipMemcpyHostToDevice);
    for (float sc : {1.0f, 0.5f, 0.25f, 0.125f, 0.0625f}) {
        hipMemset(dc, 0, 12);
        k<<<256, 64>>>(dq, dm, n, sc, dc, dc + 1, dc + 2);
        int h[3]; hipMemcpy(h, dc, 12, hipMemcpyDeviceToHost);
        printf("scale %-7g pass %d true %d missed %d\n", sc, h[0], h[1], h[2]);
    }
    return 0;
}
