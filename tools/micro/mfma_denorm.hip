// Does v_mfma_f32_32x32x16_f16 honour f16 subnormal inputs, and how exact is its accumulation?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16x __attribute__((ext_vector_type(16)));
__global__ void k(float *out, float a0, float b0, float a1, float b1, float a2, float b2) {
    const int lane = threadIdx.x & 63;
    h8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < 32) {  // k = 0..7 live in lanes 0..31
        a[0] = (_Float16)a0; b[0] = (_Float16)b0;
        a[1] = (_Float16)a1; b[1] = (_Float16)b1;
        a[2] = (_Float16)a2; b[2] = (_Float16)b2;
    }
    const f16x zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const f16x f = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = f[0];
}
int main() {
    float *d; hipMalloc(&d, 4);
    auto run = [&](const char *name, float a0, float b0, float a1, float b1, float a2, float b2, double want) {
        k<<<1, 64>>>(d, a0, b0, a1, b1, a2, b2);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("%-44s got %.10g want %.10g\n", name, h, want);
    };
    const float sub = 3.0517578125e-05f;  // 2^-15: f16 subnormal
    run("subnormal * 1", sub, 1.0f, 0, 0, 0, 0, sub);
    run("1 * subnormal", 1.0f, sub, 0, 0, 0, 0, sub);
    run("smallest subnormal 2^-24 * 1", 5.9604644775390625e-08f, 1.0f, 0, 0, 0, 0, 5.9604644775390625e-08);
    run("1*1 + 1*2^-15 (small next to big)", 1.0f, 1.0f, 1.0f, sub, 0, 0, 1.0 + sub);
    run("1*1 + (-1)*1 + 2^-15*1 (cancellation)", 1.0f, 1.0f, -1.0f, 1.0f, sub, 1.0f, sub);
    run("0.0498*1 - 0.0495*1 + 2^-13*0.124", 0.0498046875f, 1.0f, -0.049560546875f, 1.0f, 0.0001220703125f, 0.1240234375f,
        0.0498046875 - 0.049560546875 + 0.0001220703125 * 0.1240234375);
    run("1001*1 - 1000*1 + 2^-10*1", 1001.0f, 1.0f, -1000.0f, 1.0f, 0.0009765625f, 1.0f, 1.0009765625);
    return 0;
}
