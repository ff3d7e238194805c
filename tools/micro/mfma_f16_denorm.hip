// micro-test: does v_mfma_f32_16x16x32_f16 honour fp16 subnormal inputs?  A = all 2^-20 (subnormal in fp16),
// B = all 1.0 -> every output should be 32 * 2^-20 = 3.0518e-05; 0 means inputs were flushed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float *out, float a_val, float b_val) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
}
int main() {
    float *d, h[64];
    (void)hipMalloc(&d, 256);
    const float tests[3][2] = {{9.5367431640625e-07f, 1.0f}, {1.0f, 9.5367431640625e-07f}, {0.5f, 0.25f}};
    for (auto &t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
        (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
        printf("a=%g b=%g -> %.6e (expected %.6e)\n", t[0], t[1], h[0], 32.0 * (double)(float)(_Float16)t[0] * (double)(float)(_Float16)t[1]);
    }
    return 0;
}
