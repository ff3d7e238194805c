// Issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950: N independent chains per thread, one wave per SIMD (256 threads
// per workgroup, one workgroup per CU) or two.  Prints cycles per instruction and wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int PK>
__global__ void __launch_bounds__(256) k(float *out, int iters, float s) {
    f2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = (f2){threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i};
    const f2 m = {s, s * 0.5f}, c = {1e-3f, 2e-3f};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (PK) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            } else {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(m.y), "v"(c.y));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

int main() {
    float *d;
    hipMalloc(&d, 4 * 256 * 2048);
    const int iters = 20000;
    for (int wgs : {256, 512}) {
        for (int pk = 0; pk < 2; ++pk) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (pk) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, d, iters, 1.0001f);
                else hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, d, iters, 1.0001f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)wgs * 256 * iters * 8 * 2 * 2;   // 8 pairs x 2 lanes-halves x (mul + add)
            printf("workgroups %d (%d wave(s) per SIMD) %s: %.3f ms, %.1f TFLOP/s\n", wgs, wgs / 256, pk ? "v_pk_fma_f32" : "2 x v_fma_f32",
                   ms, flops / ms * 1e-9);
        }
    }
    return 0;
}
