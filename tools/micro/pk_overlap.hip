// micro-benchmark: v_pk_fma_f32 against pairs of v_fma_f32, alone and interleaved with v_mfma_f32_16x16x32_f16, 1 / 2 / 4 waves per
// SIMD.  Question: does the packed form halve the issue time of the hot loops' fp32 arithmetic, and does it still overlap with the
// 16-bit matrix instructions?  KM independent MFMA chains + NPAIR fp32 fma pairs per loop iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int KM, int NPAIR, int PK>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed) {
    f32x4 c[6];
    for (int i = 0; i < 6; ++i) c[i] = (f32x4){seed, 0.f, (float)i, 0.f};
    union { uint4 u; f16x8 v; } a, b;
    a.u = make_uint4(threadIdx.x, 2, 3, 4); b.u = make_uint4(5, 6, threadIdx.x, 8);
    const f32x2 m1 = {1.0001f * seed, 0.9999f * seed}, m2 = {0.5f * seed, 0.25f * seed};
    f32x2 x[12];
    for (int i = 0; i < 12; ++i) x[i] = (f32x2){seed * i, seed - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (KM > g) c[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c[g], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NPAIR / 6; ++j) {
                const int i = (g * (NPAIR / 6) + j) % 12;
                if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m1), "v"(m2));
                else {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i].x) : "v"(m1.x), "v"(m2.x));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i].y) : "v"(m1.y), "v"(m2.y));
                }
            }
        }
    }
    f32x4 s4 = c[0] + c[1] + c[2] + c[3] + c[4] + c[5];
    float s = s4[0] + s4[1] + s4[2] + s4[3];
    for (int i = 0; i < 12; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KM, int NPAIR, int PK>
void run(float *d, int wgs) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<KM, NPAIR, PK>), dim3(wgs), dim3(256), 0, 0, d, iters, 1.f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("mfma %d + %2d fma pairs as %-14s waves/SIMD=%d  %7.3f ms = %6.1f ns / iteration\n", KM, NPAIR, PK ? "v_pk_fma_f32" : "2 x v_fma_f32",
           wgs / 256, best, best * 1e6 / iters);
}
int main() {
    float *d;
    (void)hipMalloc(&d, 1024 * 256 * sizeof(float));
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((k<6, 24, 0>), dim3(768), dim3(256), 0, 0, d, 20000, 1.f);   // warm-up
    (void)hipDeviceSynchronize();
    for (int wgs : {256, 512, 1024}) {
        run<0, 24, 0>(d, wgs); run<0, 24, 1>(d, wgs);
        run<6, 0, 0>(d, wgs);
        run<6, 24, 0>(d, wgs); run<6, 24, 1>(d, wgs);
        run<6, 48, 0>(d, wgs); run<6, 48, 1>(d, wgs);
    }
    return 0;
}
