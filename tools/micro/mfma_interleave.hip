// micro-benchmark: per 16-slot step of the reverse edge kernel a wave issues ~130 VALU and 20 v_mfma_f32_16x16x32_f16
// (7 accumulator tiles, 3 rounds).  How many cycles does such a step take when the MFMAs are issued as one block in front of
// the VALU work, or one by one between VALU chunks -- and with the 32x32x16 shape (half as many, twice as long)?
// 1-3 waves per SIMD, no memory traffic: the best case for overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define VALU(n) _Pragma("unroll") for (int v_ = 0; v_ < (n); ++v_) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(vi++) % 32]) : "v"(m1), "v"(m2));

template <int MODE>   // 0: 20 small MFMAs then 130 VALU; 1: interleaved small; 2: 10 big (32x32x16) then VALU; 3: interleaved big; 4: VALU only; 5: small only
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed) {
    f32x4 c[7];
    f32x16 C[4];
    for (int i = 0; i < 7; ++i) c[i] = (f32x4){seed, 0, 0, (float)i};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) C[i][j] = seed * j;
    union { uint4 u; f16x8 v; } a, b;
    a.u = make_uint4(threadIdx.x, 2, 3, 4); b.u = make_uint4(5, 6, threadIdx.x, 8);
    float m1 = 1.0001f * seed, m2 = 0.5f * seed;
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = seed * i;
    for (int it = 0; it < iters; ++it) {
        int vi = 0;
        if (MODE == 0 || MODE == 5) {
#pragma unroll
            for (int i = 0; i < 20; ++i) { c[i % 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c[i % 7], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
            if (MODE == 0) { VALU(130) }
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 20; ++i) {
                c[i % 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c[i % 7], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                VALU(i < 10 ? 7 : 6)
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 10; ++i) { C[i % 4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, C[i % 4], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
            VALU(130)
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                C[i % 4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, C[i % 4], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                VALU(13)
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            VALU(130)
        }
    }
    float s = 0.f;
    for (int i = 0; i < 7; ++i) s += c[i][0] + c[i][3];
    for (int i = 0; i < 4; ++i) s += C[i][0] + C[i][15];
    for (int i = 0; i < 32; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, float *d, int wgs) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(256), 0, 0, d, iters, 1.f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-34s waves/SIMD=%d  %7.1f ns/step/wave-slot\n", name, wgs / 256, best * 1e6 / iters);
}
int main() {
    float *d;
    (void)hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((k<0>), dim3(768), dim3(256), 0, 0, d, 20000, 1.f);   // warm-up
    (void)hipDeviceSynchronize();
    for (int w = 1; w <= 3; ++w) {
        const int wgs = 256 * w;
        run<4>("valu x130", d, wgs);
        run<5>("mfma16 x20", d, wgs);
        run<0>("mfma16 x20 block + valu x130", d, wgs);
        run<1>("mfma16 x20 interleaved valu x130", d, wgs);
        run<2>("mfma32 x10 block + valu x130", d, wgs);
        run<3>("mfma32 x10 interleaved valu x130", d, wgs);
    }
    return 0;
}
