// micro-benchmark: cost of dependent accumulator chains.  NM MFMAs per step dealt round-robin over NACC accumulators (the
// predecessor of an MFMA in its chain is NACC instructions back), followed by NV independent v_fma_f32; both shapes.
// Question behind it: the 32x32 forward edge kernel has 2 tiles x 6 dependent MFMAs per step -- does distance 2 stall?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int BIG, int NM, int NACC, int NV>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed) {
    f32x4 c[8];
    f32x16 C[4];
    for (int i = 0; i < 8; ++i) c[i] = (f32x4){seed, 0, 0, (float)i};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) C[i][j] = seed * j;
    union { uint4 u; f16x8 v; } a, b;
    a.u = make_uint4(threadIdx.x, 2, 3, 4); b.u = make_uint4(5, 6, threadIdx.x, 8);
    float m1 = 1.0001f * seed, m2 = 0.5f * seed;
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = seed * i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if (BIG) C[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, C[i % NACC], 0, 0, 0);
            else c[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c[i % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v % 32]) : "v"(m1), "v"(m2));
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
    for (int i = 0; i < 4; ++i) s += C[i][0] + C[i][15];
    for (int i = 0; i < 32; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// 6 tiles x 3 products in different issue orders (G = tiles per group: rounds run inside a group of G tiles)
template <int G, int NV>
__global__ void __launch_bounds__(256) k6(float *out, int iters, float seed) {
    f32x4 c[6];
    for (int i = 0; i < 6; ++i) c[i] = (f32x4){seed, 0, 0, (float)i};
    union { uint4 u; f16x8 v; } a[6], b[3];
    for (int i = 0; i < 6; ++i) a[i].u = make_uint4(threadIdx.x, 2 + i, 3, 4);
    for (int i = 0; i < 3; ++i) b[i].u = make_uint4(5, 6 + i, threadIdx.x, 8);
    float m1 = 1.0001f * seed, m2 = 0.5f * seed;
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = seed * i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g0 = 0; g0 < 6; g0 += G)
#pragma unroll
            for (int kk = 0; kk < 3; ++kk)
#pragma unroll
                for (int t = g0; t < g0 + G; ++t) {
                    c[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t].v, b[kk].v, c[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v % 32]) : "v"(m1), "v"(m2));
    }
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += c[i][0] + c[i][3];
    for (int i = 0; i < 32; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int G, int NV>
void run6(float *d) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("16x16x32 6 tiles x 3 products, rounds inside groups of %d tiles, valu=%-3d :", G, NV);
    for (int w = 1; w <= 4; ++w) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((k6<G, NV>), dim3(256 * w), dim3(256), 0, 0, d, iters, 1.f);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("  w%d %7.1f", w, best * 1e6 / iters / w);
    }
    printf("\n");
}
template <int BIG, int NM, int NACC, int NV>
void run(float *d) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%s x%-2d acc=%d valu=%-3d :", BIG ? "32x32x16" : "16x16x32", NM, NACC, NV);
    for (int w = 1; w <= 4; ++w) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((k<BIG, NM, NACC, NV>), dim3(256 * w), dim3(256), 0, 0, d, iters, 1.f);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("  w%d %7.1f", w, best * 1e6 / iters / w);   // ns per wave-step
    }
    printf("   (ns per wave-step at 1..4 waves/SIMD)\n");
}
int main() {
    float *d;
    (void)hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((k<0, 20, 7, 130>), dim3(768), dim3(256), 0, 0, d, 20000, 1.f);   // warm-up
    (void)hipDeviceSynchronize();
    run<1, 12, 1, 0>(d); run<1, 12, 2, 0>(d); run<1, 12, 4, 0>(d);
    run<0, 12, 1, 0>(d); run<0, 12, 2, 0>(d); run<0, 12, 4, 0>(d); run<0, 12, 6, 0>(d);
    run<1, 12, 2, 84>(d); run<1, 12, 4, 84>(d);       // the 32x32 forward edge step: 32 slots
    run<0, 11, 4, 50>(d); run<0, 22, 4, 100>(d);      // the 16x16 forward edge step: 16 slots, and two of them
    run6<6, 0>(d); run6<3, 0>(d); run6<2, 0>(d); run6<1, 0>(d);
    run6<6, 40>(d); run6<3, 40>(d); run6<2, 40>(d); run6<1, 40>(d);
    run<0, 0, 1, 84>(d); run<0, 0, 1, 50>(d);
    return 0;
}
