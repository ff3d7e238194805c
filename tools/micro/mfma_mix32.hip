// micro-benchmark (round 5): instruction mix of an edge-kernel step on the two fp16 MFMA shapes.  v_mfma_f32_16x16x32_f16 does not
// overlap with vector work of any wave on its SIMD (mfma_interleave.hip); v_mfma_f32_32x32x16_f16 does.  A re-tiling of the edge
// kernels onto 32 slots x [8 features x (a, b, c, scalars)] tiles would issue 12 (forward) / 24 (reverse) of the large shape per
// 32 slots x 16 features instead of 22 / 40 of the small one, with the same vector work.  ns per 32 slots x 16 features and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// REP repetitions per iteration of [NM MFMAs round-robin over NACC accumulators, then NV independent v_fma_f32]
template <int BIG, int NM, int NACC, int NV, int REP>
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
        for (int rep = 0; rep < REP; ++rep) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                if (BIG) C[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, C[i % NACC], 0, 0, 0);
                else c[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c[i % NACC], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v % 32]) : "v"(m1), "v"(m2));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
    for (int i = 0; i < 4; ++i) s += C[i][0] + C[i][15];
    for (int i = 0; i < 32; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int BIG, int NM, int NACC, int NV, int REP>
void run(const char *name, float *d) {
    const int iters = 10000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-64s", name);
    for (int w = 1; w <= 4; ++w) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((k<BIG, NM, NACC, NV, REP>), dim3(256 * w), dim3(256), 0, 0, d, iters, 1.f);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // one iteration = one unit (32 slots x 16 features) per wave; w waves per SIMD -> ns per unit and SIMD = time / (iters * w)
        printf("  w%d %7.1f", w, best * 1e6 / iters / w);
    }
    printf("\n");
}
int main() {
    float *d;
    (void)hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<0, 20, 7, 130, 2>), dim3(512), dim3(256), 0, 0, d, 10000, 1.f);
    (void)hipDeviceSynchronize();
    printf("ns per unit (32 slots x 16 features) and SIMD, at w waves per SIMD\n");
    run<0, 20, 7, 130, 2>("reverse now:  2 x [20 small + 130 valu]", d);
    run<1, 12, 2, 130, 2>("reverse 32:   2 x [12 large (2 acc) + 130 valu]", d);
    run<1, 24, 4, 260, 1>("reverse 32:   24 large (4 acc) + 260 valu", d);
    run<1, 20, 4, 260, 1>("reverse 32:   20 large (4 acc) + 260 valu (no padding rows)", d);
    run<0, 11, 4, 50, 2>("forward now:  2 x [11 small + 50 valu]", d);
    run<1, 6, 1, 50, 2>("forward 32:   2 x [6 large (1 acc) + 50 valu]", d);
    run<1, 12, 2, 100, 1>("forward 32:   12 large (2 acc) + 100 valu", d);
    run<0, 0, 1, 260, 1>("valu x260", d);
    run<1, 24, 4, 0, 1>("large x24", d);
    run<0, 40, 7, 0, 1>("small x40", d);
    return 0;
}
