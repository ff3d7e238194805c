// micro-benchmark: do v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32 overlap with independent VALU work?
// KM independent MFMA chains and KV independent (unpacked) VALU fma per loop iteration, 1-3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MF(i) if (KM > i) { if (KIND == 0) c##i = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c##i, 0, 0, 0); \
                            else c##i = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c##i, 0, 0, 0); }
#define VA(i) if (KV > i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i % 24]) : "v"(m1), "v"(m2));
template <int KM, int KV, int KIND>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed, long long *cyc) {
    f32x4 c0 = {seed, 0, 0, 0}, c1 = {0, seed, 0, 0}, c2 = {0, 0, seed, 0}, c3 = {0, 0, 0, seed}, c4 = {seed, 1, 0, 0},
          c5 = {seed, 0, 1, 0};
    union { uint4 u; bf16x8 v; } a, b;
    a.u = make_uint4(threadIdx.x, 2, 3, 4); b.u = make_uint4(5, 6, threadIdx.x, 8);
    float fa = seed + threadIdx.x, fb = seed * 0.5f, m1 = 1.0001f * seed, m2 = 0.5f * seed;
    float x[24];
    for (int i = 0; i < 24; ++i) x[i] = seed * i;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        MF(0) VA(0) VA(1) VA(2) VA(3) VA(4) VA(5) VA(6) VA(7)
        MF(1) VA(8) VA(9) VA(10) VA(11) VA(12) VA(13) VA(14) VA(15)
        MF(2) VA(16) VA(17) VA(18) VA(19) VA(20) VA(21) VA(22) VA(23)
        MF(3) VA(24) VA(25) VA(26) VA(27) VA(28) VA(29) VA(30) VA(31)
        MF(4) VA(32) VA(33) VA(34) VA(35) VA(36) VA(37) VA(38) VA(39)
        MF(5) VA(40) VA(41) VA(42) VA(43) VA(44) VA(45) VA(46) VA(47)
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    f32x4 s4 = c0 + c1 + c2 + c3 + c4 + c5;
    float s = s4[0] + s4[1] + s4[2] + s4[3];
    for (int i = 0; i < 24; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KM, int KV, int KIND>
void run(const char *name, float *d, int wgs) {
    static long long *cyc = nullptr;
    if (!cyc) (void)hipHostMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<KM, KV, KIND>), dim3(wgs), dim3(256), 0, 0, d, iters, 1.f, cyc);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-26s waves/SIMD=%d  min %7.3f ms = %6.1f ns/iter = %6.1f cycles @2.4GHz\n", name, wgs / 256, best,
           best * 1e6 / iters, best * 1e6 / iters * 2.4);
}
int main() {
    float *d;
    (void)hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
    long long *cyc;
    (void)hipHostMalloc(&cyc, 8);
    for (int i = 0; i < 60; ++i) hipLaunchKernelGGL((k<6, 48, 1>), dim3(768), dim3(256), 0, 0, d, 20000, 1.f, cyc);   // warm-up ~0.6 s
    (void)hipDeviceSynchronize();
    for (int w = 1; w <= 3; ++w) {
        const int wgs = 256 * w;
        run<0, 48, 0>("valu x48", d, wgs);
        run<6, 0, 0>("bf16 mfma x6", d, wgs);
        run<6, 24, 0>("bf16 mfma x6 + valu x24", d, wgs);
        run<6, 48, 0>("bf16 mfma x6 + valu x48", d, wgs);
        run<6, 0, 1>("f32 mfma x6", d, wgs);
        run<6, 24, 1>("f32 mfma x6 + valu x24", d, wgs);
        run<6, 48, 1>("f32 mfma x6 + valu x48", d, wgs);
    }
    return 0;
}
