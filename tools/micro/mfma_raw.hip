// micro-test: is the result of a chain of dependent v_mfma_f32_16x16x32_bf16 safe to read with the wait states the
// compiler inserts?  Path A reads the accumulators immediately (compiler-chosen nops), path B after a long nop chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union Op { uint4 u; bf16x8 v; };
template <int NCHAIN, bool DELAY>
__device__ __forceinline__ void chains(const Op (&w)[3][3], const Op (&r)[3], float (&out)[12], float scale) {
    f32x4 acc[3];
#pragma unroll
    for (int c = 0; c < NCHAIN; ++c) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0].v, r[2].v, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][2].v, r[0].v, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][1].v, r[1].v, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0].v, r[1].v, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][1].v, r[0].v, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0].v, r[0].v, a, 0, 0, 0);
        acc[c] = a;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (DELAY) asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                            "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NCHAIN; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) out[c * 4 + k] = acc[c][k] * scale;
}
template <int NCHAIN>
__global__ void __launch_bounds__(768) k(const uint4 *wsrc, const uint4 *rsrc, float *oa, float *ob, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    Op w[3][3], r[3];
    for (int c = 0; c < 3; ++c) for (int i = 0; i < 3; ++i) w[c][i].u = wsrc[(threadIdx.x * 9 + c * 3 + i) % 4096];
    float sa[12] = {}, sb[12] = {};
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < 3; ++i) r[i].u = rsrc[((size_t)t * 3 + i + it * 7) % 65536];
        float o[12];
        chains<NCHAIN, false>(w, r, o, 1.0f);
        for (int q = 0; q < NCHAIN * 4; ++q) sa[q] += o[q];
        chains<NCHAIN, true>(w, r, o, 1.0f);
        for (int q = 0; q < NCHAIN * 4; ++q) sb[q] += o[q];
    }
    for (int q = 0; q < NCHAIN * 4; ++q) { oa[(size_t)t * 12 + q] = sa[q]; ob[(size_t)t * 12 + q] = sb[q]; }
}
int main() {
    const int n = 768 * 256;
    std::vector<unsigned> hw(4096 * 4), hr(65536 * 4);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    auto bf = [&]() { unsigned e = 0x3F00u + (rnd() % 0x100u); return (e | ((rnd() & 1u) << 15)) & 0xFFFFu; };   // |x| in [0.5, 2)
    for (auto &x : hw) x = bf() | (bf() << 16);
    for (auto &x : hr) x = bf() | (bf() << 16);
    uint4 *dw, *dr; float *oa, *ob;
    (void)hipMalloc(&dw, hw.size() * 4); (void)hipMalloc(&dr, hr.size() * 4);
    (void)hipMalloc(&oa, (size_t)n * 48); (void)hipMalloc(&ob, (size_t)n * 48);
    (void)hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> ha((size_t)n * 12), hb((size_t)n * 12);
    for (int nch = 2; nch <= 3; ++nch) {
        for (int rep = 0; rep < 3; ++rep) {
            if (nch == 2) hipLaunchKernelGGL(k<2>, dim3(n / 768), dim3(768), 0, 0, dw, dr, oa, ob, 200);
            else hipLaunchKernelGGL(k<3>, dim3(n / 768), dim3(768), 0, 0, dw, dr, oa, ob, 200);
            (void)hipMemcpy(ha.data(), oa, ha.size() * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hb.data(), ob, hb.size() * 4, hipMemcpyDeviceToHost);
            long bad = 0;
            for (size_t i = 0; i < ha.size(); ++i) if ((i % 12) < (size_t)nch * 4 && ha[i] != hb[i]) ++bad;
            printf("chains=%d rep %d: %ld of %zu accumulated outputs differ between immediate and delayed read\n", nch, rep, bad,
                   ha.size() / 12 * nch * 4);
        }
    }
    return 0;
}
