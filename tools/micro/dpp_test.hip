// micro-test: fused v_add_f32_dpp quad sums vs the builtin (mov_dpp + add) version, under partial exec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ float quad_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    return x;
}
__device__ __forceinline__ void quad_sum4(float (&x)[4]) {
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
}
__global__ void k(const float *in, float *a, float *b, int mode) {
    int t = threadIdx.x + blockIdx.x * blockDim.x;
    float x[4], y[4];
    for (int r = 0; r < 4; ++r) { x[r] = in[t * 4 + r]; y[r] = x[r]; }
    bool act = mode == 0 ? true : ((t >> 4) & 1) == 0;   // mode 1: alternate 16-lane groups active
    if (mode == 2) act = ((t >> 2) % 3) != 0;             // mode 2: some quads off
    if (act) {
        for (int r = 0; r < 4; ++r) x[r] = quad_sum(x[r]);
        quad_sum4(y);
    }
    for (int r = 0; r < 4; ++r) { a[t * 4 + r] = x[r]; b[t * 4 + r] = y[r]; }
}
int main() {
    const int n = 256 * 64;
    std::vector<float> h(n * 4), ha(n * 4), hb(n * 4);
    for (int i = 0; i < n * 4; ++i) h[i] = (float)((i * 2654435761u) % 10007) / 97.f - 40.f;
    float *d, *a, *b;
    hipMalloc(&d, n * 16); hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, a, b, mode);
        hipMemcpy(ha.data(), a, n * 16, hipMemcpyDeviceToHost);
        hipMemcpy(hb.data(), b, n * 16, hipMemcpyDeviceToHost);
        int bad = 0, first = -1;
        for (int i = 0; i < n * 4; ++i) if (ha[i] != hb[i]) { if (first < 0) first = i; ++bad; }
        printf("mode %d: mismatches %d of %d", mode, bad, n * 4);
        if (first >= 0) printf("  first at thread %d r %d: builtin %.6f asm %.6f in %.6f", first / 4, first % 4, ha[first], hb[first], h[first]);
        printf("\n");
    }
    return 0;
}
