// mfma16.h — shared device primitives of the fp16-split matrix path (node kernels, layer-0 factorisation):
// vector typedefs, global-address-space loads, activation tiles as fp16 h / l planes in LDS.
#ifndef VSSR_MFMA16_H
#define VSSR_MFMA16_H
#include "vssr_internal.h"

namespace vssr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int TA = 32;            // atoms per workgroup
constexpr int NW = 8;             // waves per workgroup
constexpr int NTHREADS = 64 * NW;
constexpr int PADH = 8;           // row pad of the fp16 planes (halves): conflict-free ds_read_b128 of 16 rows

// sigmoid on the hardware exp2 / rcp units (1 ulp each) instead of libdevice expf + IEEE division (~35 issue slots, and
// the node kernels evaluate 24 .. 32 of them per lane: a fifth of their vector instructions).  The rounding error of the
// argument product x * (-log2 e), which would grow with |x|, is recovered exactly with one fma and applied to first order,
// so the result stays within ~2 ulp of the correctly rounded sigmoid over the whole range.
#ifndef VSSR_FAST_SIGMOID
#define VSSR_FAST_SIGMOID 1
#endif
__device__ __forceinline__ float sigm(float x) {
#if VSSR_FAST_SIGMOID && defined(__HIP_DEVICE_COMPILE__)
    constexpr float C_HI = -1.44269502162933349609375f;      // fp32(-log2 e)
    constexpr float C_LO = -1.9259629911266175e-08f;         // -log2 e - C_HI
    constexpr float LN2 = 0.693147182464599609375f;
    x = fmaxf(x, -87.f);                                     // exp(-x) stays finite (sigmoid(-87) = 1.6e-38); NaN propagates
    const float y = x * C_HI;
    const float r = fmaf(x, C_LO, fmaf(x, C_HI, -y));       // exact product error + low part of the constant
    const float e = __builtin_amdgcn_exp2f(y);               // exp(-x) up to the factor 2^r
    const float en = fmaf(e, r * LN2, e);
    return __builtin_amdgcn_rcpf(1.f + en);
#else
    return 1.f / (1.f + expf(-x));
#endif
}
__device__ __forceinline__ float swish(float x) { return x * sigm(x); }
__device__ __forceinline__ float dswish(float x) {
    float sg = sigm(x);
    return sg * fmaf(x, 1.f - sg, 1.f);
}

// Weight pointers are read out of the ModelW table in memory, so the compiler would treat them as FLAT (flat loads
// tick both vmcnt and lgkmcnt).  Loading through an explicit global address space pointer gives global_load.
__device__ __forceinline__ u32x4 gload4u(const uint4 *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const u32x4 __attribute__((address_space(1))) *gptr;
    return *reinterpret_cast<gptr>(reinterpret_cast<uintptr_t>(p));
#else
    return *reinterpret_cast<const u32x4 *>(p);
#endif
}

// ---- activation tiles as fp16 planes -----------------------------------------------------------------------------------
struct Planes {   // [rows][K] in two planes; ld = K + PADH halves
    _Float16 *h, *l;
    int ld;
};
__device__ __forceinline__ Planes make_planes(_Float16 *base, int rows, int K) {
    return Planes{base, base + (size_t)rows * (K + PADH), K + PADH};
}
constexpr int plane_halves(int rows, int K) { return 2 * rows * (K + PADH); }

__device__ __forceinline__ void split1(float x, _Float16 &h, _Float16 &l) {
    const float xc = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    h = (_Float16)xc;
    l = (_Float16)(xc - (float)h);
}
__device__ __forceinline__ void store_split(const Planes &P, int row, int col, float x) {
    _Float16 h, l;
    split1(x, h, l);
    P.h[row * P.ld + col] = h;
    P.l[row * P.ld + col] = l;
}
// four consecutive columns at once (col multiple of 4): two 8-byte LDS stores
__device__ __forceinline__ void store_split4(const Planes &P, int row, int col, float4 v) {
    const f32x2 a = {__builtin_amdgcn_fmed3f(v.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.y, -65504.f, 65504.f)};
    const f32x2 b = {__builtin_amdgcn_fmed3f(v.z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.w, -65504.f, 65504.f)};
    const f16x2 ha = __builtin_convertvector(a, f16x2), hb = __builtin_convertvector(b, f16x2);
    const f16x2 la = __builtin_convertvector(a - __builtin_convertvector(ha, f32x2), f16x2);
    const f16x2 lb = __builtin_convertvector(b - __builtin_convertvector(hb, f32x2), f16x2);
    *reinterpret_cast<u32x2 *>(P.h + row * P.ld + col) = (u32x2){__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
    *reinterpret_cast<u32x2 *>(P.l + row * P.ld + col) = (u32x2){__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
}

__device__ __forceinline__ void store_split4(const Planes &P, int row, int col, f32x4 v) {
    store_split4(P, row, col, make_float4(v[0], v[1], v[2], v[3]));
}

// Cooperative tile load: NROWS rows of F floats from global, split, into plane columns [col0, col0 + F).  rowptr(row)
// must always return a readable row (tail rows are clamped to the last atom; their results are never stored), so that
// all loads are unconditional and issued back-to-back before the first LDS store.
template <int NROWS, class RowPtr>
__device__ __forceinline__ void load_rows_split(const Planes &P, int col0, RowPtr rowptr) {
    constexpr int NIT = NROWS * (F / 4) / NTHREADS;
    static_assert(NROWS * (F / 4) % NTHREADS == 0, "tile load must divide evenly");
    float4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = threadIdx.x + it * NTHREADS;
        v[it] = *reinterpret_cast<const float4 *>(rowptr(idx >> 5) + 4 * (idx & 31));
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = threadIdx.x + it * NTHREADS;
        store_split4(P, idx >> 5, col0 + 4 * (idx & 31), v[it]);
    }
}
// same in two halves, and the fp32 values stay with the threads that loaded them (residual operands of a later output
// pass): rows_request() issues the loads, rows_store_split() writes the planes -- other requests (weight pieces) can be
// queued behind the tile's loads and arrive while the tile is being split
template <int NROWS, class RowPtr>
__device__ __forceinline__ void rows_request(RowPtr rowptr, float4 (&v)[NROWS * (F / 4) / NTHREADS]) {
    constexpr int NIT = NROWS * (F / 4) / NTHREADS;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = threadIdx.x + it * NTHREADS;
        v[it] = *reinterpret_cast<const float4 *>(rowptr(idx >> 5) + 4 * (idx & 31));
    }
}
template <int NROWS>
__device__ __forceinline__ void rows_store_split(const Planes &P, int col0, const float4 (&v)[NROWS * (F / 4) / NTHREADS]) {
    constexpr int NIT = NROWS * (F / 4) / NTHREADS;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = threadIdx.x + it * NTHREADS;
        store_split4(P, idx >> 5, col0 + 4 * (idx & 31), v[it]);
    }
}

// ---- coalesced global I/O for data that lives in the accumulator layout --------------------------------------------------
// In the accumulator layout a lane owns ONE column and 4 rows per tile: written or read directly, every dword access
// moves 16 contiguous bytes per lane quad (a quarter of what an L1 access can carry), and the kernels' epilogues were bound
// by the L1 access rate (update_fwd: 30 % of its time).  Results go through an fp32 tile in LDS instead
// (T[row][FT], conflict-free for the accumulator layout), and a cooperative pass moves whole rows as float4 per lane.
constexpr int FT = F + 4;   // row stride of the staging tile (floats): rows stay 16-byte aligned, 4 rows apart = 16 banks apart
template <int NROWS, class Fn>
__device__ __forceinline__ void stage_rows(const float *T, Fn fn) {   // fn(row, first column, the tile's 4 values)
    constexpr int NIT = NROWS * (F / 4) / NTHREADS;
    static_assert(NROWS * (F / 4) % NTHREADS == 0, "tile pass must divide evenly");
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = threadIdx.x + it * NTHREADS, row = idx >> 5, c4 = idx & 31;
        fn(row, 4 * c4, *reinterpret_cast<const float4 *>(T + row * FT + 4 * c4));
    }
}

// same pass with a residual operand from global memory: all loads of the pass are issued before the first store
template <int NROWS, class Ld, class St>
__device__ __forceinline__ void stage_rows_residual(const float *T, Ld ld, St st) {   // ld(row, col) -> float4 ; st(row, col, tile, residual)
    constexpr int NIT = NROWS * (F / 4) / NTHREADS;
    static_assert(NROWS * (F / 4) % NTHREADS == 0, "tile pass must divide evenly");
    float4 r[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = threadIdx.x + it * NTHREADS;
        r[it] = ld(idx >> 5, 4 * (idx & 31));
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = threadIdx.x + it * NTHREADS, row = idx >> 5, c4 = idx & 31;
        st(row, 4 * c4, *reinterpret_cast<const float4 *>(T + row * FT + 4 * c4), r[it]);
    }
}

__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// lane geometry shared by all kernels: wave w owns columns 16 w .. 16 w + 15.  The WEIGHTS are the A operand of the MFMA
// and the activations the B operand (gemm16), so the 16 x 16 result tile comes out as D[feature][atom]: accumulator
// element (t, i) of a lane is atom row 16 t + (lane & 15) and column 16 w + 4 (lane >> 4) + i -- FOUR CONSECUTIVE FEATURES
// of one atom.  Everything a lane stores or fetches in this layout is a 16-byte (fp32) or 8-byte (fp16 piece) vector:
// plane stores are ds_write_b64 with packed conversions, staging-tile and bias accesses are b128.  (With the operands the
// other way round a lane owned one feature of four atoms: 2-byte plane stores, dword tile accesses.)
struct LaneGeo {
    int w, col0, r;
    __device__ __forceinline__ LaneGeo() {
        const int lane = threadIdx.x & 63;
        w = threadIdx.x >> 6;
        col0 = 16 * w + 4 * (lane >> 4);
        r = lane & 15;
    }
    __device__ __forceinline__ int row(int t) const { return 16 * t + r; }
};
// four consecutive floats (16-byte aligned) through the global address space (see gload4u)
__device__ __forceinline__ f32x4 gload4f(const float *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const f32x4 __attribute__((address_space(1))) *gptr;
    return *reinterpret_cast<gptr>(reinterpret_cast<uintptr_t>(p));
#else
    return *reinterpret_cast<const f32x4 *>(p);
#endif
}

}  // namespace vssr
#endif
