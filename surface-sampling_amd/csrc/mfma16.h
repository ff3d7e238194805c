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

// Saturation of the fp16 split.  An activation beyond +-65504 cannot be represented by the h piece and is clamped (the
// results stay finite, but they are no longer the model's).  Every split is watched and attributed to its ATOM (TA = 32 rows
// per tile, tile row = x * TA + atom), so that a chain which shares an atom tile with a saturating neighbour is not
// touched.  At the end of a kernel the (rare) hits raise the flag of the chain of exactly that atom;
// vssr_batch_saturated() / results["saturated"] report the flags.  (NaN does not order: a non-finite energy is flagged by
// k_finalize_energy instead.)
// Cost matters: the reverse update kernel runs at the 256-register limit.  Measured there (same-box A/Bs, ms / step):
// one running maximum per thread 2.40 -> 2.40; a per-thread bit mask of rows (compare / shift / select / or) -> 3.07
// (125 spilled registers); lane masks of the compares folded on the scalar unit -> 284 spilled SGPRs.  So the node kernels
// keep running maxima (two max3 per four elements), one per row a thread can meet: both row -> thread maps that occur
// put a thread on the rows rho and rho + 16 of the tile,
//   SAT_ACC   accumulator layout   row = 16 t + (lane & 15)     rho = lane & 15      (LaneGeo)
//   SAT_COOP  cooperative passes   row = (tid >> 5) + 16 it      rho = tid >> 5       (load_rows_split)
// and bit 4 of the row is a constant of the unrolled code, so the choice among the four maxima costs nothing.
// SAT_ANY (arbitrary maps: the layer-0 staging pass, registers to spare) keeps a bit mask of rows.
// A tracker only costs while it is live: the cooperative passes use their own instance, committed right behind the pass, and
// a pass whose values another kernel of the same evaluation has already watched (the reverse update kernel re-reads the
// forward kernel's inputs and recomputes its intermediates) takes SatNone.
#ifndef VSSR_SAT_TRACK
#define VSSR_SAT_TRACK 1   // 0: A/B builds that measure what the tracking costs (tools/build_variant.sh); never shipped
#endif
enum { SAT_ACC = 0, SAT_COOP = 1, SAT_ANY = 2 };
struct SatTrack {
    float acc_lo = 0.f, acc_hi = 0.f, coop_lo = 0.f, coop_hi = 0.f;
    unsigned rows = 0u;   // SAT_ANY
    static constexpr float LIM = 65504.f;
    template <int LAYOUT>
    __device__ __forceinline__ void see(int row, float a, float b, float c, float d) {
        if (!VSSR_SAT_TRACK) return;
        if (LAYOUT == SAT_ANY) {
            const float mx = fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d)));
            rows |= mx > LIM ? 1u << (row & (TA - 1)) : 0u;
            return;
        }
        auto upd = [&](float m) { return fmaxf(fmaxf(fmaxf(fmaxf(m, fabsf(a)), fabsf(b)), fabsf(c)), fabsf(d)); };
        const bool upper = (row >> 4) & 1;
        if (LAYOUT == SAT_ACC) { acc_lo = upper ? acc_lo : upd(acc_lo); acc_hi = upper ? upd(acc_hi) : acc_hi; }
        else { coop_lo = upper ? coop_lo : upd(coop_lo); coop_hi = upper ? upd(coop_hi) : coop_hi; }
    }
    // a0: first atom of the tile, N: atoms in the batch (tail rows are copies of the last atom)
    __device__ __forceinline__ void commit(const ActiveView &av, int a0, int N) const {
        if (!VSSR_SAT_TRACK || !av.sat) return;
        const float worst = fmaxf(fmaxf(acc_lo, acc_hi), fmaxf(coop_lo, coop_hi));
        if (!(worst > LIM) && rows == 0u) return;
        auto raise = [&](int row) {
            const int c = av.atom_cfg[min(a0 + (row & (TA - 1)), N - 1)];
            if (av.chain(c)) atomicOr(av.sat + c, 1u);   // (a switched-off chain is not being evaluated)
        };
        // (cold path.  The thread index is laundered so that the rows are recomputed HERE: otherwise the compiler reuses row
        // indices of the kernel's prologue and keeps them alive -- in scratch memory -- for this branch alone.)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int r = tid & 15, rho = (tid & (NTHREADS - 1)) >> 5;
        if (acc_lo > LIM) raise(r);
        if (acc_hi > LIM) raise(16 + r);
        if (coop_lo > LIM) raise(rho);
        if (coop_hi > LIM) raise(16 + rho);
        for (unsigned m = rows; m; m &= m - 1u) raise(__builtin_ctz(m));
    }
};

// four consecutive columns at once (col multiple of 4): two 8-byte LDS stores
struct SatNone {
    template <int LAYOUT>
    __device__ __forceinline__ void see(int, float, float, float, float) {}
};

template <int LAYOUT = SAT_ACC, class Sat>
__device__ __forceinline__ void store_split4(const Planes &P, int row, int col, float4 v, Sat &sat) {
    sat.template see<LAYOUT>(row, v.x, v.y, v.z, v.w);
    const f32x2 a = {__builtin_amdgcn_fmed3f(v.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.y, -65504.f, 65504.f)};
    const f32x2 b = {__builtin_amdgcn_fmed3f(v.z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.w, -65504.f, 65504.f)};
    const f16x2 ha = __builtin_convertvector(a, f16x2), hb = __builtin_convertvector(b, f16x2);
    const f16x2 la = __builtin_convertvector(a - __builtin_convertvector(ha, f32x2), f16x2);
    const f16x2 lb = __builtin_convertvector(b - __builtin_convertvector(hb, f32x2), f16x2);
    *reinterpret_cast<u32x2 *>(P.h + row * P.ld + col) = (u32x2){__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
    *reinterpret_cast<u32x2 *>(P.l + row * P.ld + col) = (u32x2){__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
}

template <int LAYOUT = SAT_ACC, class Sat>
__device__ __forceinline__ void store_split4(const Planes &P, int row, int col, f32x4 v, Sat &sat) {
    store_split4<LAYOUT>(P, row, col, make_float4(v[0], v[1], v[2], v[3]), sat);
}

// Cooperative tile load: NROWS rows of F floats from global, split, into plane columns [col0, col0 + F).  rowptr(row)
// must always return a readable row (tail rows are clamped to the last atom; their results are never stored), so that
// all loads are unconditional and issued back-to-back before the first LDS store.
template <int NROWS, class RowPtr, class Sat>
__device__ __forceinline__ void load_rows_split(const Planes &P, int col0, RowPtr rowptr, Sat &sat) {
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
        store_split4<SAT_COOP>(P, idx >> 5, col0 + 4 * (idx & 31), v[it], sat);
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
template <int NROWS, class Sat>
__device__ __forceinline__ void rows_store_split(const Planes &P, int col0, const float4 (&v)[NROWS * (F / 4) / NTHREADS],
                                                 Sat &sat) {
    constexpr int NIT = NROWS * (F / 4) / NTHREADS;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = threadIdx.x + it * NTHREADS;
        store_split4<SAT_COOP>(P, idx >> 5, col0 + 4 * (idx & 31), v[it], sat);
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

__device__ __forceinline__ float gload1f(const float *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const float __attribute__((address_space(1))) *gptr;
    return *reinterpret_cast<gptr>(reinterpret_cast<uintptr_t>(p));
#else
    return *p;
#endif
}

}  // namespace vssr
#endif
