// painn_edge_mfma.hip — the neighbor-sum ("message") stage of PaiNN on gfx950, layers >= 1, forward and reverse:
// LDS-staged feature slices + the radial filter on the 16-bit matrix cores with fp32-level accuracy.
//
// Math (nff MessageBlock / DistanceEmbed):
//   w_e = (Wd rbf(d_e) + bd) fcut(d_e) = Wd_ext . rho(d_e),  rho = [sin(n pi d/rc)/d * fc]_{n=1..20} ++ [fc]
//   s_i += sum_e phi_j[b] w_e[b] ;  v_i += sum_e ( phi_j[c] w_e[c] u_e + phi_j[a] w_e[a] v_j )
//
// Decomposition: one workgroup = (chain, 16-feature slice, ensemble member).  It stages the slice of phi and v of ALL
// atoms of its chain in LDS once (exactly the compulsory HBM bytes of SURVEY §8(d): every phi/v element is read by one
// workgroup only), then walks the chain's padded CSR in steps of 16 slots per wave: 4 streams x 4 slots, the 4 streams
// being the 4 centres of a "bundle" of (nearly) equal slot count (BundleWalk below).  The filter tile
// D[16 features][16 slots] = Wd_ext . rho runs on v_mfma_f32_16x16x32_f16 (fp16 2-way split, three products packed densely
// into the K dimension: two instructions per tile): weights are the A operand (resident in registers), rho the B operand,
// read as operand-ready pieces from the quad-interleaved per-slot table that k_edge_geom (nbr.hip) writes once per
// evaluation; the slot's unit vector, 1 / d and neighbor index ride in the same records as plain 32-bit words (slot_scalars):
// no other per-slot load in the hot loops.  Lane (p, fq) owns slot p and features
// 4 fq .. 4 fq + 3: it gathers the neighbor's values from the LDS tile, forms the messages in registers and accumulates
// them; a centre is written once, after a 4-lane DPP reduction.  No atomics; the summation order per centre is the CSR
// order regardless of batching.  What bounds these kernels and what was tried: DESIGN.md section 5,
// profiles/r01/NOTES_edge_r1b.md.
// Larger chains take the residual-from-memory form (<= 405 atoms) or the 8-feature-slice instantiation (<= 787), beyond that the
// gather kernels in painn.hip; the reverse pass keeps 16-feature slices up to 557 atoms (edge_class_of / edge_bclass_of); the choice
// is per chain (edge_class_of), so a chain's results do not depend on what it is batched with.
#include "vssr_internal.h"

#ifdef EDGE_PHASE_TIMING   // debug build only (tools/gpu_edge_phase.py): wall-clock ticks (100 MHz) between marks inside the hot loops,
// accumulated per wave and added to the global counters when the wave leaves; slots [0, 8): forward kernel, [8, 16): reverse kernel
__device__ unsigned long long g_edge_phase[16];
#define EPH_INIT unsigned long long eph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, eph_t = wall_clock64();
#define EPH(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long eph_n = wall_clock64(); eph_acc[k] += eph_n - eph_t; eph_t = eph_n; __builtin_amdgcn_sched_barrier(0); }
#define EPH_FLUSH(base) if ((threadIdx.x & 63) == 0) { for (int q_ = 0; q_ < 8; ++q_) atomicAdd(&g_edge_phase[(base) + q_], eph_acc[q_]); }
extern "C" int vssr_debug_edge_phases(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_edge_phase), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_edge_phase), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define EPH_INIT
#define EPH(k)
#define EPH_FLUSH(base)
#endif

#ifndef SUB16_WAVES
#define SUB16_WAVES 12   // waves per workgroup of the 16-feature multi-pass forward form (144 registers: 3 waves per SIMD; 8 waves: +6 %)
#endif

namespace vssr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- radial filter on the matrix pipe with fp32-level accuracy -----------------------------------------------------------
// fp32 MFMA shares the FP32 datapath with the VALU (measured: worse than additive), the 16-bit MFMAs do not.  Both
// operands are split into two fp16 pieces x = h + l (22 mantissa bits) and three products are kept, Wh rl + Wl rh + Wh rh;
// the dropped Wl rl is 2^-22 relative.  Measured on the real weights against fp64 this is as accurate as a plain fp32 dot
// product (nbr.hip).  K layout (round 6): a lane quarter kq owns the 5 radial indices k = kq + 4 t; their 3 x 5 products and
// one of the three envelope / bias products (bd_h fc_l | bd_l fc_h | bd_h fc_h in quarters 0 .. 2) are exactly the 16 K entries
// the lane holds in TWO matrix instructions -- 63 of the 64 K entries carry a product.  (Rounds 1-5 issued the three products
// as three instructions with 23 of 32 K entries used each, plus a two-instruction selector tile that handed out the per-slot
// scalars from the spare entries: 11 / 20 matrix instructions per 16 slots in the forward / reverse kernel instead of 6 / 12;
// same-box A/B profiles/r06/ab_kdense.txt.)  The first B operand is unit 0 of the slot's table record as loaded; the second is
// put together from both units with two register moves (dense_b2): rho_h is stored once, the freed words carry the quarter's
// per-slot scalar as fp32 and the neighbor index.  Pieces of rho are pre-split once per evaluation (k_edge_geom, nbr.hip),
// pieces of the weights once per handle (build_wd16): nothing is split or bias-multiplied in the hot loop.  (An exact 3-way
// bf16 split with six products was used first: same accuracy, twice the matrix work and 1.5x the table bytes.)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // one MFMA operand: 8 x fp16
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Ordering rule for these kernels (found the hard way, tools/gpu_stress.py): between the first MFMA of a step and the
// first VALU consumer of the final accumulators NO load -- global or LDS -- may be issued.  Dependent MFMAs wait queued
// for their accumulator and read their A / B / C registers late; the compiler reuses dead operand / accumulator registers
// as load destinations guarded only by a fixed s_nop count, and a fast return (LDS ~50 cycles, TCP hit) lands first:
// isolated wrong filter values, run-to-run differences.  mfma_load_fence() makes the order a data dependency: the slot
// index every following address is computed from passes through the same (empty) asm as results of every accumulator
// chain, so those loads cannot be issued before the consumers, which cannot issue before the MFMAs have retired.
// tools/check_mfma_loads.py verifies the emitted ISA.
// Loads written before the MFMAs must also be ISSUED before them: a memory-clobbering asm that the MFMA operands pass
// through -- loads cannot sink below it, MFMAs cannot rise above it.
__device__ __forceinline__ void mfma_pre_fence(u32x4 &a, u32x4 &b) {
    asm volatile("; mfma_pre_fence" : "+v"(a), "+v"(b) : : "memory");
}
// Everything a step consumes from global memory is "used" here, at the top of the step, BEFORE the conditional memory
// operations of a completed centre (result stores, prefetch of the next centre) are issued.  vmcnt retires in order and
// the compiler must pick one count for both sides of a branch: with the conditional operations issued after the table
// loads it could only wait for everything (measured in the ISA: s_waitcnt vmcnt(0) right behind freshly issued stores /
// prefetches on every step).  With them issued first they are older than the next table loads and every wait is exact.
__device__ __forceinline__ void arrival_fence(u32x4 &a, u32x4 &b) {
    asm volatile("; arrival_fence" : "+v"(a), "+v"(b) : : "memory");
}
__device__ __forceinline__ void arrival_fence(u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d, float &g) {
    asm volatile("; arrival_fence" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(g) : : "memory");
}
// Explicit register copy.  The table buffers are refilled (for the step after next) in the middle of the step that consumes
// them; small values that stay live to the end of the step are first moved out with a real v_mov, so that the buffer is
// dead when its refill is issued and the load can land in the same registers (otherwise the loop-carried buffer needs a
// copy of the freshly loaded value at the loop end -- and a wait for it).
__device__ __forceinline__ float take(float src) {
    float d;
    asm("v_mov_b32 %0, %1" : "=v"(d) : "v"(src));
    return d;
}
__device__ __forceinline__ unsigned take_u(unsigned src) {
    unsigned d;
    asm("v_mov_b32 %0, %1" : "=v"(d) : "v"(src));
    return d;
}
__device__ __forceinline__ void mfma_load_fence(int &index, float &a, float &b, float &c) {
    asm volatile("; mfma_load_fence" : "+v"(index), "+v"(a), "+v"(b), "+v"(c));
}
__device__ __forceinline__ void mfma_load_fence(int &index, float &a, float &b, float &c, float &d) {
    asm volatile("; mfma_load_fence" : "+v"(index), "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// The weight pointer comes out of the ModelW table in memory, so the compiler treats it as FLAT; a flat load ticks vmcnt AND lgkmcnt,
// and while one is pending every wait the compiler inserts is a full drain.  The forward kernel loads its weights right in front of
// the hot loop: the pending flat loads reached the loop header, whose wait became vmcnt(0) for EVERY iteration (the merge of the
// pre-header and the back edge) -- the two-step table prefetch was drained every second step (found with the phase clocks of
// tools/gpu_edge_phase.py: 43 % of a forward wave's time sat in that wait).  Through an explicit global pointer the loads are
// global_load, the header wait is the exact vmcnt(2).
__device__ __forceinline__ u32x4 gload_u32x4(const u32x4 *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const u32x4 __attribute__((address_space(1))) *gptr;
    return *reinterpret_cast<gptr>(reinterpret_cast<uintptr_t>(p));
#else
    return *p;
#endif
}
// D += W . rho for one 16 x 16 tile
__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
// Filter tiles of one step.  The two products of a tile form a dependent accumulator chain, and dependent MFMAs
// issued back to back stall INSIDE the matrix pipe: everything queued behind them reads its source registers much later
// than the compiler's wait-state model assumes, while the compiler already reuses those registers (observed: VALU and
// LDS writes into MFMA sources a few wait states after issue -> wrong forces).  So the products are issued in two rounds
// across all NT tiles: with three or more tiles the producer of an accumulator is at least three matrix instructions back and
// has retired when its consumer reaches the pipe; with one or two tiles every product starts its own chain (zero accumulator)
// and the two partial sums are added on the vector unit.  The scheduling barriers pin the issue order (without them the
// scheduler re-serialises the chains to save registers).  Unit 0 of the slot's table record is the first B operand as loaded,
// dense_b2 puts the second together from both units.
__device__ __forceinline__ u32x4 dense_b2(const u32x4 &u0, const u32x4 &u1) { return (u32x4){u1[0], u1[1], u0[3], u1[0]}; }
template <int NT>
__device__ __forceinline__ void filter_tiles_dense(const u32x4 (*const (&w)[NT])[2], const u32x4 *const (&b1)[NT], const u32x4 *const (&b2)[NT],
                                                   f32x4 (&acc)[NT]) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if constexpr (NT >= 3) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = mfma_f16((*w[t])[0], *b1[t], z);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = mfma_f16((*w[t])[1], *b2[t], acc[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the second operands stay live past the block: otherwise the register allocator places the destination of a chain's last
        // product over its dying B operand while the accumulator input is another register (lint rule 4, tools/check_mfma_loads.py)
#pragma unroll
        for (int t = 0; t < NT; ++t) asm volatile("; mfma operands live" : : "v"(*b2[t]));
    } else {
        f32x4 part[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = mfma_f16((*w[t])[0], *b1[t], z);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            part[t] = mfma_f16((*w[t])[1], *b2[t], z);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) asm volatile("; mfma operands live" : : "v"(*b1[t]), "v"(*b2[t]));
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] += part[t];
    }
}
// The per-slot scalars {u_x, u_y, u_z, 1 / d}: lane (slot, quarter r) loaded scalar r as a plain fp32 word with its table record;
// an all-gather over the four 16-lane rows on the gfx950 row swaps hands all four to every lane of the slot (three swaps):
// swap32 (x, x) -> (x0 x1 x0 x1), (x2 x3 x2 x3) by rows; swap16 of each with itself -> x0, x1 resp. x2, x3 in every row.
__device__ __forceinline__ void slot_scalars(unsigned x, float &s0, float &s1, float &s2, float &s3) {
    const auto a = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    const auto lo = __builtin_amdgcn_permlane16_swap(a[0], a[0], false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(a[1], a[1], false, false);
    s0 = __uint_as_float(lo[0]); s1 = __uint_as_float(lo[1]); s2 = __uint_as_float(hi[0]); s3 = __uint_as_float(hi[1]);
}

// host: weight pieces in A-operand order, dst[row][kq][operand 1, 2][4 dwords]
void build_wd16(const float *Wd, const float *bd, unsigned *dst) {
    auto split2 = [](float x, unsigned (&p)[2]) {
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        unsigned short hb, lb;
        memcpy(&hb, &h, 2);
        memcpy(&lb, &l, 2);
        p[0] = hb; p[1] = lb;
    };
    // the partners of the two B operands the edge kernels form from a table record (nbr.hip write_f16_record),
    //   B1 = [l0 l1 l2 l3 l4 env h0 h1]   A1 = [Wh0 Wh1 Wh2 Wh3 Wh4 bias Wl0 Wl1]   (bias / env: bd_h fc_l | bd_l fc_h | bd_h fc_h | 0 0
    //   B2 = [h2 h3 h4 h4 h0 h1 h2 h3]    A2 = [Wl2 Wl3 Wl4 Wh4 Wh0 Wh1 Wh2 Wh3]      in quarters 0 .. 3)
    // index t of quarter kq = radial index kq + 4 t: per quarter Wh.rho_l + Wl.rho_h + Wh.rho_h of its five radial functions.
    for (int row = 0; row < F3; ++row)
        for (int kq = 0; kq < 4; ++kq) {
            unsigned wh[5], wl[5], bp[2];
            for (int t = 0; t < 5; ++t) {
                unsigned p2[2];
                split2(Wd[(size_t)row * 20 + kq + 4 * t], p2);
                wh[t] = p2[0]; wl[t] = p2[1];
            }
            split2(bd[row], bp);
            const unsigned bias = kq == 1 ? bp[1] : kq == 3 ? 0u : bp[0];
            const unsigned a1[8] = {wh[0], wh[1], wh[2], wh[3], wh[4], bias, wl[0], wl[1]};
            const unsigned a2[8] = {wl[2], wl[3], wl[4], wh[4], wh[0], wh[1], wh[2], wh[3]};
            unsigned *o = dst + ((size_t)row * 4 + kq) * 8;
            for (int q = 0; q < 4; ++q) {
                o[q] = a1[2 * q] | (a1[2 * q + 1] << 16);
                o[4 + q] = a2[2 * q] | (a2[2 * q + 1] << 16);
            }
        }
}

// Feature slices come in two widths.  NF = features per lane: 4 -> a 16-feature slice (the default: 468 B of LDS per atom in
// the forward kernel, chains up to 350 atoms; 404 B and 405 atoms without the staged residual), 2 -> an 8-feature slice
// (208 B per atom, chains up to 787 atoms: 4 x 4
// slabs at full coverage, multi-atom adsorbate groups).  The 8-feature kernels keep the 16 x 16 matrix tile full by packing
// two filter sections into one tile: tile row i = 4 q + j carries feature 2 q + (j & 1) of section 2 T + (j >> 1), so that lane
// (slot, fq) -- which owns tile rows 4 fq .. 4 fq + 3 -- still finds the a, b and c filter values of ITS features in its own
// accumulator registers (tile 0 = [a | b], tile 1 = [c | -]).  Every chain takes the path its own atom count selects
// (vssr_api.hip classify_chains), whatever it is batched with.
#ifndef STAGE_BATCH
#define STAGE_BATCH 4   // slice staging: loads in flight per thread before the first LDS store
#endif
// Forward workgroup width: 16 waves per CU = 4 per SIMD (<= 128 VGPRs) in every case -- one workgroup of 16 waves when the
// chain's slice fills the LDS (measured on 260-atom chains: 512 threads -> 2.54, 768 -> 2.17, 1024 -> 2.03 ms / step), two of 8
// or four of 4 waves when two / four slices fit (small chains: a 74-atom chain has 19 bundles, which 16 waves share badly).

// the NF features of a lane as one register vector (NF = 1: a struct that indexes like one)
template <int N> struct FeatVec { typedef float type __attribute__((ext_vector_type(N))); };
struct FeatVec1 {
    float v;
    __device__ __forceinline__ float operator[](int) const { return v; }
};
template <> struct FeatVec<1> { typedef FeatVec1 type; };

// LDS slice layout: tile[atom][feature f][NSEG] with NSEG = {a, b, c, v_x, v_y, v_z}: the values a lane needs for its NF
// features of one neighbor are NF * 24 contiguous bytes.  (Layer 0 never comes here: it is either
// factorised by species, painn_l0.hip, or -- more than 8 species / VSSR_L0_FACTORISE=0 -- runs the gather kernels.)
template <int NF>
struct EdgeGeo {
    static_assert(NF == 4 || NF == 2 || NF == 1, "16-, 8- or 4-feature slices");
    static constexpr int FS = 4 * NF;                // features per slice
    static constexpr int NSLICE = F / FS;            // 8 or 16
    static constexpr int NSEC = 3;                   // filter sections a, b, c
    static constexpr int NT = NF == 4 ? 3 : NF == 2 ? 2 : 1;   // filter tiles per table (rho; the reverse kernel adds as many for d rho)
    static constexpr int NSEG = 6;                   // values staged per (atom, feature): phi a, b, c and v x, y, z
    static constexpr int ROW = NSEG * FS + 4;        // forward LDS row stride (floats); rows stay 16-B aligned
    static constexpr int ROWB = FS * 4 + 4;          // reverse LDS row: [feature][sbar, vbar_x, vbar_y, vbar_z] + pad
    // filter section and slice feature carried by row i of tile T (-1: the row is empty)
    // (NF = 1, a 4-feature slice: ONE tile, row 4 q + j = feature q of section j, j = 3 empty -- lane (slot, fq) finds a, b, c of its
    // feature in accumulator registers 0, 1, 2)
    __host__ __device__ static constexpr int row_section(int T, int i) {
        return NF == 4 ? T : NF == 1 ? ((i & 3) < 3 ? (i & 3) : -1) : (2 * T + ((i & 3) >> 1) < 3 ? 2 * T + ((i & 3) >> 1) : -1);
    }
    __host__ __device__ static constexpr int row_feature(int i) { return NF == 4 ? i : NF == 1 ? (i >> 2) : 2 * (i >> 2) + (i & 1); }
};
// accumulator register r' of tile T that holds the filter of section sec for the lane's feature r (r < NF)
template <int NF> __device__ __forceinline__ constexpr int sec_tile(int sec) { return NF == 4 ? sec : NF == 1 ? 0 : sec >> 1; }
template <int NF> __device__ __forceinline__ constexpr int sec_reg(int sec, int r) { return NF == 4 ? r : NF == 1 ? sec : 2 * (sec & 1) + r; }

// LDS carve-up: tile [max_atoms][ROW] | SLDS: s slice [max_atoms][FS].  The scalar residual of a centre (its own s slice, which
// no gather ever touches) either sits in LDS next to the tile (SLDS = true: 468 B per atom with 16-feature slices, <= 350
// atoms; the fastest form: 2.09 vs 2.15 ms / step on the 260-atom workload) or is requested from memory when the wave
// switches to the centre's bundle, one bundle ahead of its use (SLDS = false: 404 B per atom, <= 405 atoms -- 3 x 2 slabs with
// adsorbates stay on the 16-feature kernels: 18.8 instead of 22.6 ms / step at 368 .. 392 atoms; 8-feature slices: 208 B, <= 787).
template <int NF, bool SLDS>
size_t edge_fwd_lds_bytes_t(int max_atoms) {
    return sizeof(float) * ((size_t)max_atoms * (EdgeGeo<NF>::ROW + (SLDS ? EdgeGeo<NF>::FS : 0)) + 4);
}

// Gather rows of the LDS tiles: the row offset is formed with the 24-bit multiply-add (full rate) on a 32-bit LDS address.  Written
// on a generic `float *` the same arithmetic is 64-bit -- v_mad_u64_u32, a quarter-rate instruction, once per step and lane.
typedef float f32x2v __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) const float lds_cfloat;
__device__ __forceinline__ lds_cfloat *lds_base(const float *p) { return (lds_cfloat *)p; }
__device__ __forceinline__ lds_cfloat *lds_row(lds_cfloat *base, int row, int row_floats) {
    return base + __umul24((unsigned)row, (unsigned)row_floats);
}
__device__ __forceinline__ f32x4 lds_read4(lds_cfloat *p) { return *reinterpret_cast<__attribute__((address_space(3))) const f32x4 *>(p); }
__device__ __forceinline__ f32x2v lds_read2(lds_cfloat *p) { return *reinterpret_cast<__attribute__((address_space(3))) const f32x2v *>(p); }
#else   // (host pass of the single-source compile: address spaces do not exist there)
typedef const float lds_cfloat;
__host__ __device__ inline lds_cfloat *lds_base(const float *p) { return p; }
__host__ __device__ inline lds_cfloat *lds_row(lds_cfloat *base, int row, int row_floats) { return base + row * row_floats; }
__host__ __device__ inline f32x4 lds_read4(lds_cfloat *p) { return *reinterpret_cast<const f32x4 *>(p); }
__host__ __device__ inline f32x2v lds_read2(lds_cfloat *p) { return *reinterpret_cast<const f32x2v *>(p); }
#endif

// sum over the 4 lanes of a quad (lanes 4q..4q+3), result in every lane: two DPP quad_perm adds
__device__ __forceinline__ float quad_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    return x;
}
// (a hand-written v_add_f32_dpp variant was tried and dropped: the compiler's wait-count / hazard bookkeeping does not
// look inside inline asm, and the kernel produced run-to-run differences while table loads were in flight)
// The sums are consumed only by the quad's first lane inside a branch; without the (empty) asm the compiler sinks the
// second add into that branch and keeps a separate v_mov_b32_dpp outside (a DPP read of a lane the branch disabled
// returns 0): 3 instructions per value instead of 2.  The asm pins the complete sum in front of the branch.
template <int NF>
__device__ __forceinline__ void quad_sum_n(float (&x)[NF]) {
#pragma unroll
    for (int r = 0; r < NF; ++r) x[r] = quad_sum(x[r]);
    if constexpr (NF == 4) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
    else if constexpr (NF == 2) asm volatile("" : "+v"(x[0]), "+v"(x[1]));
    else asm volatile("" : "+v"(x[0]));
}
// the NF values of a lane to NF consecutive floats (one 16- or 8-byte store)
template <int NF>
__device__ __forceinline__ void store_feat(float *dst, const float (&v)[NF]) {
    if constexpr (NF == 4) *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    else if constexpr (NF == 2) *reinterpret_cast<float2 *>(dst) = make_float2(v[0], v[1]);
    else *dst = v[0];
}

// sum over the four 16-lane rows (lanes l, l^16, l^32, l^48), result in every lane: the gfx950 row-swap instructions
// (VALU) instead of two dependent ds_bpermute round trips through the LDS
__device__ __forceinline__ float xrow_sum(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float y = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
    const unsigned v = __float_as_uint(y);
    const auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
}

// ---- work decomposition of both kernels: bundles ---------------------------------------------------------------------------
// The neighbor build sorts the centres of every chain by padded slot count (G.bundle, nbr.hip k_bundle_sort).  Four
// consecutive entries form a bundle; a wave walks one bundle at a time, its 4 CSR streams (lanes p >> 2) taking one centre
// each, 4 slots per stream and step.  Centres of a bundle have (nearly) the same slot count, so all four complete in the
// same step (L = slot count of the longest / 4 steps; a shorter centre runs its last steps on the all-zero table entry):
// the completion code -- cross-lane reduction, residual, stores, accumulator reset -- runs once per bundle with all lanes
// active and behind a wave-uniform branch.  (Before: every stream walked its own run of centres, each completion ran
// with a quarter of the lanes and made the compiler copy every accumulator on every step.)  Bundles are dealt to the NW
// waves of the workgroup in snake order (w, 2 NW - 1 - w, 2 NW + w, ...), which balances a monotone sequence.
// Every centre owns >= 8 slots (nbr.hip), i.e. L >= 2: a table prefetch two steps ahead crosses at most one bundle
// boundary.  Entries of upcoming bundles are loaded two bundles ahead (one unconditional load per bundle).
template <int NW>
struct BundleWalk {
    const int4 *tab;   // entries of this chain
    int Nc, wave, st;  // st: stream (lane quad) inside the wave
    int nj;            // bundles of this wave
    int j, t, Lc, Ln;  // current bundle (index into the wave's list), step inside it, steps of current / next bundle
    int4 cur, nxt, pend;   // pend: raw entry of the bundle after next, in flight since the previous switch
    bool pend_ok;          // ... and whether this stream has a centre in it

    __device__ __forceinline__ int bundle_of(int jj) const { return jj * NW + ((jj & 1) ? NW - 1 - wave : wave); }
    // always a load (clamped address): the count of memory operations is path-independent.  The value is only looked at
    // one bundle later (select()), so the load never stalls the switch that issues it.
    __device__ __forceinline__ int4 load(int jj, bool &ok) const {
        const int idx = 4 * bundle_of(jj) + st;
        ok = jj < nj && idx < Nc;
        return tab[max(min(idx, Nc - 1), 0)];
    }
    __device__ __forceinline__ static int4 select(const int4 &v, bool ok) { return ok ? v : make_int4(-1, 0, 0, 0); }
    __device__ __forceinline__ static int steps(const int4 &q) { return max(__builtin_amdgcn_readfirstlane(q.z) >> 2, 2); }
    __device__ __forceinline__ void init(const int4 *table, int n, int w, int stream) {
        tab = table; Nc = n; wave = w; st = stream;
#ifdef ABL_NO_TAIL   // ablation: the bundles beyond the last full round of the workgroup's waves are dropped (results wrong)
        const int NB0 = (n + 3) >> 2, NB = NB0 >= NW ? NB0 - NB0 % NW : NB0, full = NB / NW, rem = NB - full * NW;
#else
        const int NB = (n + 3) >> 2, full = NB / NW, rem = NB - full * NW;
#endif
        nj = full + ((((full & 1) ? NW - 1 - w : w) < rem) ? 1 : 0);
        j = 0; t = 0;
        bool ok0, ok1;
        const int4 r0 = load(0, ok0), r1 = load(1, ok1);
        pend = load(2, pend_ok);
        cur = select(r0, ok0); nxt = select(r1, ok1);
        // stream 0 of a bundle holds its longest centre (descending order); lane 0 of the wave belongs to stream 0
        Lc = steps(cur); Ln = steps(nxt);
    }
    // first slot of this stream's quad `ahead` steps from now (ahead <= 2) and whether it holds real slots
    __device__ __forceinline__ int quad_ahead(int ahead, bool &valid) const {
        const int tt = t + ahead;
        const bool in_cur = tt < Lc;   // wave-uniform
        const int4 &q = in_cur ? cur : nxt;
        const int ts = in_cur ? tt : tt - Lc;
        valid = 4 * ts < q.z;
        return q.y + 4 * ts;
    }
    // bundle complete: move on.  Past the wave's last bundle the entries are empty (no centre, no real slot, 2 steps): the
    // hot loops test `j < nj` only between iterations, so a finished wave runs at most a few empty steps instead of
    // leaving the loop from its middle (an exit there makes the compiler's memory waits conservative on every step).
    __device__ __forceinline__ void advance() {
        ++j;
        cur = nxt; nxt = select(pend, pend_ok);
        pend = load(j + 2, pend_ok);
        t = 0; Lc = Ln; Ln = steps(nxt);
    }
};

// Lane roles ("slot-major"): p = lane & 15 is the slot position inside the step's 16-slot tile, stream = p >> 2
// (4 independent CSR streams per wave, 4 slots each per step), e = p & 3 the slot inside the quad, and
// fq = lane >> 4 selects features 4 fq .. 4 fq + 3 of the slice.  With the WEIGHTS as MFMA A operand
// (A[i = feature][k]) and rho as B (B[k][j = slot]) the filter tile D[feature][slot] puts the 4 feature values of
// slot p into the 4 accumulator registers of lane (p, fq): the lane that loaded rho for slot p (its k-quarter is
// fq) also owns that slot's messages, so one table address serves both and no cross-lane traffic is needed.
// Chains of this launch: `list` (chain indices of one slice-width class, built at upload) or, list == nullptr, all chains.
// XCD-aware 1-D grid: workgroup id -> XCD id % 8 (observed dispatch rule).  All (slice, model) workgroups of one chain get
// consecutive ids on ONE XCD, so the chain's rho / record tables (~1.3 MB) and its phi / v rows are fetched from HBM once and
// then served by that XCD's L2.  Returns the chain, or -1 for a workgroup without one; t = (slice, model) index.
__device__ __forceinline__ int chain_of_workgroup(const GraphView &G, const int *__restrict__ list, int n_list, int per_chain, int &t) {
    const int wg = blockIdx.x, xcd = wg & 7, rest = wg >> 3;
    t = rest % per_chain;
    const int k = (rest / per_chain) * 8 + xcd;
    if (k >= n_list) return -1;
    const int b = list ? list[k] : k;
    return G.act.chain(b) ? b : -1;
}

// SUB (chains whose slice does not fit LDS at this width: one launch per `pass`): the chain's atoms are cut into P = ceil(n / chunk_max)
// equal ranges (sub_passes / sub_chunk below); the workgroup of pass p stages only the neighbors [lo, lo + ns) of range p plus an
// all-zero row, and walks, for every centre, the window of its (neighbor-sorted) row that holds those neighbors (`bundle_tab`:
// per-pass bundle tables, nbr.hip k_bundle_sort_sub).  Windows are cut at quad boundaries, so a window may contain slots of another
// range: their neighbor index is redirected to the zero row and they contribute exactly nothing.  Pass 0 walks every centre and adds
// the residual (s_in, v_in of the centre, from memory) like the SLDS = false form; a later pass walks the centres that have
// neighbors in its range (`n_entries`: the head of its length-sorted table) and adds to what the passes before it wrote (s_msg,
// v_msg): same code, other base pointers.  Why wider slices in several passes beat narrower slices in one: every (slice, model)
// workgroup streams the chain's per-slot tables once, so the table stream per atom grows with the number of slices -- the bound of
// the 8- and 4-feature kernels on large chains (profiles/r05/NOTES_large_chains.md).
// Output pointers: __restrict__ (the single-pass forms never read what they write), plain in the multi-pass forms, whose later passes
// add to the output of the earlier ones.  (Dropping the qualifier everywhere cost the single-pass reverse kernel 5 %: other schedule.)
template <bool ALIASED> struct EdgeOut { typedef float *__restrict__ ptr; };
template <> struct EdgeOut<true> { typedef float *ptr; };

// (registers: 4 waves per SIMD = 128 everywhere, except the 16-feature multi-pass form, whose extra residual registers need the
//  256 of an 8-wave workgroup)
template <int NF, bool SLDS, int WAVES, bool SUB = false>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu((SUB && NF == 4) ? WAVES / 4 : 4, (SUB && NF == 4) ? WAVES / 4 : 4)))
k_edge_fwd_mfma(int N, int l, const ModelW *__restrict__ MW, GraphView G, const int *__restrict__ counters,
                int zero_slot, int n_models, int max_atoms, const int *__restrict__ list, int n_list,
                const float *__restrict__ s_in, const float *__restrict__ v_in, const float *__restrict__ phi,
                typename EdgeOut<SUB>::ptr s_msg, typename EdgeOut<SUB>::ptr v_msg, const int4 *__restrict__ bundle_tab,
                const int *__restrict__ n_entries_tab, int pass, int chunk_max) {
    using LY = EdgeGeo<NF>;
    constexpr int FS = LY::FS, NSLICE = LY::NSLICE, NT = LY::NT, EDGE_THREADS = 64 * WAVES;
    static_assert(!SUB || !SLDS, "the sub-range form takes its residuals from memory");
    extern __shared__ __attribute__((aligned(16))) float tile[];
    if (counters[2]) return;
    int t;
    const int b = chain_of_workgroup(G, list, n_list, NSLICE * n_models, t);
    if (b < 0) return;
    const int fs = t % NSLICE, m = t / NSLICE;
    const int a0 = G.cfg_start[b], Nc = G.cfg_start[b + 1] - a0;
    const size_t mN = (size_t)m * N;
    const int tid = threadIdx.x;
    // staged neighbors: all atoms of the chain, or (SUB) range `pass` of its sub_passes(Nc) ranges
    int lo = 0, ns = Nc;
    if constexpr (SUB) {
        if (pass >= sub_passes(Nc, chunk_max)) return;   // (uniform; this chain needs fewer passes than the launch's largest)
        const int chunk = sub_chunk(Nc, chunk_max);
        lo = pass * chunk;
        ns = min(chunk, Nc - lo);
    }
    const float *res_s = (SUB && pass) ? s_msg : s_in, *res_v = (SUB && pass) ? v_msg : v_in;   // what a centre's sums are added to

    // ---- stage the chain's feature slice: tile[atom][f][seg] ---------------------------------------------------------
    // NSEG * FS / 4 float4 per atom; loads are issued in batches of 4 per thread before any LDS store so that
    // the L2 / HBM round trips overlap (one workgroup per CU: nothing else hides them)
    float *s_tile = tile + (size_t)max_atoms * LY::ROW;                      // [atom][FS] (SLDS only)
    {
        constexpr int Q4 = FS / 4;                          // float4 per slice segment
        constexpr int PER_ATOM = (LY::NSEG + (SLDS ? 1 : 0)) * Q4;   // float4 per atom
        const int total = ns * PER_ATOM;
        if constexpr (SUB) {   // the all-zero row behind the staged ones: where slots of the other half point
            for (int k = tid; k < LY::ROW; k += EDGE_THREADS) tile[(size_t)ns * LY::ROW + k] = 0.f;
        }
        auto src_of = [&](int idx) -> const float * {
            int atom = idx / PER_ATOM, rem = idx - atom * PER_ATOM, seg = rem / Q4, q4 = rem % Q4;
            const size_t ga = mN + a0 + lo + atom;
            if (seg == LY::NSEG) return s_in + ga * F + fs * FS + q4 * 4;                  // s slice
            if (seg < 3) return phi + ga * F3 + seg * F + fs * FS + q4 * 4;                // sections a, b, c
            return v_in + (ga * 3 + (seg - 3)) * F + fs * FS + q4 * 4;                     // v_x, v_y, v_z
        };
        auto put = [&](int idx, const float4 &val) {
            int atom = idx / PER_ATOM, rem = idx - atom * PER_ATOM, seg = rem / Q4, q4 = rem % Q4;
            if (seg == LY::NSEG) {
                *reinterpret_cast<float4 *>(s_tile + atom * FS + q4 * 4) = val;
            } else {
                float *dst = tile + atom * LY::ROW + (q4 * 4) * LY::NSEG + seg;
                dst[0] = val.x; dst[LY::NSEG] = val.y; dst[2 * LY::NSEG] = val.z; dst[3 * LY::NSEG] = val.w;
            }
        };
        for (int base = tid; base < total; base += STAGE_BATCH * EDGE_THREADS) {
            float4 v4[STAGE_BATCH];
#pragma unroll
            for (int u = 0; u < STAGE_BATCH; ++u) {
                const int idx = min(base + u * EDGE_THREADS, total - 1);
                v4[u] = *reinterpret_cast<const float4 *>(src_of(idx));
            }
#pragma unroll
            for (int u = 0; u < STAGE_BATCH; ++u)
                if (base + u * EDGE_THREADS < total) put(base + u * EDGE_THREADS, v4[u]);
        }
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, p = lane & 15, fq = lane >> 4, e = p & 3;
    // ---- A operand: fp16 pieces (h, l) of the filter weights carried by tile row p, quarter fq: 2 x 16 B per tile, bias
    // column included (build_wd16); an empty row of the packed 8-feature tiles is all zero
    const LayerW &W = MW[m].layer[l];
    u32x4 wA[NT][2];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        const int sec = LY::row_section(T, p);
        const int row = max(sec, 0) * F + fs * FS + LY::row_feature(p);
        const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(W.wd16) + ((size_t)row * 4 + fq) * 2;
#pragma unroll
        for (int i3 = 0; i3 < 2; ++i3) wA[T][i3] = sec < 0 ? (u32x4){0u, 0u, 0u, 0u} : gload_u32x4(wsrc + i3);
    }

    // ---- work list: bundles of 4 centres of (nearly) equal slot count, see BundleWalk --------------------------------
    BundleWalk<EDGE_THREADS / 64> bw;
    // (SUB, pass >= 1: only the centres with neighbors in that range -- the head of the length-sorted table; the others keep what
    //  the earlier passes wrote)
    int n_entries = Nc;   // (an extra early exit here changes the compiler's memory-wait merge at the loop header: measured +5 % on the
                          //  single-pass reverse kernel; a pass without entries leaves through `bw.nj == 0` like an empty chain)
    if constexpr (SUB) {
        if (pass) n_entries = n_entries_tab[(size_t)(pass - 1) * G.n_cfg + b];
    }
    bw.init((SUB ? bundle_tab : G.bundle) + a0, n_entries, __builtin_amdgcn_readfirstlane(wave), p >> 2);
    if (bw.nj == 0) return;   // (no barrier below)
#ifdef ABL_STAGE_ONLY   // ablation: staging + prologue only (results wrong)
    if (bw.nj > 0) return;
#endif
    float ds[NF], dvx[NF], dvy[NF], dvz[NF];
#pragma unroll
    for (int r = 0; r < NF; ++r) { ds[r] = 0.f; dvx[r] = 0.f; dvy[r] = 0.f; dvz[r] = 0.f; }
    const int fcol = fs * FS + NF * fq;            // first of this lane's NF global feature columns
    // scalar residual of this stream's centre: requested when the wave switches to a bundle (for the bundle after it), used
    // when that bundle completes -- always a load (clamped row for streams without a centre), like the bundle entries
    typedef typename FeatVec<NF>::type fres;
    fres sres_cur, sres_nxt, vres_cur[SUB ? 3 : 1];   // (SUB: the vector residual is requested when its bundle STARTS -- one set of
                                                       //  registers; a window is several steps long, enough for an L2 round trip)
    auto load_residual = [&](int cc) {
        return *reinterpret_cast<const fres *>(res_s + (mN + a0 + min(max(cc, 0), Nc - 1)) * F + fcol);
    };
    auto load_residual_v = [&](int cc, int x) {   // (SUB: the centre's own v row may not be among the staged ones)
        return *reinterpret_cast<const fres *>(res_v + ((mN + a0 + min(max(cc, 0), Nc - 1)) * 3 + x) * F + fcol);
    };
    if (!SLDS) { sres_cur = load_residual(bw.cur.x); sres_nxt = load_residual(bw.nxt.x); }
    if constexpr (SUB) {
#pragma unroll
        for (int x = 0; x < 3; ++x) vres_cur[x] = load_residual_v(bw.cur.x, x);
    }

    // quad-interleaved table (nbr.hip f16_unit): unit = quad * 32 + piece * 16 + fq * 4 + e -> the 4 slot lanes of a quad read
    // 64 contiguous bytes; exhausted streams read the reserved all-zero quad (filter = 0)
    const u32x4 *rho_lane = reinterpret_cast<const u32x4 *>(G.rho16) + fq * 4 + e;
    const int zero_quad = (zero_slot + 1) / 4 - 1;   // zero_slot = capacity - 1; the last complete quad is all zero

    // Two table buffers (rho pieces; they also carry the unit vector and the neighbor id of the slot), one per step parity:
    // buffer ph is consumed by the step of parity ph and refilled right after that step's MFMAs for the step after next,
    // so a table load has ~1.6 steps to arrive (L2 / HBM latency is of the order of one step).
    u32x4 rq[2][2];
    auto fetch = [&](int buf, int quad_first_slot, bool valid) {   // quad_first_slot: first slot of the stream's quad
#ifdef ABL_TABLE_L2   // ablation: every table read hits a 256 KB window (results wrong)
        const u32x4 *rp = rho_lane + (size_t)(valid ? ((quad_first_slot >> 2) & 511) : zero_quad) * 32;
#else
        const u32x4 *rp = rho_lane + (size_t)(unsigned)(valid ? (quad_first_slot >> 2) : zero_quad) * 32;   // (unsigned: no sign extension, one shift-add)
#endif
        rq[buf][0] = rp[0]; rq[buf][1] = rp[16];
    };
    {
        bool v0, v1;
        const int q0 = bw.quad_ahead(0, v0), q1 = bw.quad_ahead(1, v1);
        fetch(0, q0, v0);
        // issue order = consumption order.  Without the barrier the scheduler hoisted the second buffer's loads above the first's in
        // the forward kernel; seen from the loop header the first buffer was then the YOUNGEST pending load, its wait vmcnt(0), and
        // -- the header wait being the merge of pre-header and back edge -- every iteration drained the whole prefetch queue.
        __builtin_amdgcn_sched_barrier(0);
        fetch(1, q1, v1);
    }

    // The 4 centres of a bundle complete in the same step: reduce the 4 slot lanes of every quad, add the residual from
    // the staged slices (no global loads), one vector store per row; all lanes take part.
    auto flush_bundle = [&]() {
        quad_sum_n<NF>(ds); quad_sum_n<NF>(dvx); quad_sum_n<NF>(dvy); quad_sum_n<NF>(dvz);
        const int c = bw.cur.x;
        if (e == 0 && c >= 0) {
            const size_t ga = mN + a0 + c;
            float so[NF], xo[NF], yo[NF], zo[NF];
            const float *sr = s_tile + c * FS + NF * fq;
            const float *vc = tile + c * LY::ROW + (NF * fq) * LY::NSEG;
#pragma unroll
            for (int r = 0; r < NF; ++r) {
                so[r] = ds[r] + (SLDS ? sr[r] : sres_cur[r]);
                xo[r] = dvx[r] + (SUB ? vres_cur[0][r] : vc[r * LY::NSEG + 3]);
                yo[r] = dvy[r] + (SUB ? vres_cur[SUB ? 1 : 0][r] : vc[r * LY::NSEG + 4]);
                zo[r] = dvz[r] + (SUB ? vres_cur[SUB ? 2 : 0][r] : vc[r * LY::NSEG + 5]);
            }
            store_feat<NF>(s_msg + ga * F + fcol, so);
            store_feat<NF>(v_msg + (ga * 3 + 0) * F + fcol, xo);
            store_feat<NF>(v_msg + (ga * 3 + 1) * F + fcol, yo);
            store_feat<NF>(v_msg + (ga * 3 + 2) * F + fcol, zo);
        }
#pragma unroll
        for (int r = 0; r < NF; ++r) { ds[r] = 0.f; dvx[r] = 0.f; dvy[r] = 0.f; dvz[r] = 0.f; }
    };

    lds_cfloat *trow = lds_base(tile) + (NF * fq) * LY::NSEG;
    EPH_INIT
    EPH(0)   // (the prologue is not timed: the clock starts here)
    while (bw.j < bw.nj) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {   // two steps per iteration: one table buffer per step parity
            arrival_fence(rq[ph][0], rq[ph][1]);
            EPH(1)   // wait for this step's table entries
#ifdef ABL_LDS_BCAST   // ablation: every gather reads row 0 or 1 (no LDS bank conflicts; results wrong)
            const int jn = (int)(rq[ph][0][3] >> 16) & 1;
#else
            int jn = (int)rq[ph][1][3];           // chain-local neighbor of this lane's slot (last word of the record's unit 1)
            if constexpr (SUB) {   // staged row of the neighbor, or the zero row for a neighbor of the other half
                const unsigned jl = (unsigned)(jn - lo);
                jn = jl < (unsigned)ns ? (int)jl : ns;
            }
#endif
            if (bw.t == bw.Lc) {   // wave-uniform: the bundle is complete
                flush_bundle();
                bw.advance();
                if (!SLDS) {
                    sres_cur = sres_nxt;
                    __builtin_amdgcn_sched_barrier(0);   // the old value leaves its registers before the load that refills them
                    sres_nxt = load_residual(bw.nxt.x);
                    if constexpr (SUB) {
#pragma unroll
                        for (int x = 0; x < 3; ++x) vres_cur[x] = load_residual_v(bw.cur.x, x);
                    }
                }
                EPH(2)   // bundle completion
            }
            // gather this slot's neighbor row: NF features x NSEG values, contiguous in LDS
            float tv[NF * LY::NSEG];
            {
                if constexpr ((NF * LY::NSEG) % 4 == 0) {
                    lds_cfloat *row = lds_row(trow, jn, LY::ROW);
#pragma unroll
                    for (int q = 0; q < NF * LY::NSEG / 4; ++q) {
                        const f32x4 t4 = lds_read4(row + 4 * q);
                        tv[4 * q] = t4[0]; tv[4 * q + 1] = t4[1]; tv[4 * q + 2] = t4[2]; tv[4 * q + 3] = t4[3];
                    }
                } else {   // 4-feature slices: the lane's six values start 24 fq bytes into the row (8-byte aligned)
                    lds_cfloat *row = lds_row(trow, jn, LY::ROW);
#pragma unroll
                    for (int q = 0; q < NF * LY::NSEG / 2; ++q) {
                        const f32x2v t2 = lds_read2(row + 2 * q);
                        tv[2 * q] = t2[0]; tv[2 * q + 1] = t2[1];
                    }
                }
            }
            // unit vector of this lane's slot: the record's scalar words, all-gathered over the slot's four lanes (slot_scalars)
            float ux, uy, uz, invd_unused;
            slot_scalars(take_u(rq[ph][1][2]), ux, uy, uz, invd_unused);   // (moved out first: the buffer must be dead at its refill, see take())
            u32x4 rb2 = dense_b2(rq[ph][0], rq[ph][1]);   // second B operand
            mfma_pre_fence(rq[ph][0], rb2);          // gathers are issued before the first MFMA
            EPH(3)   // gather issue (+ the LDS wait the clock read implies)
            // ---- filter GEMM  D[tile row][slot] = Wd_ext[row][k] rho[k][slot]  (bias . fc included): 2 instructions per tile ------
            f32x4 acc[NT];
            {
                const u32x4 (*wp[NT])[2], *bp1[NT], *bp2[NT];
#pragma unroll
                for (int s2 = 0; s2 < NT; ++s2) { wp[s2] = &wA[s2]; bp1[s2] = &rq[ph][0]; bp2[s2] = &rb2; }
                filter_tiles_dense<NT>(wp, bp1, bp2, acc);
            }
            __builtin_amdgcn_sched_barrier(0);
            EPH(4)   // matrix instructions issued
            // ---- messages of this lane's slot for its NF features (filter = 0 exactly for pads / foreign slots) ----------
            auto message = [&](int r) {
                const float *tr = tv + r * LY::NSEG;
                const float wa = acc[sec_tile<NF>(0)][sec_reg<NF>(0, r)], wb = acc[sec_tile<NF>(1)][sec_reg<NF>(1, r)],
                            wc = acc[sec_tile<NF>(2)][sec_reg<NF>(2, r)];
                ds[r] = fmaf(tr[1], wb, ds[r]);
                const float mc = tr[2] * wc, ma = tr[0] * wa;
                dvx[r] = fmaf(mc, ux, dvx[r]); dvy[r] = fmaf(mc, uy, dvy[r]); dvz[r] = fmaf(mc, uz, dvz[r]);
                dvx[r] = fmaf(ma, tr[3], dvx[r]);
                dvy[r] = fmaf(ma, tr[4], dvy[r]);
                dvz[r] = fmaf(ma, tr[5], dvz[r]);
            };
            message(0);   // consumes every accumulator tile (NF = 2: wa, wb from tile 0, wc from tile 1)
            // table entries of the step after next, into the buffer this step has just consumed; see mfma_load_fence
            bool nv;
            int nq = bw.quad_ahead(2, nv);
            mfma_load_fence(nq, ds[0], dvx[0], dvy[0]);
            EPH(5)   // matrix results arrive + first feature
            fetch(ph, nq, nv);
            __builtin_amdgcn_sched_barrier(0);
#ifdef ABL_HALF_VALU
#pragma unroll
            for (int r = 1; r < NF / 2; ++r) message(r);
#else
#pragma unroll
            for (int r = 1; r < NF; ++r) message(r);
#endif
            ++bw.t;
            EPH(6)   // prefetch issue + remaining features
        }
    }
    EPH_FLUSH(0)
}

// ======================================================================================================
// Reverse pass of the message block (oracle: painn_impl.inc "message block^T").
// Atom c in its role as SOURCE j: for every neighbor n the edge (n -> c) carried phi_c, v_c into n.
//   phibar_c[b] += w[b] sbar_n ; phibar_c[c] += w[c] (vbar_n . u_nc) ; phibar_c[a] += w[a] (vbar_n . v_c)
//   vbar_c     += phi_c[a] w[a] vbar_n
//   dE/dd(n->c) = sum_f  dw[b] phi_c[b] sbar_n + dw[c] phi_c[c] (vbar_n.u_nc) + dw[a] phi_c[a] (vbar_n.v_c)
//   dE/du(n->c) = sum_f  phi_c[c] w[c] vbar_n            (dw = Wd_ext . d rho/dd)
// Same slot-major lane layout as the forward kernel: lane (p, fq) owns slot p and features 4fq..4fq+3, so
// the sums over features are 4 in-lane terms + a reduction over the 4 feature quarters (lanes p, p+16,
// p+32, p+48).  One workgroup = (chain, model, group of SLICES_PER_WG feature slices) with its own partial edge-gradient
// buffer G[m][group][slot]: within a layer every slot is written once per group, across layers the SAME lane adds to
// it in a fixed order (deterministic read-modify-write, old value prefetched one step ahead); finalize reduces the
// groups in a streaming pass.
// Workgroup width WAVES.  4 waves: two workgroups share a CU (81 KB of LDS each at 272 atoms, 16-feature slices), one stages its
// slice while the other computes (6.4 -> 6.0 ms / step), 2 waves per SIMD with the 256-register budget.  When the slice of the
// launch's largest chain is too big for two workgroups per CU (> 301 atoms with 16-feature slices, > 573 with 8-feature
// slices) a 4-wave workgroup would run alone with ONE wave per SIMD: those launches use 8 waves (same registers, one
// workgroup per CU).  (8-feature slices, 145 registers: 4 / 6 / 8 waves at two workgroups per CU measured 13.1 / 13.4 / 23.7 ms
// per step on 490-atom chains -- 8 waves x 2 workgroups spill at 128 registers.)
template <int NF, int WAVES>
constexpr int cen_floats() { return WAVES * 4 * 4 * 6 * NF; }   // current-centre store: [stream][feature quarter][phi a, b, c, v x, y, z][NF]
template <int NF, int WAVES>
#ifdef ABL_LDS_FORCE   // ablation (profiles/r04/NOTES_force_accum.md): per-atom fixed-point force accumulators in LDS instead of per-slot records
size_t edge_bwd_lds_bytes_t(int max_atoms) { return sizeof(float) * ((size_t)max_atoms * EdgeGeo<NF>::ROWB + cen_floats<NF, WAVES>()) + 24 * (size_t)max_atoms + 32; }
#else
size_t edge_bwd_lds_bytes_t(int max_atoms) { return sizeof(float) * ((size_t)max_atoms * EdgeGeo<NF>::ROWB + cen_floats<NF, WAVES>()); }
#endif

// FIRST: the launch writes the partial edge-gradient buffers for the first time (last layer): nothing to add to.
// (The narrow slices need fewer registers -- 145 / 110 with 8- / 4-feature slices -- but wider workgroups, 12 resp. 16 waves when one
// workgroup owns the CU, changed nothing: 9.08 vs 9.01 .. 9.22 ms at 700 atoms, 37.1 vs 36.9 at 1 400, profiles/r04/ab_bwd_wide.txt;
// these launches are bound by the L2 -> L1 stream of the per-slot tables, which every slice re-reads, not by latency.)
// SUB: the multi-pass form of the forward kernel (see there) for the reverse pass: pass p stages [sbar, vbar] of the neighbors of
// range p plus a zero row and walks the windows of its per-pass bundle table; a slot whose neighbor is outside the range gathers
// zeros (every term of its contribution carries sbar_n / vbar_n) and sends its gradient record to the spare entry, so the record of
// a slot is written by exactly one pass.  Pass 0 writes phibar_c / vbar_c (+ the residual vbar_msg_c, from memory); a later pass
// adds to what is there.
template <int NF, bool FIRST, int WAVES, bool SUB = false>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, 2)))   // 2 waves per SIMD either way: use the 256 VGPRs
k_edge_bwd_mfma(int N, int l, const ModelW *__restrict__ MW, GraphView G,
                const int *__restrict__ counters, int zero_slot, int n_models, int max_atoms,
                const int *__restrict__ list, int n_list,
                const float *__restrict__ v_in, const float *__restrict__ phi, const float *__restrict__ sbar_msg,
                const float *__restrict__ vbar_msg, typename EdgeOut<SUB>::ptr phibar, typename EdgeOut<SUB>::ptr vbar_in,
                float *__restrict__ gbar, long long gbar_stride, int n_groups, int group_off, int rec,
                const int4 *__restrict__ bundle_tab, const int *__restrict__ n_entries_tab, int pass, int chunk_max) {
    using LY = EdgeGeo<NF>;
    constexpr int FS = LY::FS, NSG = LY::NSLICE, NT = LY::NT, ROWB = LY::ROWB, BWD_THREADS = 64 * WAVES;
    typedef typename FeatVec<NF>::type fvx;   // the lane's NF features (ext vectors: arrays of HIP float4 stay in scratch memory)
    extern __shared__ __attribute__((aligned(16))) float tile[];
    if (counters[2]) return;
    int t;
    const int b = chain_of_workgroup(G, list, n_list, NSG * n_models, t);
    if (b < 0) return;
    const int fs = t % NSG, m = t / NSG;   // feature slice = partial-gradient group
    const int a0 = G.cfg_start[b], Nc = G.cfg_start[b + 1] - a0;
    const size_t mN = (size_t)m * N;
    const int tid = threadIdx.x;
    int lo = 0, ns = Nc;   // staged neighbors: all atoms of the chain, or (SUB) range `pass`
    if constexpr (SUB) {
        if (pass >= sub_passes(Nc, chunk_max)) return;   // (uniform)
        const int chunk = sub_chunk(Nc, chunk_max);
        lo = pass * chunk;
        ns = min(chunk, Nc - lo);
    }
    const int lane = tid & 63, wave = tid >> 6, p = lane & 15, fq = lane >> 4, e = p & 3;
    const int last_slot = max(G.row_start[a0 + Nc] - 1, 0);
    const LayerW &W = MW[m].layer[l];
    // partial edge-gradient buffer of this (model, layer set, slice): rec floats per slot -- 4 (float4 records, group 0 doubles
    // as the final buffer) or 3 (compact per-layer buffers, reduced into the separate final buffer by k_reduce_gpart)
    // records of a FIRST launch are the compact 12-byte ones (painn.hip launches FIRST only onto them): a constant stride lets the
    // per-step store address be a shift-add instead of a quarter-rate 64-bit multiply-add
    const int rec_c = FIRST ? 3 : rec;
    float *gb = gbar + (size_t)(m * n_groups + group_off + fs) * gbar_stride * rec_c;
    const u32x4 *rho_lane = reinterpret_cast<const u32x4 *>(G.rho16) + fq * 4 + e;   // quad-interleaved tables, see forward
    const u32x4 *drho_lane = reinterpret_cast<const u32x4 *>(G.drho16) + fq * 4 + e;
    const int zero_quad = (zero_slot + 1) / 4 - 1;
    const int fcol = fs * FS + NF * fq;

    // which gradient component the reduce-scatter of the hot loop leaves in this lane's row: the same swap network run
    // once on tags (0, 1, 2 and 3 = the zero filler), so the mapping never depends on a reading of the ISA manual
    int gcomp_id;
    {
        unsigned t0 = 0u, t1 = 1u, t2 = 2u, t3 = 3u;
        asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
        const auto a = __builtin_amdgcn_permlane32_swap(t0, t1, false, false);
        const auto bq = __builtin_amdgcn_permlane32_swap(t2, t3, false, false);
        const auto cq = __builtin_amdgcn_permlane16_swap(a[0], bq[0], false, false);
        gcomp_id = (a[0] == a[1] && bq[0] == bq[1] && cq[0] == cq[1]) ? (int)cq[0] : 3;
        if (!(a[0] == a[1] && bq[0] == bq[1] && cq[0] == cq[1])) __builtin_trap();
    }

    // ---- stage [atom][f][sbar, vbar_x, vbar_y, vbar_z] of this slice ------------------------------------------
    {
        constexpr int Q4 = FS / 4;           // float4 per segment
        const int total = ns * 4 * Q4;       // 4 segments per atom
        if constexpr (SUB) {   // the all-zero row behind the staged ones
            for (int k = tid; k < ROWB; k += BWD_THREADS) tile[(size_t)ns * ROWB + k] = 0.f;
        }
        for (int base = tid; base < total; base += STAGE_BATCH * BWD_THREADS) {
            float4 v4[STAGE_BATCH];
#pragma unroll
            for (int u = 0; u < STAGE_BATCH; ++u) {
                const int idx = min(base + u * BWD_THREADS, total - 1);
                const int atom = idx / (4 * Q4), seg = (idx / Q4) & 3, q4 = idx % Q4;
                const size_t ga = mN + a0 + lo + atom;
                const float *src = seg == 0 ? sbar_msg + ga * F + fs * FS + q4 * 4
                                            : vbar_msg + (ga * 3 + (seg - 1)) * F + fs * FS + q4 * 4;
                v4[u] = *reinterpret_cast<const float4 *>(src);
            }
#pragma unroll
            for (int u = 0; u < STAGE_BATCH; ++u) {
                const int idx = base + u * BWD_THREADS;
                if (idx < total) {
                    const int atom = idx / (4 * Q4), seg = (idx / Q4) & 3, q4 = idx % Q4;
                    float *dst = tile + atom * ROWB + (q4 * 4) * 4 + seg;
                    dst[0] = v4[u].x; dst[4] = v4[u].y; dst[8] = v4[u].z; dst[12] = v4[u].w;
                }
            }
        }
    }
    // ---- A operand: the slice's filter rows in tile order (see the forward kernel) ------------------------------------------
    u32x4 wA[NT][2];   // fp16 pieces (h, l), bias column included (build_wd16); empty rows of the packed 8-feature tiles are zero
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        const int sec = LY::row_section(T, p);
        const int row = max(sec, 0) * F + fs * FS + LY::row_feature(p);
        const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(W.wd16) + ((size_t)row * 4 + fq) * 2;
#pragma unroll
        for (int i3 = 0; i3 < 2; ++i3) wA[T][i3] = sec < 0 ? (u32x4){0u, 0u, 0u, 0u} : gload_u32x4(wsrc + i3);
    }
#ifdef ABL_LDS_FORCE
    unsigned long long *facc = reinterpret_cast<unsigned long long *>(tile + (((size_t)max_atoms * ROWB + cen_floats<NF, WAVES>() + 1) & ~(size_t)1));
    for (int i = tid; i < 3 * Nc + 3; i += BWD_THREADS) facc[i] = 0ull;
    float gown = 0.f;
#endif
    __syncthreads();

    BundleWalk<BWD_THREADS / 64> bw;   // work list: bundles of 4 centres, see the forward kernel
    int n_entries = Nc;   // (an extra early exit here changes the compiler's memory-wait merge at the loop header: measured +5 % on the
                          //  single-pass reverse kernel; a pass without entries leaves through `bw.nj == 0` like an empty chain)
    if constexpr (SUB) {
        if (pass) n_entries = n_entries_tab[(size_t)(pass - 1) * G.n_cfg + b];
    }
    bw.init((SUB ? bundle_tab : G.bundle) + a0, n_entries, __builtin_amdgcn_readfirstlane(wave), p >> 2);
    if (bw.nj == 0) return;   // (no barrier below)
#ifdef ABL_STAGE_ONLY   // ablation: staging + prologue only (results wrong)
    if (bw.nj > 0) return;
#endif

    // ---- per-centre data of this lane's NF features: phi_c (a, b, c) and v_c -------------------------------------------
    // The CURRENT centre's values live in a small LDS record per (stream, feature quarter) and are re-read every step
    // next to the neighbor gathers (6 broadcast LDS reads); the NEXT bundle's values are in flight in registers
    // and are parked in the record when the wave moves on.  (Keeping both sets in registers made the compiler
    // rotate ~50 registers per completed centre and wait on the prefetch it had just issued.)
    const int cen_idx = max_atoms * ROWB + ((wave * 4 + (p >> 2)) * 4 + fq) * 6 * NF;   // float index in tile[]
    fvx *cen = reinterpret_cast<fvx *>(tile + cen_idx);
    fvx nx[6];
    auto load_centre = [&](int cc) {   // always a load (any valid row for streams without a centre)
        const size_t ga = mN + a0 + min(max(cc, 0), Nc - 1);
        const float *pr = phi + ga * F3 + fcol;
        nx[0] = *reinterpret_cast<const fvx *>(pr);
        nx[1] = *reinterpret_cast<const fvx *>(pr + F);
        nx[2] = *reinterpret_cast<const fvx *>(pr + 2 * F);
        nx[3] = *reinterpret_cast<const fvx *>(v_in + (ga * 3 + 0) * F + fcol);
        nx[4] = *reinterpret_cast<const fvx *>(v_in + (ga * 3 + 1) * F + fcol);
        nx[5] = *reinterpret_cast<const fvx *>(v_in + (ga * 3 + 2) * F + fcol);
    };
    auto park_centre = [&]() {   // the quad's 4 slot lanes hold identical values: any of them may write
#pragma unroll
        for (int q = 0; q < 6; ++q) cen[q] = nx[q];
    };
    load_centre(bw.cur.x);
    park_centre();
    load_centre(bw.nxt.x);
    // (SUB) what the centre's sums are added to, requested when its bundle starts: vbar_msg_c (pass 0: the residual, the centre's own
    // row may not be staged) resp. the phibar_c / vbar_c the earlier passes wrote
    fvx rres[SUB ? 6 : 1];
    auto load_rres = [&](int cc) {
        const size_t ga = mN + a0 + min(max(cc, 0), Nc - 1);
        const float *vsrc = pass ? vbar_in : vbar_msg;
#pragma unroll
        for (int x = 0; x < 3; ++x) rres[SUB ? 3 + x : 0] = *reinterpret_cast<const fvx *>(vsrc + (ga * 3 + x) * F + fcol);
        if (pass) {   // (uniform)
#pragma unroll
            for (int q = 0; q < 3; ++q) rres[SUB ? q : 0] = *reinterpret_cast<const fvx *>(phibar + ga * F3 + q * F + fcol);
        }
    };
    if constexpr (SUB) load_rres(bw.cur.x);
    float accb[NF], accc[NF], accx[NF], accy[NF], accz[NF];
#pragma unroll
    for (int r = 0; r < NF; ++r) { accb[r] = 0.f; accc[r] = 0.f; accx[r] = 0.f; accy[r] = 0.f; accz[r] = 0.f; }

    // the 4 centres of a bundle complete in the same step: all lanes reduce, the first lane of every quad writes
    auto flush_bundle = [&]() {
        quad_sum_n<NF>(accb); quad_sum_n<NF>(accc); quad_sum_n<NF>(accx); quad_sum_n<NF>(accy); quad_sum_n<NF>(accz);
        const float (&tb)[NF] = accb, (&tc)[NF] = accc, (&tx)[NF] = accx, (&ty)[NF] = accy, (&tz)[NF] = accz;
        const int c = bw.cur.x;
#ifdef ABL_LDS_FORCE
        {
            const float go = quad_sum(gown);
            gown = 0.f;
            if (e == 0 && c >= 0 && gcomp_id < 3)
                atomicAdd(&facc[3 * c + gcomp_id], (unsigned long long)__float2ll_rn(go * 4294967296.f));
        }
#endif
        if (e == 0 && c >= 0) {
            const size_t ga = mN + a0 + c;
            const fvx cv[6] = {cen[0], cen[1], cen[2], cen[3], cen[4], cen[5]};   // record of the completed centre
            float a_[NF], x_[NF], y_[NF], z_[NF];
            const float *res = tile + c * ROWB + (NF * fq) * 4;   // [r][sbar, vbar_x, vbar_y, vbar_z]
#pragma unroll
            for (int r = 0; r < NF; ++r) {
                a_[r] = fmaf(cv[5][r], tz[r], fmaf(cv[4][r], ty[r], cv[3][r] * tx[r]));
                x_[r] = fmaf(cv[0][r], tx[r], SUB ? rres[SUB ? 3 : 0][r] : res[4 * r + 1]);
                y_[r] = fmaf(cv[0][r], ty[r], SUB ? rres[SUB ? 4 : 0][r] : res[4 * r + 2]);
                z_[r] = fmaf(cv[0][r], tz[r], SUB ? rres[SUB ? 5 : 0][r] : res[4 * r + 3]);
            }
            float *pbp = phibar + ga * F3 + fcol;
            if (SUB && pass) {   // (uniform) add to what the earlier passes wrote
                float b_[NF], c_[NF];
#pragma unroll
                for (int r = 0; r < NF; ++r) { a_[r] += rres[0][r]; b_[r] = tb[r] + rres[SUB ? 1 : 0][r]; c_[r] = tc[r] + rres[SUB ? 2 : 0][r]; }
                store_feat<NF>(pbp, a_);
                store_feat<NF>(pbp + F, b_);
                store_feat<NF>(pbp + 2 * F, c_);
            } else {
            store_feat<NF>(pbp, a_);
            store_feat<NF>(pbp + F, tb);
            store_feat<NF>(pbp + 2 * F, tc);
            }
            store_feat<NF>(vbar_in + (ga * 3 + 0) * F + fcol, x_);
            store_feat<NF>(vbar_in + (ga * 3 + 1) * F + fcol, y_);
            store_feat<NF>(vbar_in + (ga * 3 + 2) * F + fcol, z_);
        }
#pragma unroll
        for (int r = 0; r < NF; ++r) { accb[r] = 0.f; accc[r] = 0.f; accx[r] = 0.f; accy[r] = 0.f; accz[r] = 0.f; }
    };

    // Table entries of this lane's slot; exhausted streams read the all-zero quad.  Two buffers, one per step parity:
    // buffer ph is consumed by the step of parity ph and refilled right after that step's MFMAs for the step after next
    // (~1.6 steps for a load to arrive).  The old partial edge gradient of the slot (written by the previous layer for
    // this lane's slot; a slot is visited once per launch, so the early read is safe) travels with the tables.
    float *gcomp = gb + min(gcomp_id, rec_c - 1);   // component this row ends up with (id 3: none -- that row only ever writes the spare entry)
    u32x4 rq[2][2], dq[2][2];
    float gold[2] = {0.f, 0.f};
    auto fetch = [&](int buf, int quad_first_slot, bool valid) {
#ifdef ABL_TABLE_L2
        const size_t off = (size_t)(valid ? ((quad_first_slot >> 2) & 511) : zero_quad) * 32;
#else
        const size_t off = (size_t)(unsigned)(valid ? (quad_first_slot >> 2) : zero_quad) * 32;   // (unsigned: no sign extension, one shift-add)
#endif
        const u32x4 *rp = rho_lane + off, *dp = drho_lane + off;
        rq[buf][0] = rp[0]; rq[buf][1] = rp[16];
#ifdef ABL_3LOADS   // ablation: three table loads per step instead of four (results wrong)
        dq[buf][0] = dp[0]; dq[buf][1] = dq[buf][0];
#else
        dq[buf][0] = dp[0]; dq[buf][1] = dp[16];
#endif
#ifdef ABL_NO_GBAR   // ablation build (tools/build_variant.sh): same instruction stream, the partial edge-gradient buffers stay in L2
        if (!FIRST) gold[buf] = gcomp[(size_t)min((quad_first_slot + e) & 1023, last_slot) * rec_c];
#else
        if (!FIRST) gold[buf] = gcomp[(size_t)min(quad_first_slot + e, last_slot) * rec_c];
#endif
    };
    {
        bool v0, v1;
        const int q0 = bw.quad_ahead(0, v0), q1 = bw.quad_ahead(1, v1);
        fetch(0, q0, v0);
        // issue order = consumption order.  Without the barrier the scheduler hoisted the second buffer's loads above the first's in
        // the forward kernel; seen from the loop header the first buffer was then the YOUNGEST pending load, its wait vmcnt(0), and
        // -- the header wait being the merge of pre-header and back edge -- every iteration drained the whole prefetch queue.
        __builtin_amdgcn_sched_barrier(0);
        fetch(1, q1, v1);
    }
    lds_cfloat *trow = lds_base(tile) + (NF * fq) * 4;
    EPH_INIT
    EPH(0)
    while (bw.j < bw.nj) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            arrival_fence(rq[ph][0], rq[ph][1], dq[ph][0], dq[ph][1], gold[ph]);
            EPH(1)
            bool in_range = true;
#ifdef ABL_LDS_BCAST   // ablation: every gather reads row 0 or 1 (no LDS bank conflicts; results wrong)
            const int jn = (int)(rq[ph][0][3] >> 16) & 1;
#else
            int jn = (int)rq[ph][1][3];           // chain-local neighbor of this lane's slot (last word of the record's unit 1)
            if constexpr (SUB) {   // staged row of the neighbor, or the zero row for a neighbor of another range
                const unsigned jl = (unsigned)(jn - lo);
                in_range = jl < (unsigned)ns;
                jn = in_range ? (int)jl : ns;
            }
#endif
            const float gold_cur = FIRST ? 0.f : take(gold[ph]);
            if (bw.t == bw.Lc) {   // wave-uniform: the bundle is complete
                flush_bundle();
                bw.advance();
                park_centre();                       // centre data of the bundle that starts now
                __builtin_amdgcn_sched_barrier(0);   // the old values leave their registers before the loads that refill them are issued
                load_centre(bw.nxt.x);
                if constexpr (SUB) load_rres(bw.cur.x);
                EPH(2)
            }
            bool real_slot;
            const int my_slot = bw.quad_ahead(0, real_slot) + e;
            float tb[4 * NF];
            {
                lds_cfloat *row = lds_row(trow, jn, ROWB);
#pragma unroll
                for (int q = 0; q < NF; ++q) {
                    const f32x4 t4 = lds_read4(row + 4 * q);
                    tb[4 * q] = t4[0]; tb[4 * q + 1] = t4[1]; tb[4 * q + 2] = t4[2]; tb[4 * q + 3] = t4[3];
                }
            }
            // (the record's index passes through an empty asm: otherwise the compiler forwards the parked registers to these
            // reads, hoists them into the tail of the previous step and shuffles / waits on the registers of the prefetch
            // it has just issued)
            fvx cv[6];
            {
                int ci = cen_idx;   // (the INDEX is laundered: a laundered pointer loses its LDS address space -> flat loads)
                asm volatile("" : "+v"(ci));
                const fvx *cr = reinterpret_cast<const fvx *>(tile + ci);
#pragma unroll
                for (int q = 0; q < 6; ++q) cv[q] = cr[q];
            }
            // unit vector c -> n (edge (n -> c) has -u) and 1 / d: the record's scalar words, all-gathered over the slot's four lanes
            float ux, uy, uz, invd;
            slot_scalars(take_u(rq[ph][1][2]), ux, uy, uz, invd);   // (moved out first: the buffer must be dead at its refill, see take())
            u32x4 rb2 = dense_b2(rq[ph][0], rq[ph][1]), db2 = dense_b2(dq[ph][0], dq[ph][1]);   // second B operands
            mfma_pre_fence(rq[ph][0], rb2);          // gathers are issued before the first MFMA
            mfma_pre_fence(dq[ph][0], db2);
            EPH(3)
            // filter and its radial derivative for this lane's slot and NF features (bias . fc / bias . fc' included): 2 instructions per tile
            f32x4 awd[2 * NT];   // tiles [0, NT): filter w, [NT, 2 NT): radial derivative dw
            {
                const u32x4 (*wp[2 * NT])[2], *bp1[2 * NT], *bp2[2 * NT];
#pragma unroll
                for (int s2 = 0; s2 < NT; ++s2) {
                    wp[s2] = &wA[s2]; bp1[s2] = &rq[ph][0]; bp2[s2] = &rb2;
                    wp[NT + s2] = &wA[s2]; bp1[NT + s2] = &dq[ph][0]; bp2[NT + s2] = &db2;
                }
                filter_tiles_dense<2 * NT>(wp, bp1, bp2, awd);
            }
            const f32x4 *aw = awd, *ad = awd + NT;
            __builtin_amdgcn_sched_barrier(0);
            EPH(4)
            float dpart = 0.f, ub0 = 0.f, ub1 = 0.f, ub2 = 0.f;
            auto feature = [&](int r) {
                const float sbn = tb[4 * r], vb0 = tb[4 * r + 1], vb1 = tb[4 * r + 2], vb2 = tb[4 * r + 3];
                const float wB = aw[sec_tile<NF>(1)][sec_reg<NF>(1, r)], wC = aw[sec_tile<NF>(2)][sec_reg<NF>(2, r)];
                const float dB = ad[sec_tile<NF>(1)][sec_reg<NF>(1, r)], dC = ad[sec_tile<NF>(2)][sec_reg<NF>(2, r)];
                const float pn = -fmaf(vb2, uz, fmaf(vb1, uy, vb0 * ux));   // vbar_n . u_(n->c)
                accb[r] = fmaf(wB, sbn, accb[r]);
                accc[r] = fmaf(wC, pn, accc[r]);
                dpart = fmaf(cv[1][r] * sbn, dB, dpart);
                dpart = fmaf(cv[2][r] * pn, dC, dpart);
                const float wAa = aw[sec_tile<NF>(0)][sec_reg<NF>(0, r)], dA = ad[sec_tile<NF>(0)][sec_reg<NF>(0, r)];
                const float q = fmaf(vb2, cv[5][r], fmaf(vb1, cv[4][r], vb0 * cv[3][r]));
                accx[r] = fmaf(wAa, vb0, accx[r]);
                accy[r] = fmaf(wAa, vb1, accy[r]);
                accz[r] = fmaf(wAa, vb2, accz[r]);
                dpart = fmaf(cv[0][r] * q, dA, dpart);
                const float mc = cv[2][r] * wC;
                ub0 = fmaf(mc, vb0, ub0); ub1 = fmaf(mc, vb1, ub1); ub2 = fmaf(mc, vb2, ub2);
            };
            feature(0);   // consumes every accumulator tile
            bool nv;
            int nq = bw.quad_ahead(2, nv);   // the step after next, into the buffer this step has just consumed
            mfma_load_fence(nq, accb[0], accc[0], dpart, ub0);
            EPH(5)
            fetch(ph, nq, nv);
            __builtin_amdgcn_sched_barrier(0);
#ifdef ABL_HALF_VALU   // ablation: half of the per-feature arithmetic (results wrong; profiles/r03/NOTES_packed_fp32.md)
#pragma unroll
            for (int r = 1; r < NF / 2; ++r) feature(r);
#else
#pragma unroll
            for (int r = 1; r < NF; ++r) feature(r);
#endif
            EPH(6)   // prefetch issue + remaining features
            // Gradient of edge (n -> c), unit vector -u:  g = -(dE/dd) u + (ub - (ub.u) u) / d, linear in (dpart, ub).
            // The map is applied to the lane's partial sums BEFORE the reduction over the 4 feature quarters (lanes
            // p, p+16, p+32, p+48), so only 3 values cross lanes, and the reduction is a reduce-scatter on the
            // gfx950 row swaps: swap32 pairs (g0, g1) / (g2, 0) -> rows {0,1} hold half sums of g0 / g2, rows {2,3} of
            // g1 / 0; swap16 of the two -> row 0 = g0, row 1 = g2, row 2 = g1 (complete sums, fixed order): 3 swaps +
            // 3 adds for the whole quantity, and row r writes one component.
            {
                const float dotu = fmaf(ub2, uz, fmaf(ub1, uy, ub0 * ux));
                float g0 = fmaf(-dpart, ux, fmaf(-dotu, ux, ub0) * invd);
                float g1 = fmaf(-dpart, uy, fmaf(-dotu, uy, ub1) * invd);
                float g2 = fmaf(-dpart, uz, fmaf(-dotu, uz, ub2) * invd);
                const auto h01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(g0), __float_as_uint(g1), false, false);
                const auto h2z = __builtin_amdgcn_permlane32_swap(__float_as_uint(g2), 0u, false, false);
                const float A = __uint_as_float(h01[0]) + __uint_as_float(h01[1]);
                const float B = __uint_as_float(h2z[0]) + __uint_as_float(h2z[1]);
                const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(A), __float_as_uint(B), false, false);
                const float gsum = __uint_as_float(q[0]) + __uint_as_float(q[1]);
                // three rows of a real slot write one component each; every other lane writes the spare entry, so the
                // store is unconditional and the memory-operation count of a step does not depend on the path
                const bool real = gcomp_id < 3 && real_slot && invd > 0.f && (!SUB || in_range);
#if defined(ABL_LDS_FORCE)
                gown += real ? gsum : 0.f;
                atomicAdd(&facc[real ? 3 * jn + gcomp_id : 3 * Nc], (unsigned long long)__float2ll_rn(-gsum * 4294967296.f));
#elif defined(ABL_NO_GSTORE)   // ablation: the store only happens for a value the arithmetic never produces
                if (gsum == 1.2345e33f) gcomp[(size_t)(real ? my_slot : zero_slot) * rec_c] = gsum + gold_cur;
#elif defined(ABL_STORE4)   // ablation: one 16-byte store every fourth step instead of a dword store per step (same bytes, results wrong)
                if ((bw.t & 3) == 3) {
                    float *dst = gcomp + (size_t)(real ? my_slot : zero_slot) * rec_c;
                    asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(dst), "v"((f32x4){gsum, gsum, gsum, gsum}) : "memory");
                }
#elif defined(ABL_ROW3_NO_GSTORE)   // ablation: the row without a component does not store
                if (gcomp_id < 3) gcomp[(size_t)(real ? my_slot : zero_slot) * rec_c] = gsum + gold_cur;
#elif defined(ABL_NO_GBAR)
                gcomp[(size_t)(real ? (my_slot & 1023) : zero_slot) * rec_c] = gsum + gold_cur;
#else
                gcomp[(unsigned)(real ? my_slot : zero_slot) * (unsigned)rec_c] = gsum + gold_cur;   // (32-bit index: shift-add for the constant stride)
#endif
            }
            ++bw.t;
            EPH(7)   // reduce-scatter + store
        }
    }
    EPH_FLUSH(8)
#ifdef ABL_LDS_FORCE   // partial forces of this (chain, slice, model, layer) leave as Nc x 3 fixed-point values (here: into the head of its gradient buffer)
    __syncthreads();
    {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(gb) + (size_t)3 * a0;
        for (int i = tid; i < 3 * Nc; i += BWD_THREADS) dst[i] = facc[i];
    }
#endif
}



int edge_mfma_init(vssr_handle *h) {
#define SET_LDS(K) VSSR_HIP(h, hipFuncSetAttribute((const void *)(K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    SET_LDS((k_edge_fwd_mfma<4, true, 16>)); SET_LDS((k_edge_fwd_mfma<4, false, 16>)); SET_LDS((k_edge_fwd_mfma<2, false, 16>)); SET_LDS((k_edge_fwd_mfma<2, false, 16, true>)); SET_LDS((k_edge_fwd_mfma<4, false, SUB16_WAVES, true>));
    SET_LDS((k_edge_fwd_mfma<4, true, 8>)); SET_LDS((k_edge_fwd_mfma<4, true, 4>)); SET_LDS((k_edge_fwd_mfma<1, false, 16>));
    SET_LDS((k_edge_bwd_mfma<1, true, 4>)); SET_LDS((k_edge_bwd_mfma<1, false, 4>));
    SET_LDS((k_edge_bwd_mfma<1, true, 8>)); SET_LDS((k_edge_bwd_mfma<1, false, 8>));
    SET_LDS((k_edge_bwd_mfma<4, true, 4>)); SET_LDS((k_edge_bwd_mfma<4, false, 4>));
    SET_LDS((k_edge_bwd_mfma<2, true, 4>)); SET_LDS((k_edge_bwd_mfma<2, false, 4>));
    SET_LDS((k_edge_bwd_mfma<4, true, 8>)); SET_LDS((k_edge_bwd_mfma<4, false, 8>));
    SET_LDS((k_edge_bwd_mfma<4, true, 8, true>)); SET_LDS((k_edge_bwd_mfma<4, false, 8, true>));
    SET_LDS((k_edge_bwd_mfma<2, true, 8>)); SET_LDS((k_edge_bwd_mfma<2, false, 8>));
#undef SET_LDS
    return VSSR_OK;
}

// Slice-width class of a chain by its atom count (EDGE_CLASS_*): the widest slice whose forward LDS tile holds the chain
// (the reverse tile is smaller).  nf = features per lane of the class (4 / 2).
int edge_class_of(int n_atoms) {
    if (edge_fwd_lds_bytes_t<4, true>(n_atoms) <= 160 * 1024) return EDGE_CLASS_FS16;
    if (edge_fwd_lds_bytes_t<4, false>(n_atoms) <= 160 * 1024) return EDGE_CLASS_FS16M;
    if (edge_fwd_lds_bytes_t<2, false>(n_atoms) <= 160 * 1024) return EDGE_CLASS_FS8;
    if (edge_fwd_lds_bytes_t<1, false>(n_atoms) <= 160 * 1024) return EDGE_CLASS_FS4;   // 112 B per atom: <= 1 462 atoms
    return EDGE_CLASS_GATHER;
}
// reverse path of a chain: 16-feature slices while the reverse tile fits LDS (8-wave workgroup), 8-feature slices for the rest
// of the forward 8-feature class, gather kernels exactly where the forward pass gathers
int edge_bclass_of(int n_atoms) {
    const int fwd = edge_class_of(n_atoms);
    if (fwd == EDGE_CLASS_GATHER) return EDGE_BCLASS_GATHER;
    if (edge_bwd_lds_bytes_t<4, 8>(n_atoms) <= 160 * 1024) return EDGE_BCLASS_FS16;
    if (edge_bwd_lds_bytes_t<2, 8>(n_atoms) <= 160 * 1024) return EDGE_BCLASS_FS8;     // 144 B per atom: <= 1 127 atoms
    return EDGE_BCLASS_FS4;
}
int edge_class_groups(int bcls) { return edge_bclass_slices(bcls); }
static_assert(edge_bclass_slices(EDGE_BCLASS_FS16) == EdgeGeo<4>::NSLICE && edge_bclass_slices(EDGE_BCLASS_FS8) == EdgeGeo<2>::NSLICE &&
              edge_bclass_slices(EDGE_BCLASS_FS4) == EdgeGeo<1>::NSLICE, "slices per reverse class");

// layers >= 1 only (layer 0: painn_l0.hip or the gather kernels, see painn_run).  cls: EDGE_BCLASS_FS16 / _FS8 (reverse) resp.
// EDGE_CLASS_FS16 / _FS16M / _FS8 (forward); list / n_list: the chains of that class; max_atoms: the largest of them.
// bundle_sub != nullptr: the multi-pass 16-feature form (per-pass bundle tables of these chains, cut for the reverse chunk size; the
// partial edge-gradient buffers are then those of the 16-feature class: 8 slices)
void launch_edge_bwd_mfma(hipStream_t st, int cls, int N, const int *list, int n_list, int M, int l, int layer_first, int max_atoms,
                          const ModelW *MW, const GraphView &G, const int *counters, int zero_slot,
                          const float *v_in, const float *phi, const float *sbar_msg, const float *vbar_msg,
                          float *phibar, float *vbar_in, float *gbar, long long gbar_stride, int n_groups, int group_off, int rec,
                          const int4 *bundle_sub, int sub_chunk) {
    if (n_list <= 0) return;
    if (cls == EDGE_BCLASS_FS16P) {
        const int chunk_max = sub_chunk, passes = sub_passes(max_atoms, chunk_max), staged = min(chunk_max, max_atoms) + 1;
        const int *n_entries = reinterpret_cast<const int *>(bundle_sub + (size_t)SUB_MAX_PASSES * N);
        for (int pass = 0; pass < passes; ++pass) {
#define LAUNCH_BWD_SUB(FIRST)                                                                                                    \
    hipLaunchKernelGGL((k_edge_bwd_mfma<4, FIRST, 8, true>), dim3(((n_list + 7) / 8) * 8 * EdgeGeo<4>::NSLICE * M), dim3(64 * 8),   \
                       (edge_bwd_lds_bytes_t<4, 8>(staged)), st, N, l, MW, G, counters, zero_slot, M, staged, list, n_list, v_in, phi, \
                       sbar_msg, vbar_msg, phibar, vbar_in, gbar, gbar_stride, n_groups, group_off, rec, bundle_sub + (size_t)pass * N, \
                       n_entries, pass, chunk_max)
            if (layer_first) LAUNCH_BWD_SUB(true); else LAUNCH_BWD_SUB(false);
#undef LAUNCH_BWD_SUB
        }
        return;
    }
#define LAUNCH_BWD(NF, FIRST, WAVES)                                                                                             \
    hipLaunchKernelGGL((k_edge_bwd_mfma<NF, FIRST, WAVES>), dim3(((n_list + 7) / 8) * 8 * EdgeGeo<NF>::NSLICE * M), dim3(64 * WAVES), \
                       (edge_bwd_lds_bytes_t<NF, WAVES>(max_atoms)), st, N, l, MW, G, counters, zero_slot, M, max_atoms, list, n_list, \
                       v_in, phi, sbar_msg, vbar_msg, phibar, vbar_in, gbar, gbar_stride, n_groups, group_off, rec,                \
                       (const int4 *)nullptr, (const int *)nullptr, 0, 0)
    // 4-wave workgroups when two of them can share a CU (LDS) AND there are enough workgroups to give every CU two; a small batch (a
    // single chain = 24 workgroups) takes 8 waves so that a lone workgroup still fills its CU's SIMDs: the per-workgroup walk is the
    // latency of the launch (1 chain of 260 atoms: 103 -> 55 us).  The width never changes a result (a centre's sums run in slot order
    // inside one wave whichever wave that is).
#define LAUNCH_BWD_W(NF, FIRST)                                                                                                  \
    do {                                                                                                                         \
        const bool two_fit = 2 * edge_bwd_lds_bytes_t<NF, 4>(max_atoms) <= 160 * 1024;                                           \
        const bool many = (long long)n_list * EdgeGeo<NF>::NSLICE * M > 256;                                                     \
        if (two_fit && many) LAUNCH_BWD(NF, FIRST, 4); else LAUNCH_BWD(NF, FIRST, 8);                                            \
    } while (0)
    if (cls == EDGE_BCLASS_FS4) { if (layer_first) LAUNCH_BWD_W(1, true); else LAUNCH_BWD_W(1, false); }
    else if (cls == EDGE_BCLASS_FS8) { if (layer_first) LAUNCH_BWD_W(2, true); else LAUNCH_BWD_W(2, false); }
    else { if (layer_first) LAUNCH_BWD_W(4, true); else LAUNCH_BWD_W(4, false); }
#undef LAUNCH_BWD_W
#undef LAUNCH_BWD
}

// bundle_sub: the per-pass bundle tables (nbr.hip k_bundle_sort_sub: [SUB_MAX_PASSES][N] entries, then [SUB_MAX_PASSES - 1][n_cfg] entry
// counts) of the chains of the 4-feature class (788 .. 1 462 atoms), or nullptr: those chains take the 4-feature kernel.  sub_width:
// 16 or 8 = slice width of the multi-pass form.
void launch_edge_fwd_mfma(hipStream_t st, int cls, int N, const int *list, int n_list, int M, int l, int max_atoms, const ModelW *MW,
                          const GraphView &G, const int *counters, int zero_slot, const float *s_in, const float *v_in,
                          const float *phi, float *s_msg, float *v_msg, const int4 *bundle_sub, int sub_width, int sub_chunk) {
    if (n_list <= 0) return;
#define LAUNCH_FWD(NF, SLDS, WAVES)                                                                                                \
    hipLaunchKernelGGL((k_edge_fwd_mfma<NF, SLDS, WAVES>), dim3(((n_list + 7) / 8) * 8 * EdgeGeo<NF>::NSLICE * M), dim3(64 * WAVES), \
                       (edge_fwd_lds_bytes_t<NF, SLDS>(max_atoms)), st, N, l, MW, G, counters, zero_slot, M, max_atoms, list, n_list, \
                       s_in, v_in, phi, s_msg, v_msg, (const int4 *)nullptr, (const int *)nullptr, 0, 0)
    if (cls == EDGE_CLASS_FS16) {   // workgroups per CU that the launch's largest slice allows -> waves per workgroup
        // (a launch with fewer workgroups than CUs -- a single chain -- takes the widest form: its latency is one workgroup's walk)
        const size_t lds = edge_fwd_lds_bytes_t<4, true>(max_atoms);
        const long long wgs = (long long)n_list * EdgeGeo<4>::NSLICE * M;
        if (4 * lds <= 160 * 1024 && wgs > 4 * 256) LAUNCH_FWD(4, true, 4);
        else if (2 * lds <= 160 * 1024 && wgs > 2 * 256) LAUNCH_FWD(4, true, 8);
        else LAUNCH_FWD(4, true, 16);
    } else if (cls == EDGE_CLASS_FS16M) LAUNCH_FWD(4, false, 16);
    else if (cls == EDGE_CLASS_FS8) LAUNCH_FWD(2, false, 16);
    else if (bundle_sub) {
        // wider slices over P neighbor sub-ranges: pass 0 (range 0 staged; writes sums + residual), then the passes that add to it
        const int chunk_max = sub_chunk, passes = sub_passes(max_atoms, chunk_max);
        const int staged = min(chunk_max, max_atoms) + 1;   // rows of the LDS tile: the largest range + the zero row
        const int *n_entries = reinterpret_cast<const int *>(bundle_sub + (size_t)SUB_MAX_PASSES * N);
        for (int pass = 0; pass < passes; ++pass) {
            if (sub_width == 16)
                hipLaunchKernelGGL((k_edge_fwd_mfma<4, false, SUB16_WAVES, true>), dim3(((n_list + 7) / 8) * 8 * EdgeGeo<4>::NSLICE * M), dim3(64 * SUB16_WAVES),
                                   (edge_fwd_lds_bytes_t<4, false>(staged)), st, N, l, MW, G, counters, zero_slot, M, staged, list, n_list,
                                   s_in, v_in, phi, s_msg, v_msg, bundle_sub + (size_t)pass * N, n_entries, pass, chunk_max);
            else
                hipLaunchKernelGGL((k_edge_fwd_mfma<2, false, 16, true>), dim3(((n_list + 7) / 8) * 8 * EdgeGeo<2>::NSLICE * M), dim3(64 * 16),
                                   (edge_fwd_lds_bytes_t<2, false>(staged)), st, N, l, MW, G, counters, zero_slot, M, staged, list, n_list,
                                   s_in, v_in, phi, s_msg, v_msg, bundle_sub + (size_t)pass * N, n_entries, pass, chunk_max);
        }
    } else LAUNCH_FWD(1, false, 16);
#undef LAUNCH_FWD
}

}  // namespace vssr
