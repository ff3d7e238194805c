// painn_node_mfma.hip — per-atom ("node") stages of PaiNN on the gfx950 matrix cores.
//
// The node stages are chains of small dense layers applied to every atom of every
// chain and ensemble member: genuinely dense GEMMs with M = atoms (x3 Cartesian rows for U/V), K, N in {128, 256, 384}.
// gfx950 has no fp32 matrix rate above the vector rate (v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate and
// measurably conflicts with the VALU; no xf32 exists), so the GEMMs run on the 16-bit matrix pipe with fp32-level
// accuracy: both operands are split into two fp16 pieces x = h + l (h = fp16(x), l = fp16(x - h): 22 mantissa bits;
// fp16 subnormals are honoured by the matrix core) and the three products a_h w_l + a_l w_h + a_h w_h are accumulated in
// fp32 on v_mfma_f32_16x16x32_f16.
//
// Structure of every kernel: one workgroup = 8 waves = a tile of 32 atoms of one ensemble member.
//   * activations live in LDS as two fp16 planes (h and l, same bytes as an fp32 tile): every element is split ONCE by
//     the thread that stores it (clamped to +-65504 first; measured activations / adjoints of the SrTiO3 models peak at
//     ~160, tools/gpu_ranges.py), and an activation fragment is one ds_read_b128 per piece, no arithmetic;
//   * weights were pre-split at vssr_create into fragment order (pack_mfma_tiles16): a wave streams the pieces of its
//     column tiles with coalesced 1 KiB dwordx4 loads, L2-resident (all workgroups read the same ~1 MB);
//   * the weight pieces are the MFMA's A operand and the activation pieces its B operand: the result tile is
//     D[feature][atom], a lane owns four consecutive features of one atom (LaneGeo, mfma16.h) and stores / fetches
//     8- and 16-byte vectors (planes, staging tiles, biases);
//   * wave w owns output features [16w, 16w+16) of every section for all 32 atoms (two 16-row tiles), so gates, norms and
//     residuals that combine several GEMM outputs for the same (atom, feature) stay in one lane's registers; with half
//     the per-wave accumulator state of a 32-column layout two waves share a SIMD and hide each other's latencies;
//   * fp32 residuals: update_fwd keeps its input tile in registers, update_bwd keeps vbar as an fp32 LDS tile.
// MFMA issue order follows the hazard rules of painn_edge_mfma.hip: the three products of a tile are a dependent chain,
// so products are issued in rounds over >= 3 independent accumulators (K-interleaved partial accumulators for GEMMs with
// two tiles), pinned with scheduling barriers, and no load is issued inside the MFMA block of a chunk group.
#include "mfma16.h"

#ifdef PHASE_TIMING   // debug build only (tools/gpu_phase.py): wall-clock ticks between phase marks, first thread of every workgroup
__device__ unsigned long long g_phase[64];
#define PH_INIT unsigned long long ph_t = wall_clock64();
#define PH_RESET ph_t = wall_clock64();
#define PH(k) { const unsigned long long ph_n = wall_clock64(); if (threadIdx.x == 0) atomicAdd(&g_phase[k], ph_n - ph_t); ph_t = wall_clock64(); }
extern "C" int vssr_debug_phases(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[64] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define PH_INIT
#define PH_RESET
#define PH(k)
#endif

namespace vssr {

// ---- GEMM on the planes ------------------------------------------------------------------------------------------------
// acc[t][c] += A(rows 16 t .. 16 t + 15, K) . W(column tile c)^T.  Activation fragment of lane (r = lane & 15, g = lane >> 4)
// for chunk q: row 16 t + r, k = 32 q + 8 g .. + 7 -> one ds_read_b128 per piece.  wq[c]: pieces of a 16-column tile,
// [K/32][2][64 lanes] uint4 (pack_mfma_tiles16).  The weight pieces are the MFMA's A operand, the activation pieces its B
// operand: D[feature][atom], lane (r, g) holds atom row r, columns 4 g .. 4 g + 3 of the tile (LaneGeo, mfma16.h).
// gemm16<K, NRT, NCT, PF>.  PF = 0: chunk groups one after the other (load, wait, multiply) in a rolled loop -- the
// smallest register footprint (update_fwd runs two workgroups per CU inside 128 registers).
// PF = 1: software pipeline.  Measured per-phase timings (profiles/r02/NOTES_node_kernels.md): a chunk group cost ~0.65 us
// wall whatever its matrix work (12 .. 36 MFMAs), i.e. the load -> wait -> MFMA round trip.  The weight pieces (global
// memory, L2 latency several hundred cycles under load) of group i + 1 are requested BEFORE the matrix instructions of
// group i are issued and land in the other half of a ping-pong register buffer; GEMMs with fewer than three tiles also
// read the A fragments (LDS) of group i + 1 ahead.  The loop is fully unrolled (K is a template parameter) so that the
// buffer halves are static registers.  Issue rules as before: loads are issued in front of an MFMA block, never inside
// it; the block is pinned with scheduling barriers.
// Cross-GEMM preload: a GEMM phase of these kernels starts behind an epilogue and a barrier, and the first weight pieces
// used to be requested after them -- one exposed L2 round trip per GEMM, 5 .. 9 per workgroup (phase timings:
// profiles/r02/NOTES_node_kernels.md).  WPre holds the pieces of the first NPRE chunk groups; gemm16_preload() issues their
// loads BEFORE the preceding epilogue (the ping-pong buffers of the previous GEMM are dead there, so the peak register
// demand does not grow) and gemm16<..., NPRE> starts on them.  Plain global loads survive __syncthreads().
template <int NRT, int NCT>
struct GemmShape {
    static constexpr int NT = NRT * NCT, NACC = NT >= 3 ? 1 : 2;
};
template <int NRT, int NCT, int NPRE>
struct WPre {
    u32x4 b[NPRE][GemmShape<NRT, NCT>::NACC][NCT][2];
};
template <int NRT, int NCT, int NPRE>
__device__ __forceinline__ void gemm16_preload(const uint4 *const (&wq)[NCT], WPre<NRT, NCT, NPRE> &p) {
    constexpr int NACC = GemmShape<NRT, NCT>::NACC;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int grp = 0; grp < NPRE; ++grp)
#pragma unroll
        for (int j = 0; j < NACC; ++j)
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc)
                    p.b[grp][j][c][pc] = gload4u(wq[c] + ((size_t)((grp * NACC + j) * 2 + pc) * 64 + lane));
    __builtin_amdgcn_sched_barrier(0);   // the requests stay in front of whatever follows
}

template <int K, int NRT, int NCT, int PF = 0, int NPRE = 0>
__device__ __forceinline__ void gemm16(const Planes &A, const uint4 *const (&wq)[NCT], f32x4 (&acc)[NRT][NCT],
                                       const WPre<NRT, NCT, (NPRE > 0 ? NPRE : 1)> *pre = nullptr) {
    constexpr int NT = NRT * NCT, NACC = NT >= 3 ? 1 : 2;
    static_assert(NPRE == 0 || PF == 1, "preloaded groups need the pipelined path");
    constexpr int NQ = K / 32;
    static_assert(NQ % NACC == 0, "chunk groups");
    constexpr int NG = NQ / NACC;
    constexpr bool PFA = PF && NT < 3;   // A fragments one group ahead as well
    constexpr int wi[3] = {0, 1, 0}, ri[3] = {1, 0, 0};   // A-piece, W-piece: a_h w_l, a_l w_h, a_h w_h
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const _Float16 *ah = A.h + r * A.ld + 8 * g, *al = A.l + r * A.ld + 8 * g;
    f32x4 part[NACC][NRT][NCT];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int t = 0; t < NRT; ++t)
#pragma unroll
            for (int c = 0; c < NCT; ++c) part[j][t][c] = j == 0 ? acc[t][c] : (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (PF == 0) {
#pragma unroll 1
        for (int q0 = 0; q0 < NQ; q0 += NACC) {
            u32x4 a[NACC][NRT][2], b[NACC][NCT][2];
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                const int q = q0 + j;
#pragma unroll
                for (int c = 0; c < NCT; ++c)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) b[j][c][pc] = gload4u(wq[c] + ((size_t)(q * 2 + pc) * 64 + lane));
#pragma unroll
                for (int t = 0; t < NRT; ++t) {
                    a[j][t][0] = *reinterpret_cast<const u32x4 *>(ah + t * 16 * A.ld + 32 * q);
                    a[j][t][1] = *reinterpret_cast<const u32x4 *>(al + t * 16 * A.ld + 32 * q);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j < NACC; ++j)
#pragma unroll
                    for (int t = 0; t < NRT; ++t)
#pragma unroll
                        for (int c = 0; c < NCT; ++c) {
                            part[j][t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                __builtin_bit_cast(f16x8, b[j][c][ri[k]]), __builtin_bit_cast(f16x8, a[j][t][wi[k]]),
                                part[j][t][c], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
        }
    } else {
        constexpr int NB = NPRE > 2 ? NPRE : 2;   // ring of weight-piece buffers
        static_assert(NPRE <= NG, "more preloaded groups than the GEMM has");
        u32x4 b[NB][NACC][NCT][2], a[PFA ? 2 : 1][NACC][NRT][2];
        auto load_b = [&](int buf, int grp) {
#pragma unroll
            for (int j = 0; j < NACC; ++j)
#pragma unroll
                for (int c = 0; c < NCT; ++c)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc)
                        b[buf][j][c][pc] = gload4u(wq[c] + ((size_t)((grp * NACC + j) * 2 + pc) * 64 + lane));
        };
        auto load_a = [&](int buf, int grp) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                const int q = grp * NACC + j;
#pragma unroll
                for (int t = 0; t < NRT; ++t) {
                    a[buf][j][t][0] = *reinterpret_cast<const u32x4 *>(ah + t * 16 * A.ld + 32 * q);
                    a[buf][j][t][1] = *reinterpret_cast<const u32x4 *>(al + t * 16 * A.ld + 32 * q);
                }
            }
        };
        if constexpr (NPRE > 0) {
#pragma unroll
            for (int g0 = 0; g0 < NPRE; ++g0)
#pragma unroll
                for (int j = 0; j < NACC; ++j)
#pragma unroll
                    for (int c = 0; c < NCT; ++c)
#pragma unroll
                        for (int pc = 0; pc < 2; ++pc) b[g0][j][c][pc] = pre->b[g0][j][c][pc];
        } else {
            load_b(0, 0);
        }
        if (PFA) load_a(0, 0);
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            const int cur = grp % NB, ca = PFA ? (grp & 1) : 0;
            // one group ahead into the other buffer; with NPRE >= 2 groups already on their way the ring is refilled NPRE
            // groups ahead instead, into the buffer of THIS group, after its MFMA block (below)
            if (NPRE < 2 && grp + 1 < NG) load_b(cur ^ 1, grp + 1);
            if (PFA) { if (grp + 1 < NG) load_a((grp + 1) & 1, grp + 1); }
            else load_a(0, grp);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j < NACC; ++j)
#pragma unroll
                    for (int t = 0; t < NRT; ++t)
#pragma unroll
                        for (int c = 0; c < NCT; ++c) {
                            part[j][t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                __builtin_bit_cast(f16x8, b[cur][j][c][ri[k]]), __builtin_bit_cast(f16x8, a[ca][j][t][wi[k]]),
                                part[j][t][c], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
            // End of the group's MFMA block (marker for tools/check_mfma_loads.py).  The loads that follow refill fragment
            // registers of MFMAs that have all been ISSUED (in-order issue; every round runs over >= 4 independent
            // accumulators, so none of them waits inside the pipe with unread sources).
            asm volatile("; gemm16_group_end");
            __builtin_amdgcn_sched_barrier(0);
            if (NPRE >= 2 && grp + NPRE < NG) {
                load_b(cur, grp + NPRE);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NRT; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            acc[t][c] = part[0][t][c];
#pragma unroll
            for (int j = 1; j < NACC; ++j) acc[t][c] += part[j][t][c];
        }
}

#ifndef UPD_PF
#define UPD_PF 1   // pipelined GEMMs in the reverse update kernel (256-register budget)
#endif
template <int NRT, int NCT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[NRT][NCT]) {
#pragma unroll
    for (int t = 0; t < NRT; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// 16-column tile `tile` of a packed matrix with inner dimension K: K/32 chunks x 2 pieces x 64 lanes uint4
#define WTILE(name, tile, K) (W.q##name + (size_t)(tile) * 4 * (K))

// ---- message MLP forward: phi = W2 swish(W1 s + b1) + b2 ---------------------------------------------------
// RT = 16-row tiles per wave: 2 = the 32-atom tile of every node kernel; 4 = 64 atoms per workgroup (the weight stream per atom
// halves: the row-scaling experiment of round 4, profiles/r04/NOTES_node_rows.md; only RT = 2 is instantiated).  The saturation watch
// attributes rows modulo 32 (mfma16.h), exact for RT = 2 only: the RT = 4 form is an experiment, never the default.
template <int RT, int PF>
__global__ void __launch_bounds__(NTHREADS)
k_msg_mlp_mfma(int N, int l, ActiveView av, const ModelW *__restrict__ MW, const float *__restrict__ s_in,
               float *__restrict__ phi) {
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
    constexpr int TA = 16 * RT;
    const Planes xs = make_planes(ldsh, TA, F), hs = make_planes(ldsh + plane_halves(TA, F), TA, F);
    const int m = blockIdx.y, a0 = blockIdx.x * TA;
    if (!av.tile(min(a0, N - 1), min(a0 + TA - 1, N - 1))) return;   // every chain of this atom tile is switched off
    const LaneGeo L;
    SatTrack sat;
    const LayerW &W = MW[m].layer[l];
    const size_t mN = (size_t)m * N;
    load_rows_split<TA>(xs, 0, [&](int row) { return s_in + (mN + min(a0 + row, N - 1)) * F; }, sat);
    __syncthreads();
    {
        f32x4 acc[RT][1];
        zero_acc(acc);
        const uint4 *wp[1] = {WTILE(W1, L.w, F)};
        const f32x4 b = gload4f(W.b1 + L.col0);
        __builtin_amdgcn_sched_barrier(0);   // the bias is requested in front of the GEMM whose epilogue adds it
        gemm16<F, RT, 1, PF>(xs, wp, acc);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            f32x4 hv = acc[t][0] + b;
#pragma unroll
            for (int i = 0; i < 4; ++i) hv[i] = swish(hv[i]);
            store_split4(hs, L.row(t), L.col0, hv, sat);
        }
    }
    __syncthreads();
    f32x4 acc[RT][3];
    zero_acc(acc);
    const uint4 *wp[3] = {WTILE(W2, L.w, F), WTILE(W2, NW + L.w, F), WTILE(W2, 2 * NW + L.w, F)};
    f32x4 b2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) b2[c] = gload4f(W.b2 + c * F + L.col0);
    __builtin_amdgcn_sched_barrier(0);
    gemm16<F, RT, 3, PF>(hs, wp, acc);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const f32x4 b = b2[c];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int a = a0 + L.row(t);
            if (a < N) *reinterpret_cast<f32x4 *>(phi + (mN + a) * F3 + c * F + L.col0) = acc[t][c] + b;
        }
    }
    sat.commit(av, a0, N);
}

// ---- message MLP reverse: sbar_in = sbar_msg + W1^T[(W2^T phibar) * swish'(W1 s + b1)] ------------------------------
__global__ void __launch_bounds__(NTHREADS)
k_msg_mlp_bwd_mfma(int N, int l, ActiveView av, const ModelW *__restrict__ MW, const float *__restrict__ s_in,
                   const float *__restrict__ phibar, const float *__restrict__ sbar_msg, float *__restrict__ sbar_in) {
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
    const Planes xs = make_planes(ldsh, TA, F);                          // s tile, later h1bar
    const Planes pb = make_planes(ldsh + plane_halves(TA, F), TA, F3);   // phibar tile
    const int m = blockIdx.y, a0 = blockIdx.x * TA;
    if (!av.tile(min(a0, N - 1), min(a0 + TA - 1, N - 1))) return;   // every chain of this atom tile is switched off
    const LaneGeo L;
    SatTrack sat;
    const LayerW &W = MW[m].layer[l];
    const size_t mN = (size_t)m * N;
    load_rows_split<TA>(xs, 0, [&](int row) { return s_in + (mN + min(a0 + row, N - 1)) * F; }, sat);
#pragma unroll
    for (int c = 0; c < 3; ++c)
        load_rows_split<TA>(pb, c * F, [&](int row) { return phibar + (mN + min(a0 + row, N - 1)) * F3 + c * F; }, sat);
    __syncthreads();
    f32x4 h1[2][1], a1[2][1];
    zero_acc(h1);
    zero_acc(a1);
    const f32x4 b1v = gload4f(W.b1 + L.col0);
    __builtin_amdgcn_sched_barrier(0);
    {
        const uint4 *wp[1] = {WTILE(W1, L.w, F)};
        gemm16<F, 2, 1>(xs, wp, h1);
        const uint4 *wq[1] = {WTILE(W2t, L.w, F3)};
        gemm16<F3, 2, 1>(pb, wq, a1);
    }
    __syncthreads();  // everyone is done reading xs
    {
        const f32x4 b = b1v;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 hv;
#pragma unroll
            for (int i = 0; i < 4; ++i) hv[i] = a1[t][0][i] * dswish(h1[t][0][i] + b[i]);
            store_split4(xs, L.row(t), L.col0, hv, sat);
        }
    }
    __syncthreads();
    f32x4 acc[2][1];
    zero_acc(acc);
    const uint4 *wp[1] = {WTILE(W1t, L.w, F)};
    gemm16<F, 2, 1>(xs, wp, acc);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int a = a0 + L.row(t);
        if (a < N)
            *reinterpret_cast<f32x4 *>(sbar_in + (mN + a) * F + L.col0) = gload4f(sbar_msg + (mN + a) * F + L.col0) + acc[t][0];
    }
    sat.commit(av, a0, N);
}

// ---- update block -----------------------------------------------------------------------------------------------
// LDS map (halves): vt [3*TA][F] | hs [TA][2F] | as [TA][F], two planes each (the reverse pass overlays it)
// RT = 16-row tiles per wave = TA / 16.  RT = 2: 32 atoms per workgroup, one workgroup per CU (the weight stream from L2 is
// amortised over 32 atoms); RT = 1: 16 atoms, half the LDS and accumulator registers, two workgroups per CU whose memory /
// LDS / matrix phases overlap.
template <int RT>
struct UpdLds {
    static constexpr int TA = 16 * RT;
    static constexpr int OFF_VT = 0;
    static constexpr int OFF_HS = plane_halves(3 * TA, F);
    static constexpr int OFF_AS = OFF_HS + plane_halves(TA, 2 * F);
    static constexpr int HALVES = OFF_AS + plane_halves(TA, F);   // RT = 2: 51 712 halves = 103 424 B
    static_assert(plane_halves(TA, F3) <= OFF_HS, "qb overlays vt");
    static_assert(plane_halves(3 * TA, 2 * F) <= HALVES, "[Ubar | Vbar] overlays the whole region");
};

template <int RT>
struct UpdRegs {
    f32x4 uv[3 * RT][2];   // [RT x + t][0] = U v_x, [..][1] = V v_x for atom row 16 t + r, columns col0 .. col0 + 3
    f32x4 h3[RT];          // pre-activation of the gate MLP
    f32x4 gate[RT][2];     // a_vv, a_sv (the reverse pass does not need a_ss)
    f32x4 nrm[RT], inner[RT];
};

// Shared forward part: needs vt (v_msg tile, rows x*TA+atom) and hs[:, :F] (s_msg tile) loaded + synced.
template <int RT, int PHB, int PF, class Sat>   // PHB: first phase-timing slot (debug builds); PF: pipelined GEMMs
__device__ __forceinline__ void update_forward(const LayerW &W, _Float16 *ldsh, const LaneGeo &L, UpdRegs<RT> &R, Sat &sat) {
    constexpr int TA = 16 * RT, OFF_VT = UpdLds<RT>::OFF_VT, OFF_HS = UpdLds<RT>::OFF_HS, OFF_AS = UpdLds<RT>::OFF_AS;
    const Planes vt = make_planes(ldsh + OFF_VT, 3 * TA, F), hs = make_planes(ldsh + OFF_HS, TA, 2 * F),
                 as_ = make_planes(ldsh + OFF_AS, TA, F);
    zero_acc(R.uv);
    PH_INIT
    {
        const uint4 *wp[2] = {WTILE(U, L.w, F), WTILE(V, L.w, F)};
        gemm16<F, 3 * RT, 2, PF>(vt, wp, R.uv);
    }
    PH(PHB + 1)
#pragma unroll
    for (int t = 0; t < RT; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float n2 = 0.f, in = 0.f;
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                const float vv = R.uv[RT * x + t][1][i];
                n2 += fmaf(vv, vv, 1e-15f);
                in = fmaf(R.uv[RT * x + t][0][i], vv, in);
            }
            R.nrm[t][i] = sqrtf(n2);
            R.inner[t][i] = in;
        }
        store_split4(hs, L.row(t), F + L.col0, R.nrm[t], sat);
    }
    PH(PHB + 2)
    __syncthreads();
    PH(PHB + 3)
    {
        f32x4 acc[RT][1];
        zero_acc(acc);
        const uint4 *wp[1] = {WTILE(W3, L.w, 2 * F)};
        const f32x4 b = gload4f(W.b3 + L.col0);
        __builtin_amdgcn_sched_barrier(0);   // the bias is requested in front of the GEMM whose epilogue adds it
        gemm16<2 * F, RT, 1, PF>(hs, wp, acc);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            R.h3[t] = acc[t][0] + b;
            f32x4 sw;
#pragma unroll
            for (int i = 0; i < 4; ++i) sw[i] = swish(R.h3[t][i]);
            store_split4(as_, L.row(t), L.col0, sw, sat);
        }
    }
    PH(PHB + 4)
    __syncthreads();
    PH(PHB + 5)
    {   // only the gates the reverse pass reads: a_vv and a_sv (a_ss enters the forward output alone)
        zero_acc(R.gate);
        const uint4 *wp[2] = {WTILE(W4, L.w, F), WTILE(W4, NW + L.w, F)};
        f32x4 b4[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) b4[c] = gload4f(W.b4 + c * F + L.col0);
        __builtin_amdgcn_sched_barrier(0);
        gemm16<F, RT, 2, PF>(as_, wp, R.gate);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const f32x4 b = b4[c];
#pragma unroll
            for (int t = 0; t < RT; ++t) R.gate[t][c] += b;
        }
    }
    PH(PHB + 6)
}

template <int RT, class Sat>
__device__ __forceinline__ void load_update_v(_Float16 *ldsh, const float *__restrict__ v_msg, size_t mN, int a0, int N, Sat &sat) {
    constexpr int TA = 16 * RT;
    const Planes vt = make_planes(ldsh + UpdLds<RT>::OFF_VT, 3 * TA, F);
    load_rows_split<3 * TA>(vt, 0, [&](int row) {
        int x = row / TA, a = min(a0 + (row % TA), N - 1);
        return v_msg + ((mN + a) * 3 + x) * F;
    }, sat);
}
template <int RT, class Sat>
__device__ __forceinline__ void load_update_s(_Float16 *ldsh, const float *__restrict__ s_msg, size_t mN, int a0, int N, Sat &sat) {
    constexpr int TA = 16 * RT;
    const Planes hs = make_planes(ldsh + UpdLds<RT>::OFF_HS, TA, 2 * F);
    load_rows_split<TA>(hs, 0, [&](int row) { return s_msg + (mN + min(a0 + row, N - 1)) * F; }, sat);
}

// Forward kernel, compact LDS layout:
//   [0, VT): v tile (h | l planes, 96 rows)  -- after GEMM1: |Vv| plane at 0, swish(h3) plane behind it, at the end the
//   [VT, VT + XS): s tile (32 rows)             fp32 output tile over everything
//   [CF_LDS_HALVES, + XS): planes of the block's scalar OUTPUT (TAIL = 1 only)
constexpr int CF_XS = plane_halves(3 * TA, F);                    // halves
constexpr int CF_LDS_HALVES = CF_XS + plane_halves(TA, F);        // 34 816 halves = 69 632 B
static_assert(2 * plane_halves(TA, F) <= CF_XS, "|Vv| and swish planes overlay the v tile");
static_assert(4 * TA * FT * sizeof(float) <= CF_LDS_HALVES * sizeof(_Float16), "output tile overlays the planes");
// One workgroup per CU with the 256-register budget: the fp32 input tile stays in the registers of the threads that loaded
// it and serves as the residual of the output pass (no second read of s_msg / v_msg), and the GEMMs run the software
// pipeline.  (A 128-register variant with two workgroups per CU re-read the residual and could not pipeline: 1.15 vs 1.09
// ms / step, profiles/r02/NOTES_node_kernels.md.)
// TAIL = 2: the block's VECTOR output is not produced (the last block: only s reaches the readout -- no a_vv gate, no v rows in
// the output pass: 1.5 KB per atom less to write).  TAIL = 1: the message MLP of the NEXT layer, phi = W2 swish(W1 s_out + b1) + b2 (weights of layer l + 1), runs as the
// kernel's tail on the scalar output tile while it is still on the chip: its planes are written by the output pass, the
// separate k_msg_mlp_mfma launch and its read of s disappear.
// Optional (VSSR_UPD_SAVE=1, off by default): forward intermediates handed to the reverse update kernel instead of being
// recomputed there -- per model, atom tile and thread UPD_SAVE_REGS f32x4: U v and V v (12), the gate MLP's pre-activation
// (2), a_vv and a_sv (4).  Measured on MI355X (B = 256, ms / step): the reverse kernel gains 0.24 (2.45 -> 2.21: three GEMMs
// and two tile loads less, 4.6 KB / atom more to read), the forward kernel loses 0.27 .. 0.30 (1.30 -> 1.57 with the stores
// where the values are born, 1.60 with all of them at the end): with one workgroup per CU the drain of 147 KB of stores per
// workgroup is as exposed as the loads they replace.  Energies bit-identical, forces to fp32 rounding (tests/test_gpu_parity.py).
#ifndef UPD_SAVE_LATE
#define UPD_SAVE_LATE 1
#endif
constexpr int UPD_SAVE_UV = 0, UPD_SAVE_H3 = 12, UPD_SAVE_GATE = 14, UPD_SAVE_REGS = 18;
__device__ __forceinline__ f32x4 &save_slot(f32x4 *save, int m, int k) {
    return save[(((size_t)m * gridDim.x + blockIdx.x) * UPD_SAVE_REGS + k) * NTHREADS + threadIdx.x];
}
size_t update_save_bytes(int N, int M) { return sizeof(f32x4) * (size_t)M * ((N + TA - 1) / TA) * UPD_SAVE_REGS * NTHREADS; }

template <int TAIL>
__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_update_fwd_mfma(int N, int l, ActiveView av, const ModelW *__restrict__ MW, const float *__restrict__ s_msg,
                  const float *__restrict__ v_msg, float *__restrict__ s_out, float *__restrict__ v_out,
                  float *__restrict__ phi_next, f32x4 *__restrict__ save) {
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
    constexpr bool NOV = TAIL == 2;   // last layer: only the scalar output is consumed (readout); v_out is not written
    const int m = blockIdx.y, a0 = blockIdx.x * TA;
    if (!av.tile(min(a0, N - 1), min(a0 + TA - 1, N - 1))) return;   // every chain of this atom tile is switched off
    const LaneGeo L;
    SatTrack sat;
    const LayerW &W = MW[m].layer[l];
    const size_t mN = (size_t)m * N;
    const Planes vt = make_planes(ldsh, 3 * TA, F), xs = make_planes(ldsh + CF_XS, TA, F);
    const Planes nr = make_planes(ldsh, TA, F), as_ = make_planes(ldsh + plane_halves(TA, F), TA, F);   // overlays of vt
    constexpr int PF = 1;
    float4 keep_v[3 * TA * (F / 4) / NTHREADS], keep_s[TA * (F / 4) / NTHREADS];
    PH_INIT
    // weight pieces of every GEMM are requested one phase ahead (gemm16_preload): here those of the first two
    const uint4 *wUV[2] = {WTILE(U, L.w, F), WTILE(V, L.w, F)};
    const uint4 *wW3a[1] = {WTILE(W3, L.w, 2 * F)};
    const uint4 *wW3b[1] = {WTILE(W3, L.w, 2 * F) + (F / 32) * 2 * 64};   // chunks F/32 .. 2F/32 - 1 of the same column tile
    const uint4 *wW4[3] = {WTILE(W4, L.w, F), WTILE(W4, NW + L.w, F), WTILE(W4, 2 * NW + L.w, F)};
    WPre<6, 2, 1> pUV;
    WPre<2, 1, 1> pW3a;
    // the tile first, the weight pieces behind it (loads return in order): they arrive while the tile is being split
    rows_request<3 * TA>([&](int row) {
        int x = row / TA, a = min(a0 + (row % TA), N - 1);
        return v_msg + ((mN + a) * 3 + x) * F;
    }, keep_v);
    rows_request<TA>([&](int row) { return s_msg + (mN + min(a0 + row, N - 1)) * F; }, keep_s);
    __builtin_amdgcn_sched_barrier(0);
    gemm16_preload(wUV, pUV);
    gemm16_preload(wW3a, pW3a);
    rows_store_split<3 * TA>(vt, 0, keep_v, sat);
    rows_store_split<TA>(xs, 0, keep_s, sat);
    __syncthreads();
    PH(0)
    f32x4 uv[6][2];   // [2 x + t][0] = U v_x, [..][1] = V v_x
    zero_acc(uv);
    gemm16<F, 6, 2, PF, 1>(vt, wUV, uv, &pUV);
    PH(1)
    f32x4 h3[2][1];   // gate MLP, first layer: the s half of [s ; |Vv|] now, the |Vv| half after the barrier
    zero_acc(h3);
    gemm16<F, 2, 1, PF, 1>(xs, wW3a, h3, &pW3a);
    WPre<2, 1, 2> pW3b;
    gemm16_preload(wW3b, pW3b);
    PH(2)
    __syncthreads();   // every wave is done with the v tile
    PH(3)
    f32x4 inner[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x4 nv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float n2 = 0.f, in = 0.f;
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                const float vv = uv[2 * x + t][1][i];
                n2 += fmaf(vv, vv, 1e-15f);
                in = fmaf(uv[2 * x + t][0][i], vv, in);
            }
            inner[t][i] = in;
            nv[i] = sqrtf(n2);
        }
        store_split4(nr, L.row(t), L.col0, nv, sat);
    }
    __syncthreads();
    PH(4)
    WPre<2, 3, 1> pW4;
    WPre<2, 2, 1> pW4n;   // NOV: only a_sv and a_ss
    const uint4 *wW4n[2] = {wW4[1], wW4[2]};
    {
        const f32x4 b = gload4f(W.b3 + L.col0);
        __builtin_amdgcn_sched_barrier(0);   // the bias is requested in front of the GEMM whose epilogue adds it
        gemm16<F, 2, 1, PF, 2>(nr, wW3b, h3, &pW3b);
        if constexpr (NOV) gemm16_preload(wW4n, pW4n); else gemm16_preload(wW4, pW4);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 sw = h3[t][0] + b;
            h3[t][0] = sw;   // pre-activation of the gate MLP, kept for the reverse kernel (save_slot below)
#pragma unroll
            for (int i = 0; i < 4; ++i) sw[i] = swish(sw[i]);
            store_split4(as_, L.row(t), L.col0, sw, sat);
        }
    }
    PH(5)
    __syncthreads();
    PH(6)
    f32x4 gate[2][3];   // a_vv, a_sv, a_ss
    zero_acc(gate);
    const uint4 *wW1n[1] = {nullptr}, *wW2n[3] = {nullptr, nullptr, nullptr};
    WPre<2, 1, 2> pW1n;
    {
        f32x4 b4[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) b4[c] = gload4f(W.b4 + c * F + L.col0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NOV) {   // the vector output is not wanted: a_vv is not needed, 24 matrix instructions less per wave
            f32x4 g2[2][2];
            zero_acc(g2);
            gemm16<F, 2, 2, PF, 1>(as_, wW4n, g2, &pW4n);
#pragma unroll
            for (int t = 0; t < 2; ++t) { gate[t][1] = g2[t][0]; gate[t][2] = g2[t][1]; }
        } else {
            gemm16<F, 2, 3, PF, 1>(as_, wW4, gate, &pW4);
        }
        if constexpr (TAIL == 1) {
            const LayerW &Wn = MW[m].layer[l + 1];
            wW1n[0] = Wn.qW1 + (size_t)L.w * 4 * F;
#pragma unroll
            for (int c = 0; c < 3; ++c) wW2n[c] = Wn.qW2 + (size_t)(c * NW + L.w) * 4 * F;
            gemm16_preload(wW1n, pW1n);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int t = 0; t < 2; ++t) gate[t][c] += b4[c];
        }
    }
    // What the reverse update kernel needs from this forward pass, in ACCUMULATOR order (both kernels cut the atoms into the
    // same tiles and lanes: a coalesced 8 KB store per register and workgroup, read back the same way; 4.6 KB / atom).
    auto save_all = [&]() {
        if (!save) return;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) save_slot(save, m, UPD_SAVE_UV + 2 * j + c) = uv[j][c];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            save_slot(save, m, UPD_SAVE_H3 + t) = h3[t][0];
#pragma unroll
            for (int c = 0; c < 2; ++c) save_slot(save, m, UPD_SAVE_GATE + 2 * t + c) = gate[t][c];
        }
    };
#if UPD_SAVE_LATE == 0
    save_all();
#endif
    PH(7)
    __syncthreads();   // every wave is done with the planes: the region becomes the fp32 output tile
    PH(8)
    // increments (rows [0, TA): s, rows TA (1 + x) + atom: v_x); the residual is added in the coalesced pass
    float *T = reinterpret_cast<float *>(ldsh);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = L.row(t);
        f32x4 ds;
#pragma unroll
        for (int i = 0; i < 4; ++i) ds[i] = fmaf(gate[t][1][i], inner[t][i], gate[t][2][i]);
        *reinterpret_cast<f32x4 *>(T + row * FT + L.col0) = ds;
        if constexpr (!NOV) {
#pragma unroll
            for (int x = 0; x < 3; ++x) *reinterpret_cast<f32x4 *>(T + (TA * (1 + x) + row) * FT + L.col0) = gate[t][0] * uv[2 * x + t][0];
        }
    }
    __syncthreads();
    auto gofs = [&](int row, int col) -> size_t {   // q = row / TA: 0 = s, 1..3 = v_x, v_y, v_z (tail rows clamped to the last atom)
        const int q = row / TA, a = min(a0 + row % TA, N - 1);
        return q == 0 ? (mN + a) * F + col : ((mN + a) * 3 + (q - 1)) * F + col;
    };
    PH(9)
    const Planes xo = make_planes(ldsh + CF_LDS_HALVES, TA, F);   // TAIL: planes of s_out behind the output tile
    auto put = [&](int row, int col, const float4 &d, const float4 &r) {
        const float4 o = make_float4(r.x + d.x, r.y + d.y, r.z + d.z, r.w + d.w);
        if (TAIL == 1 && row < TA) store_split4<SAT_COOP>(xo, row, col, o, sat);   // (rows past the last atom: clamped copies, never stored)
        if (a0 + row % TA >= N) return;
        if (row >= TA && !v_out) return;   // (last block with the stored-intermediates option: a_vv is needed, the vector output is not)
        *reinterpret_cast<float4 *>((row < TA ? s_out : v_out) + gofs(row, col)) = o;
    };
    {   // the loading pass and this pass map (thread, iteration) to (row, column) identically: rows [0, TA) = s
        constexpr int NS = TA * (F / 4) / NTHREADS, NV = 3 * TA * (F / 4) / NTHREADS;
#pragma unroll
        for (int it = 0; it < (NOV ? NS : NS + NV); ++it) {
            const int idx = threadIdx.x + it * NTHREADS, row = idx >> 5, c4 = idx & 31;
            put(row, 4 * c4, *reinterpret_cast<const float4 *>(T + row * FT + 4 * c4), it < NS ? keep_s[it < NS ? it : 0] : keep_v[it < NS ? 0 : it - NS]);
        }
    }
    PH(10)
    if constexpr (TAIL == 1) {
        const LayerW &Wn = MW[m].layer[l + 1];
        const Planes hn = make_planes(ldsh, TA, F);   // swish(W1 s + b1), over the (finished) output tile
        __syncthreads();   // s_out planes complete; every wave is done with the output tile
        WPre<2, 3, 1> pW2n;
        {
            f32x4 acc[2][1];
            zero_acc(acc);
            const f32x4 b = gload4f(Wn.b1 + L.col0);
            __builtin_amdgcn_sched_barrier(0);   // the bias is requested in front of the GEMM whose epilogue adds it
            gemm16<F, 2, 1, 1, 2>(xo, wW1n, acc, &pW1n);
            gemm16_preload(wW2n, pW2n);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 hv = acc[t][0] + b;
#pragma unroll
                for (int i = 0; i < 4; ++i) hv[i] = swish(hv[i]);
                store_split4(hn, L.row(t), L.col0, hv, sat);
            }
        }
        __syncthreads();
        f32x4 acc[2][3];
        zero_acc(acc);
        f32x4 b2[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) b2[c] = gload4f(Wn.b2 + c * F + L.col0);
        __builtin_amdgcn_sched_barrier(0);
        gemm16<F, 2, 3, 1, 1>(hn, wW2n, acc, &pW2n);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f32x4 b = b2[c];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int a = a0 + L.row(t);
                if (a < N) *reinterpret_cast<f32x4 *>(phi_next + (mN + a) * F3 + c * F + L.col0) = acc[t][c] + b;
            }
        }
        PH(11)
    }
#if UPD_SAVE_LATE
    save_all();   // behind everything that waits on vmcnt: loads queued behind these stores would wait for them as well
#endif
    sat.commit(av, a0, N);
}

// ---- readout on the matrix pipe ---------------------------------------------------------------
// e_i = w6 . swish(W5 s_i + b5) + b6 (+ excluded volume) ; sbar_i = W5^T (w6 * swish'(W5 s_i + b5)).  Needs the s tile in
// `xs` (planes, synced).  hp: planes [TA][RH] for the hidden adjoint; red: [RH / 16][TA] floats.  The same code serves the
// energy-only kernel and the head of the fused reverse kernel, so both produce identical per-atom energies.
constexpr int RH = 64;   // hidden width of the compiled matrix-pipe readout (other widths use k_readout, painn.hip)
template <int RT, bool WANT_SBAR, class Sat>
__device__ __forceinline__ void readout_head(const ModelW &W, const Planes &xs, const Planes &hp, float *red, const LaneGeo &L,
                                             int a0, int N, size_t mN, const float *__restrict__ e_excl,
                                             float *__restrict__ e_atom, f32x4 (&sb)[RT], Sat &sat) {
    constexpr int NCW = RH / 16;   // waves that own a column tile of the hidden layer
    constexpr int TA = 16 * RT;
    const int lane = threadIdx.x & 63;
    if (L.w < NCW) {   // wave-uniform
        f32x4 acc[RT][1];
        zero_acc(acc);
        const uint4 *wp[1] = {W.qW5 + (size_t)L.w * 4 * F};
        const f32x4 b = gload4f(W.b5 + L.col0), w6 = gload4f(W.w6 + L.col0);
        __builtin_amdgcn_sched_barrier(0);   // the bias is requested in front of the GEMM whose epilogue adds it
        gemm16<F, RT, 1>(xs, wp, acc);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const f32x4 hval = acc[t][0] + b;
            f32x4 hb;
            float es = 0.f;   // the tile's 16 hidden units of atom row(t): 4 in this lane, then the 4 lane groups, fixed order
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                es = fmaf(w6[i], swish(hval[i]), es);
                hb[i] = w6[i] * dswish(hval[i]);
            }
            if (WANT_SBAR) store_split4(hp, L.row(t), L.col0, hb, sat);
            es = xrow_sum_f32(es);   // gfx950 row swaps (vssr_internal.h) instead of two dependent LDS permutes
            if (lane < 16) red[L.w * TA + L.row(t)] = es;
        }
    }
    __syncthreads();
    if (threadIdx.x < TA) {
        const int atom = a0 + threadIdx.x;
        if (atom < N) {
            float e = gload1f(W.b6);   // (a plain W.b6[0] is a FLAT load: the pointer comes out of the ModelW table)
#pragma unroll
            for (int k = 0; k < NCW; ++k) e += red[k * TA + threadIdx.x];
            if (e_excl) e += e_excl[atom];
            e_atom[mN + atom] = e;
        }
    }
    if (WANT_SBAR) {
        f32x4 acc[RT][1];
        zero_acc(acc);
        const uint4 *wp[1] = {W.qW5t + (size_t)L.w * 4 * RH};
        gemm16<RH, RT, 1>(hp, wp, acc);
#pragma unroll
        for (int t = 0; t < RT; ++t) sb[t] = acc[t][0];
    }
}

// energy-only evaluations: readout of the final scalar features
__global__ void __launch_bounds__(NTHREADS)
k_readout_mfma(int N, ActiveView av, const ModelW *__restrict__ MW, const float *__restrict__ s, const float *__restrict__ e_excl,
               float *__restrict__ e_atom) {
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
    const int m = blockIdx.y, a0 = blockIdx.x * TA;
    if (!av.tile(min(a0, N - 1), min(a0 + TA - 1, N - 1))) return;   // every chain of this atom tile is switched off
    const LaneGeo L;
    SatTrack sat;
    const size_t mN = (size_t)m * N;
    const Planes xs = make_planes(ldsh, TA, F), hp = make_planes(ldsh + plane_halves(TA, F), TA, RH);
    float *red = reinterpret_cast<float *>(ldsh + plane_halves(TA, F) + plane_halves(TA, RH));
    load_rows_split<TA>(xs, 0, [&](int row) { return s + (mN + min(a0 + row, N - 1)) * F; }, sat);
    __syncthreads();
    f32x4 sb[2];
    readout_head<2, false>(MW[m], xs, hp, red, L, a0, N, mN, e_excl, e_atom, sb, sat);
    sat.commit(av, a0, N);
}

// reverse: (sbar, vbar) of the block outputs -> (sbar_msg, vbar_msg) of its inputs.
// Adjoint algebra as in the oracle (oracle/painn_impl.inc, "update block^T"):
//   abar_vv = sum_x vbar_x Uv_x ; qbar = [abar_vv, sbar*inner, sbar]
//   h3bar = (W4^T qbar) * swish'(h3) ; [sbar_extra ; nbar] = W3^T h3bar
//   Ubar_x = vbar_x a_vv + sbar a_sv Vv_x ; Vbar_x = sbar a_sv Uv_x + nbar Vv_x / |Vv|
//   vbar_msg = vbar + U^T Ubar + V^T Vbar ; sbar_msg = sbar + sbar_extra
// The producer of sbar is fused in as the kernel's HEAD (the result is born in the accumulator layout the body wants, so it
// never travels through HBM and the scalar, uncoalesced sbar reads / writes of separate kernels disappear):
//   MODE 0: sbar is read from memory (readout widths other than RH);
//   MODE 1: last layer -- the readout and its reverse (e_atom is written here; vbar is zero by construction);
//   MODE 2: layer l < L - 1 -- the reverse of the message MLP of layer l + 1:
//           sbar = sbar_msg(l+1) + W1^T[(W2^T phibar) * swish'(W1 s_in(l+1) + b1)]   (weights of layer l + 1)
// s_next: MODE 1: final scalar features, MODE 2: s_in(l+1); sbar_src: MODE 0: sbar, MODE 2: sbar_msg(l+1) (a different
// buffer than the sbar_msg this launch writes).
// SAVED: the forward intermediates come from the buffer update_fwd wrote (save_slot) instead of being recomputed from
// (s_msg, v_msg): no tile loads of s_msg / v_msg, no U / V / W3 / W4 GEMMs here.
template <int MODE, int RT, bool SAVED = false>
__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(RT == 1 ? 4 : 2, RT == 1 ? 4 : 2)))
k_update_bwd_mfma(int N, int l, int vbar_is_zero, ActiveView av, const ModelW *__restrict__ MW, const float *__restrict__ s_msg,
                  const float *__restrict__ v_msg, const float *__restrict__ sbar_src, const float *__restrict__ vbar,
                  const float *__restrict__ s_next, const float *__restrict__ phibar, const float *__restrict__ e_excl,
                  float *__restrict__ e_atom, float *__restrict__ sbar_msg, float *__restrict__ vbar_msg,
                  const f32x4 *__restrict__ save) {
    static_assert(!SAVED || RT == 2, "the saved layout is that of the 32-atom forward kernel");
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
    constexpr int TA = 16 * RT, OFF_VT = UpdLds<RT>::OFF_VT, OFF_AS = UpdLds<RT>::OFF_AS, UPD_LDS_HALVES = UpdLds<RT>::HALVES;
    const int m = blockIdx.y, a0 = blockIdx.x * TA;
    if (!av.tile(min(a0, N - 1), min(a0 + TA - 1, N - 1))) return;   // every chain of this atom tile is switched off
    const LaneGeo L;
    // MODE 0 is the cold instantiation (readout widths other than RH): the pipelined GEMMs cost it 4 spilled registers, the rolled
    // ones none; the hot modes keep the pipeline
    constexpr int BPF = MODE == 0 ? 0 : UPD_PF;
    // Saturation watch (mfma16.h).  s_msg, v_msg and the recomputed forward intermediates were watched by update_fwd(l) of this
    // evaluation, s_in(l + 1) by its tail: SatNone.  What is new here are the adjoints: the cooperative loads (phibar; final s for
    // the readout) and every epilogue that stores adjoints use an instance of their own, committed right behind the pass: no
    // tracker register lives across a GEMM (the kernel runs at the 256-register limit: two registers across it cost 0.13 ms / step).
    SatNone unwatched;
    const LayerW &W = MW[m].layer[l];
    const size_t mN = (size_t)m * N;
    PH_INIT
    f32x4 sb[RT];   // sbar in the accumulator layout
    if (MODE != 1) {   // requested first, needed after the head / the forward recomputation
#pragma unroll
        for (int t = 0; t < RT; ++t) sb[t] = gload4f(sbar_src + (mN + min(a0 + L.row(t), N - 1)) * F + L.col0);
    }
    // vbar of the tile (96 rows, fp32) stays in LDS behind the planes for the whole kernel: it is needed three times in the
    // accumulator layout (an atom row and four columns per lane): one ds_read_b128 each instead of global round trips
    float *VB = reinterpret_cast<float *>(ldsh + UPD_LDS_HALVES);   // [x * TA + atom][FT]
    if (MODE == 1) vbar_is_zero = 1;   // (the region hosts the readout scratch)
    if (!vbar_is_zero) {
        constexpr int NIT = 3 * TA * (F / 4) / NTHREADS;
        float4 vv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = threadIdx.x + it * NTHREADS, row = idx >> 5, c4 = idx & 31;
            vv[it] = *reinterpret_cast<const float4 *>(vbar + ((mN + min(a0 + row % TA, N - 1)) * 3 + row / TA) * F + 4 * c4);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = threadIdx.x + it * NTHREADS, row = idx >> 5, c4 = idx & 31;
            *reinterpret_cast<float4 *>(VB + row * FT + 4 * c4) = vv[it];
        }
    }
    UpdRegs<RT> R;
    auto load_saved = [&]() {   // requested in front of the head, consumed behind it
        const f32x4 *sv = save + (((size_t)m * gridDim.x + blockIdx.x) * UPD_SAVE_REGS) * NTHREADS + threadIdx.x;
#pragma unroll
        for (int j = 0; j < 3 * RT; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) R.uv[j][c] = gload4f(reinterpret_cast<const float *>(sv + (size_t)(UPD_SAVE_UV + 2 * j + c) * NTHREADS));
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            R.h3[t] = gload4f(reinterpret_cast<const float *>(sv + (size_t)(UPD_SAVE_H3 + t) * NTHREADS));
#pragma unroll
            for (int c = 0; c < 2; ++c) R.gate[t][c] = gload4f(reinterpret_cast<const float *>(sv + (size_t)(UPD_SAVE_GATE + 2 * t + c) * NTHREADS));
        }
    };
    if (!SAVED) load_update_s<RT>(ldsh, s_msg, mN, a0, N, unwatched);
    else load_saved();
    if (MODE == 0) {
        load_update_v<RT>(ldsh, v_msg, mN, a0, N, unwatched);
        __syncthreads();
    } else if (MODE == 1) {
        // head: readout of s_next (tile in the `as` region; hidden adjoint + reduction scratch in the unused vbar region)
        if (!SAVED) load_update_v<RT>(ldsh, v_msg, mN, a0, N, unwatched);
        const Planes xs = make_planes(ldsh + OFF_AS, TA, F);
        const Planes hp = make_planes(ldsh + UPD_LDS_HALVES, TA, RH);
        float *red = reinterpret_cast<float *>(ldsh + UPD_LDS_HALVES + plane_halves(TA, RH));
        {
            SatTrack satc;   // (the last layer's output is not split by update_fwd<0>)
            load_rows_split<TA>(xs, 0, [&](int row) { return s_next + (mN + min(a0 + row, N - 1)) * F; }, satc);
            satc.commit(av, a0, N);
        }
        __syncthreads();
        PH(40)
        {
            SatTrack sate;
            readout_head<RT, true>(MW[m], xs, hp, red, L, a0, N, mN, e_excl, e_atom, sb, sate);
            sate.commit(av, a0, N);
        }
        PH(41)
        // (the body's first barrier orders the last reads of xs before anything overwrites the `as` region)
    } else {
        // head: reverse of the message MLP of layer l + 1.  phibar tile over the (not yet loaded) v tile, s_next / h1bar in
        // the `as` region; the v tile follows once the head is done with the region.
        const LayerW &Wn = MW[m].layer[l + 1];
        const Planes xs = make_planes(ldsh + OFF_AS, TA, F), pb = make_planes(ldsh + OFF_VT, TA, F3);
        load_rows_split<TA>(xs, 0, [&](int row) { return s_next + (mN + min(a0 + row, N - 1)) * F; }, unwatched);
        {
            SatTrack satc;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                load_rows_split<TA>(pb, c * F, [&](int row) { return phibar + (mN + min(a0 + row, N - 1)) * F3 + c * F; }, satc);
            satc.commit(av, a0, N);
        }
        __syncthreads();
        PH(42)
        f32x4 h1[RT][1], a1[RT][1];
        zero_acc(h1);
        zero_acc(a1);
        const f32x4 b1n = gload4f(Wn.b1 + L.col0);
        __builtin_amdgcn_sched_barrier(0);
        {
            const uint4 *wp[1] = {Wn.qW1 + (size_t)L.w * 4 * F};
            gemm16<F, RT, 1, BPF>(xs, wp, h1);
            const uint4 *wq[1] = {Wn.qW2t + (size_t)L.w * 4 * F3};
            gemm16<F3, RT, 1, BPF>(pb, wq, a1);
        }
        PH(43)
        __syncthreads();   // everyone is done reading xs and pb
        if (!SAVED) load_update_v<RT>(ldsh, v_msg, mN, a0, N, unwatched);   // the v tile takes the place of the phibar tile
        PH(44)
        {
            const f32x4 b = b1n;
            SatTrack sate;
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                f32x4 hv;
#pragma unroll
                for (int i = 0; i < 4; ++i) hv[i] = a1[t][0][i] * dswish(h1[t][0][i] + b[i]);
                store_split4(xs, L.row(t), L.col0, hv, sate);
            }
            sate.commit(av, a0, N);
        }
        __syncthreads();
        f32x4 acc[RT][1];
        zero_acc(acc);
        const uint4 *wp[1] = {Wn.qW1t + (size_t)L.w * 4 * F};
        gemm16<F, RT, 1, BPF>(xs, wp, acc);
#pragma unroll
        for (int t = 0; t < RT; ++t) sb[t] += acc[t][0];
        PH(45)
        // (the body's first barrier orders the last reads of xs before anything overwrites the `as` region)
    }
    PH(16)
    if constexpr (!SAVED) {
        update_forward<RT, 16, BPF>(W, ldsh, L, R, unwatched);
    } else {
        // <U v, V v> of the saved products (same operations, same order as update_forward: bit-identical)
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float in = 0.f;
#pragma unroll
                for (int x = 0; x < 3; ++x) in = fmaf(R.uv[RT * x + t][0][i], R.uv[RT * x + t][1][i], in);
                R.inner[t][i] = in;
            }
        __syncthreads();   // every wave is done with the head's tiles (the planes below overlay them)
    }
    PH_RESET
    // Every wave has passed the barrier in front of GEMM3, i.e. finished GEMM1/GEMM2: vt and hs are free.
    const Planes qb = make_planes(ldsh + OFF_VT, TA, F3);    // overlays vt
    SatTrack sat;   // (scoped to this epilogue; the next one starts afresh)
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int row = L.row(t);
        f32x4 abar_vv = {0.f, 0.f, 0.f, 0.f};
        if (!vbar_is_zero) {   // wave-uniform
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                const f32x4 vb = *reinterpret_cast<const f32x4 *>(VB + (x * TA + row) * FT + L.col0);
#pragma unroll
                for (int i = 0; i < 4; ++i) abar_vv[i] = fmaf(vb[i], R.uv[RT * x + t][0][i], abar_vv[i]);
            }
        }
        store_split4(qb, row, L.col0, abar_vv, sat);
        store_split4(qb, row, F + L.col0, sb[t] * R.inner[t], sat);
        store_split4(qb, row, 2 * F + L.col0, sb[t], sat);
    }
    sat.commit(av, a0, N);
    sat = SatTrack();
    PH(23)
    __syncthreads();   // qb complete; every wave is past GEMM3, so `as` may be overwritten
    PH(24)
    const Planes hb = make_planes(ldsh + OFF_AS, TA, F);    // h3bar
    {
        f32x4 acc[RT][1];
        zero_acc(acc);
        const uint4 *wp[1] = {WTILE(W4t, L.w, F3)};
        gemm16<F3, RT, 1, BPF>(qb, wp, acc);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            f32x4 hv;
#pragma unroll
            for (int i = 0; i < 4; ++i) hv[i] = acc[t][0][i] * dswish(R.h3[t][i]);
            store_split4(hb, L.row(t), L.col0, hv, sat);
        }
        sat.commit(av, a0, N);
        sat = SatTrack();
    }
    PH(25)
    __syncthreads();
    PH(26)
    f32x4 hbar[RT][2];   // [t][0] = d/d s_msg part, [t][1] = d/d norm part, both for this lane's four features
    zero_acc(hbar);
    {
        const uint4 *wp[2] = {WTILE(W3t, L.w, F), WTILE(W3t, NW + L.w, F)};
        gemm16<F, RT, 2, BPF>(hb, wp, hbar);
    }
    PH(27)
    __syncthreads();   // all waves are done with qb / hb: the whole region becomes the [Ubar | Vbar] tile
    PH(28)
    const Planes ab = make_planes(ldsh, 3 * TA, 2 * F);     // cols [0,F) = Ubar, [F,2F) = Vbar
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int row = L.row(t);
        f32x4 sc, sa;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // |Vv| is recomputed from Vv (same operations, same order as the forward part: bit-identical) instead of being
            // carried in 8 registers across the W4^T / W3^T phases -- this kernel runs at the 256-register limit
            float n2 = 0.f;
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                const float vv = R.uv[RT * x + t][1][i];
                n2 += fmaf(vv, vv, 1e-15f);
            }
            sc[i] = hbar[t][1][i] / sqrtf(n2);
            sa[i] = sb[t][i] * R.gate[t][1][i];
        }
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            f32x4 vbo = {0.f, 0.f, 0.f, 0.f};
            if (!vbar_is_zero) vbo = *reinterpret_cast<const f32x4 *>(VB + (x * TA + row) * FT + L.col0);
            f32x4 ub, vb2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float u = R.uv[RT * x + t][0][i], v = R.uv[RT * x + t][1][i];
                ub[i] = fmaf(vbo[i], R.gate[t][0][i], sa[i] * v);
                vb2[i] = fmaf(sa[i], u, sc[i] * v);
            }
            store_split4(ab, x * TA + row, L.col0, ub, sat);
            store_split4(ab, x * TA + row, F + L.col0, vb2, sat);
        }
    }
    sat.commit(av, a0, N);
    PH(29)
    __syncthreads();
    PH(30)
    f32x4 out[3 * RT][1];
    zero_acc(out);
    {
        const uint4 *wp[1] = {WTILE(UVt, L.w, 2 * F)};
        gemm16<2 * F, 3 * RT, 1, BPF>(ab, wp, out);
    }
    PH(31)
    __syncthreads();   // every wave is done with the [Ubar | Vbar] planes: the region becomes the fp32 output tile
    PH(32)
    float *T = reinterpret_cast<float *>(ldsh);   // rows [0, TA): sbar_msg, rows TA (1 + x) + atom: increment of vbar_msg_x
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int row = L.row(t);
        *reinterpret_cast<f32x4 *>(T + row * FT + L.col0) = sb[t] + hbar[t][0];
#pragma unroll
        for (int x = 0; x < 3; ++x) *reinterpret_cast<f32x4 *>(T + (TA * (1 + x) + row) * FT + L.col0) = out[RT * x + t][0];
    }
    __syncthreads();
    stage_rows<4 * TA>(T, [&](int row, int col, const float4 &d) {
        const int q = row / TA, a = a0 + row % TA;
        if (a >= N) return;
        if (q == 0) {
            *reinterpret_cast<float4 *>(sbar_msg + (mN + a) * F + col) = d;
        } else {
            const size_t off = ((mN + a) * 3 + (q - 1)) * F + col;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!vbar_is_zero) r = *reinterpret_cast<const float4 *>(VB + (row - TA) * FT + col);
            *reinterpret_cast<float4 *>(vbar_msg + off) = make_float4(r.x + d.x, r.y + d.y, r.z + d.z, r.w + d.w);
        }
    });
    PH(33)
}

// ---- host: weight packing + launch helpers ------------------------------------------------------------------------
// fp16-split fragment order for v_mfma_f32_16x16x32_f16, 16-column tiles:
//   dst[tile][q][piece][lane][j] = piece(W[tile*16 + (lane&15)][32 q + 8 (lane>>4) + 2 j]) | piece(W[..][.. + 1]) << 16
void pack_mfma_tiles16(const float *Wsrc, int rows, int K, unsigned *dst) {
    auto split2 = [](float x, unsigned (&p)[2]) {
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        unsigned short hb, lb;
        memcpy(&hb, &h, 2);
        memcpy(&lb, &l, 2);
        p[0] = hb; p[1] = lb;
    };
    const int ntile = rows / 16, nq = K / 32;
    for (int tile = 0; tile < ntile; ++tile)
        for (int q = 0; q < nq; ++q)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const size_t row = (size_t)tile * 16 + (lane & 15);
                    const int k = 32 * q + 8 * (lane >> 4) + 2 * j;
                    unsigned p0[2], p1[2];
                    split2(Wsrc[row * K + k], p0);
                    split2(Wsrc[row * K + k + 1], p1);
                    for (int pc = 0; pc < 2; ++pc)
                        dst[((((size_t)tile * nq + q) * 2 + pc) * 64 + lane) * 4 + j] = p0[pc] | (p1[pc] << 16);
                }
}

size_t node_mfma_lds_bytes(int which) {
    switch (which) {
        case 0: return sizeof(_Float16) * 2 * plane_halves(TA, F);                        // msg mlp fwd
        case 1: return sizeof(_Float16) * (plane_halves(TA, F) + plane_halves(TA, F3));   // msg mlp bwd
        case 4: return sizeof(_Float16) * (plane_halves(TA, F) + plane_halves(TA, RH)) + sizeof(float) * (RH / 16) * TA;   // readout
        case 6: return 96 * 1024;   // update fwd: compact layout + the s_out planes of the fused tail (87 040 B), one workgroup per CU
        default: return sizeof(_Float16) * UpdLds<2>::HALVES + sizeof(float) * 3 * TA * FT;  // update bwd: planes + fp32 vbar tile
    }
}
static_assert(sizeof(_Float16) * plane_halves(16, RH) + sizeof(float) * (RH / 16) * 16 <= sizeof(float) * 3 * 16 * FT,
              "readout scratch fits the vbar region");
// (RT = 1 -- 16-atom tiles, two workgroups per CU -- was measured at -5 % for this kernel without the pipelined GEMMs and
// spills with them; only RT = 2 is instantiated.  profiles/r02/NOTES_node_kernels.md)

static_assert(sizeof(_Float16) * (CF_LDS_HALVES + plane_halves(TA, F)) <= 96 * 1024, "update fwd LDS request");

bool readout_mfma_supported(int hidden) { return hidden == RH; }

int node_mfma_init(vssr_handle *h) {
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_msg_mlp_mfma<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_msg_mlp_bwd_mfma, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)node_mfma_lds_bytes(1)));
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_update_fwd_mfma<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)node_mfma_lds_bytes(6)));
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_update_fwd_mfma<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)node_mfma_lds_bytes(6)));
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_update_fwd_mfma<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)node_mfma_lds_bytes(6)));
#define SET_UPD(MODE)                                                                                                     \
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_update_bwd_mfma<MODE, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)node_mfma_lds_bytes(3)));
    SET_UPD(0) SET_UPD(1) SET_UPD(2)
#undef SET_UPD
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_update_bwd_mfma<1, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)node_mfma_lds_bytes(3)));
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_update_bwd_mfma<2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)node_mfma_lds_bytes(3)));
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_readout_mfma, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)node_mfma_lds_bytes(4)));
    return VSSR_OK;
}

void launch_msg_mlp_mfma(hipStream_t st, int N, int M, int l, const ActiveView &av, const ModelW *MW, const float *s_in,
                         float *phi) {
    // (64-atom tiles and pipelined GEMMs were measured here in round 4 -- the row-scaling experiment, profiles/r04/NOTES_node_rows.md:
    //  -3.3 % on this kernel -- and are not built any more)
    hipLaunchKernelGGL((k_msg_mlp_mfma<2, 0>), dim3((N + TA - 1) / TA, M), dim3(NTHREADS), node_mfma_lds_bytes(0), st, N, l, av, MW,
                       s_in, phi);
}
void launch_msg_mlp_bwd_mfma(hipStream_t st, int N, int M, int l, const ActiveView &av, const ModelW *MW, const float *s_in,
                             const float *phibar, const float *sbar_msg, float *sbar_in) {
    hipLaunchKernelGGL(k_msg_mlp_bwd_mfma, dim3((N + TA - 1) / TA, M), dim3(NTHREADS), node_mfma_lds_bytes(1), st, N, l, av, MW,
                       s_in, phibar, sbar_msg, sbar_in);
}
// phi_next != nullptr: also phi of layer l + 1 (the fused message MLP; layer l + 1 must exist)
// save != nullptr: also the intermediates for the reverse kernel (update_save_bytes)
void launch_update_fwd_mfma(hipStream_t st, int N, int M, int l, const ActiveView &av, const ModelW *MW, const float *s_msg,
                            const float *v_msg, float *s_out, float *v_out, float *phi_next, void *save) {
    f32x4 *sv = reinterpret_cast<f32x4 *>(save);
    if (phi_next)
        hipLaunchKernelGGL(k_update_fwd_mfma<1>, dim3((N + TA - 1) / TA, M), dim3(NTHREADS), node_mfma_lds_bytes(6), st, N, l,
                           av, MW, s_msg, v_msg, s_out, v_out, phi_next, sv);
    else if (!v_out && !sv)   // (the stored-intermediates option needs a_vv)
        hipLaunchKernelGGL(k_update_fwd_mfma<2>, dim3((N + TA - 1) / TA, M), dim3(NTHREADS), node_mfma_lds_bytes(6), st, N, l,
                           av, MW, s_msg, v_msg, s_out, v_out, phi_next, sv);
    else
        hipLaunchKernelGGL(k_update_fwd_mfma<0>, dim3((N + TA - 1) / TA, M), dim3(NTHREADS), node_mfma_lds_bytes(6), st, N, l,
                           av, MW, s_msg, v_msg, s_out, v_out, phi_next, sv);
}
void launch_readout_mfma(hipStream_t st, int N, int M, const ActiveView &av, const ModelW *MW, const float *s,
                         const float *e_excl, float *e_atom) {
    hipLaunchKernelGGL(k_readout_mfma, dim3((N + TA - 1) / TA, M), dim3(NTHREADS), node_mfma_lds_bytes(4), st, N, av, MW, s,
                       e_excl, e_atom);
}
// mode 0: sbar_src = sbar ; mode 1: s_next = final s, e_excl (or null), e_atom ; mode 2: s_next = s_in(l+1), phibar,
// sbar_src = sbar_msg(l+1)
void launch_update_bwd_mfma(hipStream_t st, int N, int M, int l, int mode, int vbar_is_zero, const ActiveView &av, const ModelW *MW,
                            const float *s_msg, const float *v_msg, const float *sbar_src, const float *vbar,
                            const float *s_next, const float *phibar, const float *e_excl, float *e_atom,
                            float *sbar_msg, float *vbar_msg, const void *save) {
    const f32x4 *sv = reinterpret_cast<const f32x4 *>(save);
    if (mode == 1) vbar_is_zero = 1;
#if defined(UPD_BWD_RT) && UPD_BWD_RT == 1
    // A/B build only (tools/build_variant.sh rt1 -DUPD_BWD_RT=1 [-DUPD_PF=0], profiles/r05/NOTES_node_occupancy.md): 16-atom tiles,
    // 8 waves at 128 registers, two workgroups per CU.  Never shipped: it spills, and the saturation watch attributes rows modulo 32.
    if (!sv && mode != 0) {
        const dim3 grid1((N + 15) / 16, M), blk1(NTHREADS);
        const size_t lds1 = sizeof(_Float16) * UpdLds<1>::HALVES + sizeof(float) * 3 * 16 * FT;
        static bool once = [] {
            (void)hipFuncSetAttribute((const void *)k_update_bwd_mfma<1, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            (void)hipFuncSetAttribute((const void *)k_update_bwd_mfma<2, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            return true;
        }();
        (void)once;
        if (mode == 1)
            hipLaunchKernelGGL((k_update_bwd_mfma<1, 1, false>), grid1, blk1, lds1, st, N, l, vbar_is_zero, av, MW, s_msg, v_msg, sbar_src,
                               vbar, s_next, phibar, e_excl, e_atom, sbar_msg, vbar_msg, sv);
        else
            hipLaunchKernelGGL((k_update_bwd_mfma<2, 1, false>), grid1, blk1, lds1, st, N, l, vbar_is_zero, av, MW, s_msg, v_msg, sbar_src,
                               vbar, s_next, phibar, e_excl, e_atom, sbar_msg, vbar_msg, sv);
        return;
    }
#endif
    const dim3 grid((N + TA - 1) / TA, M), blk(NTHREADS);
    const size_t lds = node_mfma_lds_bytes(3);
#define LAUNCH_UPD(MODE, SAVED)                                                                                             \
    hipLaunchKernelGGL((k_update_bwd_mfma<MODE, 2, SAVED>), grid, blk, lds, st, N, l, vbar_is_zero, av, MW, s_msg, v_msg, sbar_src, \
                       vbar, s_next, phibar, e_excl, e_atom, sbar_msg, vbar_msg, sv)
    if (mode == 1) { if (sv) LAUNCH_UPD(1, true); else LAUNCH_UPD(1, false); }
    else if (mode == 2) { if (sv) LAUNCH_UPD(2, true); else LAUNCH_UPD(2, false); }
    else LAUNCH_UPD(0, false);
#undef LAUNCH_UPD
}

}  // namespace vssr
