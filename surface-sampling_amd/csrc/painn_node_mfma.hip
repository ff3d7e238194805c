// painn_node_mfma.hip — per-atom ("node") stages of PaiNN on the gfx950 matrix cores.
//
// The node stages are chains of small dense layers applied to every
// atom of every chain and ensemble member: genuinely dense GEMMs with M = atoms (x3 Cartesian rows for
// U/V), K, N in {128, 256, 384}.  gfx950 has no fp32-rate-above-vector matrix path (v_mfma_f32_32x32x2_f32 runs at the
// fp32 vector rate, and measurably conflicts with the VALU; no xf32 exists), so the GEMMs run on the 16-bit matrix pipe
// with fp32-level accuracy: both operands are split into two fp16 pieces x = h + l (h = fp16(x), l = fp16(x - h): 22
// mantissa bits; fp16 subnormals are honoured by the matrix core) and the three products a_h w_l + a_l w_h + a_h w_h are
// accumulated in fp32 on v_mfma_f32_32x32x16_f16 (gemm_acc16) -- 3 MFMAs of 32 cycles replace 8 fp32 MFMAs of 64 cycles
// per K = 16.  Weights are pre-split at vssr_create (pack_mfma_tiles16), activations are split in registers when a
// fragment is read from its fp32 LDS tile (clamped to the fp16 range first: measured activations / adjoints of the
// SrTiO3 models peak at ~160, tools/gpu_ranges.py).  -DVSSR_NODE_FP32 selects the fp32 MFMA path (gemm_acc).
//
// Structure of every kernel: one workgroup = 4 waves = a tile of 32 atoms of one ensemble member.
//   * activations (A operand) live in LDS, row-major with a +4 float pad (conflict-free ds_read_b128);
//   * weights (B operand) were re-packed at vssr_create into MFMA fragment order
//       packed[tile][q][lane][4] = W[tile*32 + (lane&31)][(lane>>5)*(K/2) + 4q .. 4q+3]
//     so each wave streams its own column tiles with one fully coalesced 1 KiB dwordx4 load per 4 k-steps,
//     L2-resident (all workgroups read the same 0.7 MB per layer);
//   * the k index is permuted identically on A and B (lane half h covers k in [h*K/2, (h+1)*K/2)), which
//     only reorders the fp32 summation;
//   * wave w owns output features [32w, 32w+32) of every section, so gates, norms and residuals that
//     combine several GEMM outputs for the same (atom, feature) stay in one lane's registers.
#include "vssr_internal.h"

namespace vssr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TA = 32;           // atoms per workgroup
constexpr int LDV = F + 4;       // LDS row stride for K = 128 operands
constexpr int LDH = 2 * F + 4;   // K = 256
constexpr int LDQ = 3 * F + 4;   // K = 384

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float swish(float x) { return x * sigm(x); }
__device__ __forceinline__ float dswish(float x) {
    float sg = sigm(x);
    return sg * fmaf(x, 1.f - sg, 1.f);
}
// row of accumulator register `reg` inside a 32x32 tile (MI355X guide: C/D layout of 32x32 MFMA)
__device__ __forceinline__ int crow(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// Weight pointers are read out of the ModelW table in memory, so the compiler would treat them as FLAT (flat loads
// tick both vmcnt and lgkmcnt).  Loading through an explicit global address space pointer gives global_load.
__device__ __forceinline__ float4 gload4(const float *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float gf32x4 __attribute__((ext_vector_type(4)));
    typedef const gf32x4 __attribute__((address_space(1))) *gptr;
    const gf32x4 v = *reinterpret_cast<gptr>(reinterpret_cast<uintptr_t>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4 *>(p);
#endif
}

// acc[t][c] += A(rows t*32.., K) * packed tile c
template <int K, int NRT, int NCT>
__device__ __forceinline__ void gemm_acc(const float *__restrict__ lds_a, int ld,
                                         const float *const (&wp)[NCT], f32x16 (&acc)[NRT][NCT]) {
    const int lane = threadIdx.x & 63, half = lane >> 5, r = lane & 31;
    const float *a_base = lds_a + r * ld + half * (K / 2);
#pragma unroll 2
    for (int q = 0; q < K / 8; ++q) {
        float bv[NCT][4], av[NRT][4];
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            float4 b = gload4(wp[c] + ((size_t)q * 64 + lane) * 4);
            bv[c][0] = b.x; bv[c][1] = b.y; bv[c][2] = b.z; bv[c][3] = b.w;
        }
#pragma unroll
        for (int t = 0; t < NRT; ++t) {
            float4 a = *reinterpret_cast<const float4 *>(a_base + t * 32 * ld + 4 * q);
            av[t][0] = a.x; av[t][1] = a.y; av[t][2] = a.z; av[t][3] = a.w;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < NRT; ++t)
#pragma unroll
                for (int c = 0; c < NCT; ++c)
                    acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][s], bv[c][s], acc[t][c], 0, 0, 0);
    }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32x4 gload4u(const uint4 *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const u32x4 __attribute__((address_space(1))) *gptr;
    return *reinterpret_cast<gptr>(reinterpret_cast<uintptr_t>(p));
#else
    return *reinterpret_cast<const u32x4 *>(p);
#endif
}

// 2-way fp16 split of 8 consecutive fp32 values into two MFMA operands (h and l pieces, 8 x fp16 each)
__device__ __forceinline__ void split8(const float4 lo, const float4 hi, u32x4 (&o)[2]) {
    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 xc = {__builtin_amdgcn_fmed3f(x[2 * j], -65504.f, 65504.f),
                          __builtin_amdgcn_fmed3f(x[2 * j + 1], -65504.f, 65504.f)};
        const f16x2 h = __builtin_convertvector(xc, f16x2);
        const f16x2 l = __builtin_convertvector(xc - __builtin_convertvector(h, f32x2), f16x2);
        o[0][j] = __builtin_bit_cast(unsigned, h);
        o[1][j] = __builtin_bit_cast(unsigned, l);
    }
}

// acc[t][c] += A(rows t*32.., K) * W tile c, fp16-split.  wq[c]: the tile's pieces, [K/16][2][64 lanes] uint4
// (pack_mfma_tiles16).  Lane (r = lane & 31, half = lane >> 5) supplies row r / column r and k = 16 q + 8 half .. + 7.
// Issue order (see painn_edge_mfma.hip, MFMA hazard rules): the three partial products of a tile form a dependent
// accumulator chain, so products are issued in rounds over all independent accumulators; a GEMM with fewer than three
// tiles gets NACC K-interleaved partial accumulators (summed at the end) so that a chain's producer is always >= 3
// MFMAs back.  No load is issued inside the MFMA block of a chunk group.
template <int K, int NRT, int NCT>
__device__ __forceinline__ void gemm_acc16(const float *__restrict__ lds_a, int ld, const uint4 *const (&wq)[NCT],
                                           f32x16 (&acc)[NRT][NCT]) {
    constexpr int NT = NRT * NCT, NACC = NT >= 3 ? 1 : (NT == 2 ? 2 : 4);
    constexpr int wi[3] = {0, 1, 0}, ri[3] = {1, 0, 0};   // A-piece, W-piece: a_h w_l, a_l w_h, a_h w_h
    const int lane = threadIdx.x & 63, half = lane >> 5, r = lane & 31;
    const float *a_base = lds_a + r * ld + 8 * half;
    f32x16 part[NACC][NRT][NCT];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int t = 0; t < NRT; ++t)
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                if (j == 0) part[0][t][c] = acc[t][c];
                else
#pragma unroll
                    for (int i = 0; i < 16; ++i) part[j][t][c][i] = 0.f;
            }
#pragma unroll 1
    for (int q0 = 0; q0 < K / 16; q0 += NACC) {
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
                const float *ap = a_base + t * 32 * ld + 16 * q;
                split8(*reinterpret_cast<const float4 *>(ap), *reinterpret_cast<const float4 *>(ap + 4), a[j][t]);
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
                        part[j][t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(f16x8, a[j][t][wi[k]]), __builtin_bit_cast(f16x8, b[j][c][ri[k]]),
                            part[j][t][c], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
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

// weight-tile pointers of the two paths: tile index and K select the 32-column tile of a packed matrix
#ifdef VSSR_NODE_FP32
#define WTILE(name, tile, K) (W.p##name + (size_t)(tile) * 32 * (K))
#define WPTR const float *
#define GEMM gemm_acc
#else
#define WTILE(name, tile, K) (W.q##name + (size_t)(tile) * 8 * (K))
#define WPTR const uint4 *
#define GEMM gemm_acc16
#endif

template <int NRT, int NCT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NRT][NCT]) {
#pragma unroll
    for (int t = 0; t < NRT; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][c][i] = 0.f;
}

// Cooperative tile load: NROWS rows of F floats.  rowptr(row) must always return a readable row
// (tail rows are clamped to the last atom; their results are never stored), so that all loads are
// unconditional and issued back-to-back before the first LDS store (no per-element branch / vmcnt(0)).
template <int NROWS, class RowPtr>
__device__ __forceinline__ void load_rows(float *lds, int ld, int col0, RowPtr rowptr) {
    constexpr int NIT = NROWS * (F / 4) / 256;
    float4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int idx = threadIdx.x + it * 256;
        v[it] = *reinterpret_cast<const float4 *>(rowptr(idx >> 5) + 4 * (idx & 31));
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int idx = threadIdx.x + it * 256;
        *reinterpret_cast<float4 *>(lds + (idx >> 5) * ld + col0 + 4 * (idx & 31)) = v[it];
    }
}

// ---- message MLP forward: phi = W2 swish(W1 s + b1) + b2 ---------------------------------------------------
__global__ void __launch_bounds__(256, 1)
k_msg_mlp_mfma(int N, int l, const ModelW *__restrict__ MW, const float *__restrict__ s_in,
               float *__restrict__ phi) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *xs = lds;                 // [TA][LDV]
    float *hs = lds + TA * LDV;      // [TA][LDV]
    const int m = blockIdx.y, a0 = blockIdx.x * TA, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, col = 32 * w + (lane & 31);
    const LayerW &W = MW[m].layer[l];
    const size_t mN = (size_t)m * N;
    load_rows<TA>(xs, LDV, 0, [&](int row) { return s_in + (mN + min(a0 + row, N - 1)) * F; });
    __syncthreads();
    {
        f32x16 acc[1][1];
        zero_acc(acc);
        WPTR wp[1] = {WTILE(W1, w, F)};
        GEMM<F, 1, 1>(xs, LDV, wp, acc);
        float b = W.b1[col];
#pragma unroll
        for (int i = 0; i < 16; ++i) hs[crow(i, half) * LDV + col] = swish(acc[0][0][i] + b);
    }
    __syncthreads();
    f32x16 acc[1][3];
    zero_acc(acc);
    WPTR wp[3] = {WTILE(W2, w, F), WTILE(W2, 4 + w, F), WTILE(W2, 8 + w, F)};
    GEMM<F, 1, 3>(hs, LDV, wp, acc);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float b = W.b2[c * F + col];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int a = a0 + crow(i, half);
            if (a < N) phi[(mN + a) * F3 + c * F + col] = acc[0][c][i] + b;
        }
    }
}

// ---- message MLP reverse: sbar_in = sbar_msg + W1^T[(W2^T phibar) * swish'(W1 s + b1)] ------------------------------
__global__ void __launch_bounds__(256, 1)
k_msg_mlp_bwd_mfma(int N, int l, const ModelW *__restrict__ MW, const float *__restrict__ s_in,
                   const float *__restrict__ phibar, const float *__restrict__ sbar_msg, float *__restrict__ sbar_in) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *xs = lds;                 // [TA][LDV]  s tile, later h1bar
    float *pb = lds + TA * LDV;      // [TA][LDQ]  phibar tile
    const int m = blockIdx.y, a0 = blockIdx.x * TA, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, col = 32 * w + (lane & 31);
    const LayerW &W = MW[m].layer[l];
    const size_t mN = (size_t)m * N;
    load_rows<TA>(xs, LDV, 0, [&](int row) { return s_in + (mN + min(a0 + row, N - 1)) * F; });
#pragma unroll
    for (int c = 0; c < 3; ++c)
        load_rows<TA>(pb, LDQ, c * F, [&](int row) { return phibar + (mN + min(a0 + row, N - 1)) * F3 + c * F; });
    __syncthreads();
    f32x16 h1[1][1], a1[1][1];
    zero_acc(h1);
    zero_acc(a1);
    {
        WPTR wp[1] = {WTILE(W1, w, F)};
        GEMM<F, 1, 1>(xs, LDV, wp, h1);
        WPTR wq[1] = {WTILE(W2t, w, F3)};
        GEMM<F3, 1, 1>(pb, LDQ, wq, a1);
    }
    __syncthreads();  // everyone is done reading xs
    {
        float b = W.b1[col];
#pragma unroll
        for (int i = 0; i < 16; ++i) xs[crow(i, half) * LDV + col] = a1[0][0][i] * dswish(h1[0][0][i] + b);
    }
    __syncthreads();
    f32x16 acc[1][1];
    zero_acc(acc);
    WPTR wp[1] = {WTILE(W1t, w, F)};
    GEMM<F, 1, 1>(xs, LDV, wp, acc);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int a = a0 + crow(i, half);
        if (a < N) sbar_in[(mN + a) * F + col] = sbar_msg[(mN + a) * F + col] + acc[0][0][i];
    }
}

// ---- update block -----------------------------------------------------------------------------------------------
// LDS map (floats): vt [3*TA][LDV] | hs [TA][LDH] | as [TA][LDV]   (contiguous; the reverse pass overlays it)
constexpr int OFF_VT = 0;
constexpr int OFF_HS = 3 * TA * LDV;
constexpr int OFF_AS = OFF_HS + TA * LDH;
constexpr int UPD_LDS_FLOATS = OFF_AS + TA * LDV;   // 25 216 floats = 100 864 B

struct UpdRegs {
    f32x16 uv[3][2];   // [x][0] = U v, [x][1] = V v   for feature `col`, 16 atoms (rows) per lane
    f32x16 h3;         // pre-activation of the gate MLP
    f32x16 gate[3];    // a_vv, a_sv, a_ss
    float nrm[16], inner[16];
};

// Shared forward part: needs vt (v_msg tile, rows x*TA+atom) and hs[:, :F] (s_msg tile) loaded + synced.
__device__ __forceinline__ void update_forward(const LayerW &W, float *lds, int w, int half, int col, UpdRegs &R) {
    float *vt = lds + OFF_VT, *hs = lds + OFF_HS, *as_ = lds + OFF_AS;
    zero_acc(R.uv);
    {
        WPTR wp[2] = {WTILE(U, w, F), WTILE(V, w, F)};
        GEMM<F, 3, 2>(vt, LDV, wp, R.uv);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float n2 = 0.f, in = 0.f;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            float vv = R.uv[x][1][i];
            n2 += fmaf(vv, vv, 1e-15f);
            in = fmaf(R.uv[x][0][i], vv, in);
        }
        R.nrm[i] = sqrtf(n2);
        R.inner[i] = in;
        hs[crow(i, half) * LDH + F + col] = R.nrm[i];
    }
    __syncthreads();
    {
        f32x16 acc[1][1];
        zero_acc(acc);
        WPTR wp[1] = {WTILE(W3, w, 2 * F)};
        GEMM<2 * F, 1, 1>(hs, LDH, wp, acc);
        float b = W.b3[col];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            R.h3[i] = acc[0][0][i] + b;
            as_[crow(i, half) * LDV + col] = swish(R.h3[i]);
        }
    }
    __syncthreads();
    {
        f32x16 acc[1][3];
        zero_acc(acc);
        WPTR wp[3] = {WTILE(W4, w, F), WTILE(W4, 4 + w, F), WTILE(W4, 8 + w, F)};
        GEMM<F, 1, 3>(as_, LDV, wp, acc);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float b = W.b4[c * F + col];
#pragma unroll
            for (int i = 0; i < 16; ++i) R.gate[c][i] = acc[0][c][i] + b;
        }
    }
}

__device__ __forceinline__ void load_update_tiles(float *lds, const float *__restrict__ s_msg,
                                                  const float *__restrict__ v_msg, size_t mN, int a0, int N) {
    load_rows<3 * TA>(lds + OFF_VT, LDV, 0, [&](int row) {
        int x = row / TA, a = min(a0 + (row % TA), N - 1);
        return v_msg + ((mN + a) * 3 + x) * F;
    });
    load_rows<TA>(lds + OFF_HS, LDH, 0, [&](int row) { return s_msg + (mN + min(a0 + row, N - 1)) * F; });
}

__global__ void __launch_bounds__(256, 1)
k_update_fwd_mfma(int N, int l, const ModelW *__restrict__ MW, const float *__restrict__ s_msg,
                  const float *__restrict__ v_msg, float *__restrict__ s_out, float *__restrict__ v_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int m = blockIdx.y, a0 = blockIdx.x * TA, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, col = 32 * w + (lane & 31);
    const LayerW &W = MW[m].layer[l];
    const size_t mN = (size_t)m * N;
    load_update_tiles(lds, s_msg, v_msg, mN, a0, N);
    __syncthreads();
    UpdRegs R;
    update_forward(W, lds, w, half, col, R);
    const float *vt = lds + OFF_VT, *hs = lds + OFF_HS;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int row = crow(i, half), a = a0 + row;
        if (a >= N) continue;
        size_t g = mN + a;
        s_out[g * F + col] = fmaf(R.gate[1][i], R.inner[i], hs[row * LDH + col]) + R.gate[2][i];
#pragma unroll
        for (int x = 0; x < 3; ++x)
            v_out[(g * 3 + x) * F + col] = fmaf(R.gate[0][i], R.uv[x][0][i], vt[(x * TA + row) * LDV + col]);
    }
}

// reverse: (sbar, vbar) of the block outputs -> (sbar_msg, vbar_msg) of its inputs.
// Adjoint algebra as in the oracle (oracle/painn_impl.inc, "update block^T"):
//   abar_vv = sum_x vbar_x Uv_x ; qbar = [abar_vv, sbar*inner, sbar]
//   h3bar = (W4^T qbar) * swish'(h3) ; [sbar_extra ; nbar] = W3^T h3bar
//   Ubar_x = vbar_x a_vv + sbar a_sv Vv_x ; Vbar_x = sbar a_sv Uv_x + nbar Vv_x / |Vv|
//   vbar_msg = vbar + U^T Ubar + V^T Vbar ; sbar_msg = sbar + sbar_extra
__global__ void __launch_bounds__(256, 1)
k_update_bwd_mfma(int N, int l, int vbar_is_zero, const ModelW *__restrict__ MW, const float *__restrict__ s_msg,
                  const float *__restrict__ v_msg, const float *__restrict__ sbar, const float *__restrict__ vbar,
                  float *__restrict__ sbar_msg, float *__restrict__ vbar_msg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int m = blockIdx.y, a0 = blockIdx.x * TA, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, col = 32 * w + (lane & 31);
    const LayerW &W = MW[m].layer[l];
    const size_t mN = (size_t)m * N;
    load_update_tiles(lds, s_msg, v_msg, mN, a0, N);
    __syncthreads();
    UpdRegs R;
    update_forward(W, lds, w, half, col, R);
    // Every wave has passed the barrier in front of GEMM3, i.e. finished GEMM1/GEMM2: vt and hs are free.
    float *qb = lds + OFF_VT;    // [TA][LDQ] overlays vt (12 416 <= 12 672 floats)
    float sb[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int row = crow(i, half);
        size_t g = mN + min(a0 + row, N - 1);
        sb[i] = sbar[g * F + col];
        float abar_vv = 0.f;
        if (!vbar_is_zero) {   // wave-uniform
#pragma unroll
            for (int x = 0; x < 3; ++x) abar_vv = fmaf(vbar[(g * 3 + x) * F + col], R.uv[x][0][i], abar_vv);
        }
        qb[row * LDQ + col] = abar_vv;
        qb[row * LDQ + F + col] = sb[i] * R.inner[i];
        qb[row * LDQ + 2 * F + col] = sb[i];
    }
    __syncthreads();   // qb complete; every wave is past GEMM3, so `as` may be overwritten
    float *hb = lds + OFF_AS;    // [TA][LDV] h3bar
    {
        f32x16 acc[1][1];
        zero_acc(acc);
        WPTR wp[1] = {WTILE(W4t, w, F3)};
        GEMM<F3, 1, 1>(qb, LDQ, wp, acc);
#pragma unroll
        for (int i = 0; i < 16; ++i) hb[crow(i, half) * LDV + col] = acc[0][0][i] * dswish(R.h3[i]);
    }
    __syncthreads();
    f32x16 hbar[1][2];   // [0] = d/d s_msg part, [1] = d/d norm part, both for feature `col`
    zero_acc(hbar);
    {
        WPTR wp[2] = {WTILE(W3t, w, F), WTILE(W3t, 4 + w, F)};
        GEMM<F, 1, 2>(hb, LDV, wp, hbar);
    }
    __syncthreads();   // all waves are done with qb / hb: the whole region becomes the [Ubar | Vbar] tile
    float *ab = lds;     // [3*TA][LDH]: cols [0,F) = Ubar, [F,2F) = Vbar   (24 960 <= 25 216 floats)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int row = crow(i, half), a = a0 + row;
        size_t g = mN + min(a, N - 1);
        if (a < N) sbar_msg[g * F + col] = sb[i] + hbar[0][0][i];
        float avv = R.gate[0][i], asv = R.gate[1][i];
        float sc = hbar[0][1][i] / R.nrm[i];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            float vbo = vbar_is_zero ? 0.f : vbar[(g * 3 + x) * F + col];
            float u = R.uv[x][0][i], v = R.uv[x][1][i];
            float sa = sb[i] * asv;
            ab[(x * TA + row) * LDH + col] = fmaf(vbo, avv, sa * v);
            ab[(x * TA + row) * LDH + F + col] = fmaf(sa, u, sc * v);
        }
    }
    __syncthreads();
    f32x16 out[3][1];
    zero_acc(out);
    {
        WPTR wp[1] = {WTILE(UVt, w, 2 * F)};
        GEMM<2 * F, 3, 1>(ab, LDH, wp, out);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int row = crow(i, half), a = a0 + row;
        if (a >= N) continue;
        size_t g = mN + a;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            float vbo = vbar_is_zero ? 0.f : vbar[(g * 3 + x) * F + col];
            vbar_msg[(g * 3 + x) * F + col] = vbo + out[x][0][i];
        }
    }
}

// ---- host: weight packing + launch helpers ------------------------------------------------------------------------
// packed[tile][q][lane][t] = W[tile*32 + (lane&31)][(lane>>5)*(K/2) + 4q + t]   (W row-major [rows][K])
void pack_mfma_tiles(const float *Wsrc, int rows, int K, float *dst) {
    const int ntile = rows / 32, nq = K / 8;
    for (int tile = 0; tile < ntile; ++tile)
        for (int q = 0; q < nq; ++q)
            for (int lane = 0; lane < 64; ++lane)
                for (int t = 0; t < 4; ++t)
                    dst[(((size_t)tile * nq + q) * 64 + lane) * 4 + t] =
                        Wsrc[(size_t)(tile * 32 + (lane & 31)) * K + (lane >> 5) * (K / 2) + 4 * q + t];
}

// fp16-split fragment order for v_mfma_f32_32x32x16_f16:
//   dst[tile][q][piece][lane][j] = piece(W[tile*32 + (lane&31)][16 q + 8 (lane>>5) + 2 j]) | piece(W[..][.. + 1]) << 16
void pack_mfma_tiles16(const float *Wsrc, int rows, int K, unsigned *dst) {
    auto split2 = [](float x, unsigned (&p)[2]) {
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        unsigned short hb, lb;
        memcpy(&hb, &h, 2);
        memcpy(&lb, &l, 2);
        p[0] = hb; p[1] = lb;
    };
    const int ntile = rows / 32, nq = K / 16;
    for (int tile = 0; tile < ntile; ++tile)
        for (int q = 0; q < nq; ++q)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const size_t row = (size_t)tile * 32 + (lane & 31);
                    const int k = 16 * q + 8 * (lane >> 5) + 2 * j;
                    unsigned p0[2], p1[2];
                    split2(Wsrc[row * K + k], p0);
                    split2(Wsrc[row * K + k + 1], p1);
                    for (int pc = 0; pc < 2; ++pc)
                        dst[((((size_t)tile * nq + q) * 2 + pc) * 64 + lane) * 4 + j] = p0[pc] | (p1[pc] << 16);
                }
}

size_t node_mfma_lds_bytes(int which) {
    switch (which) {
        case 0: return sizeof(float) * 2 * TA * LDV;            // msg mlp fwd
        case 1: return sizeof(float) * (TA * LDV + TA * LDQ);   // msg mlp bwd
        default: return sizeof(float) * UPD_LDS_FLOATS;         // update fwd / bwd
    }
}

int node_mfma_init(vssr_handle *h) {
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_msg_mlp_mfma, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)node_mfma_lds_bytes(0)));
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_msg_mlp_bwd_mfma, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)node_mfma_lds_bytes(1)));
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_update_fwd_mfma, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)node_mfma_lds_bytes(2)));
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_update_bwd_mfma, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)node_mfma_lds_bytes(2)));
    return VSSR_OK;
}

void launch_msg_mlp_mfma(hipStream_t st, int N, int M, int l, const ModelW *MW, const float *s_in, float *phi) {
    hipLaunchKernelGGL(k_msg_mlp_mfma, dim3((N + TA - 1) / TA, M), dim3(256), node_mfma_lds_bytes(0), st, N, l, MW,
                       s_in, phi);
}
void launch_msg_mlp_bwd_mfma(hipStream_t st, int N, int M, int l, const ModelW *MW, const float *s_in,
                             const float *phibar, const float *sbar_msg, float *sbar_in) {
    hipLaunchKernelGGL(k_msg_mlp_bwd_mfma, dim3((N + TA - 1) / TA, M), dim3(256), node_mfma_lds_bytes(1), st, N, l, MW,
                       s_in, phibar, sbar_msg, sbar_in);
}
void launch_update_fwd_mfma(hipStream_t st, int N, int M, int l, const ModelW *MW, const float *s_msg,
                            const float *v_msg, float *s_out, float *v_out) {
    hipLaunchKernelGGL(k_update_fwd_mfma, dim3((N + TA - 1) / TA, M), dim3(256), node_mfma_lds_bytes(2), st, N, l, MW,
                       s_msg, v_msg, s_out, v_out);
}
void launch_update_bwd_mfma(hipStream_t st, int N, int M, int l, int vbar_is_zero, const ModelW *MW,
                            const float *s_msg, const float *v_msg, const float *sbar, const float *vbar,
                            float *sbar_msg, float *vbar_msg) {
    hipLaunchKernelGGL(k_update_bwd_mfma, dim3((N + TA - 1) / TA, M), dim3(256), node_mfma_lds_bytes(2), st, N, l,
                       vbar_is_zero, MW, s_msg, v_msg, sbar, vbar, sbar_msg, vbar_msg);
}

}  // namespace vssr
