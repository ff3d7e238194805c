// painn_edge_mfma.hip — the neighbor-sum ("message") stage of PaiNN on gfx950: LDS-staged
// feature slices + the radial filter on the fp32 matrix cores.
//
// Math (nff MessageBlock / DistanceEmbed):
//   w_e = (Wd rbf(d_e) + bd) fcut(d_e) = Wd_ext . rho(d_e),  rho = [sin(n pi d/rc)/d * fc]_{n=1..20} ++ [fc]
//   s_i += sum_e phi_j[b] w_e[b] ;  v_i += sum_e ( phi_j[c] w_e[c] u_e + phi_j[a] w_e[a] v_j )
//
// Decomposition: one workgroup = (chain, 16-feature slice, ensemble member).  It stages the slice of
// phi and v of ALL atoms of its chain in LDS once (exactly the compulsory HBM bytes of SURVEY §8(d):
// every phi/v element is read by one workgroup only), then walks the chain's padded CSR.  The filter
// GEMM  [slots x 24] . [24 x 16]  runs on v_mfma_f32_16x16x4_f32 with slots as rows, so a lane holds the
// filter values of ITS lane-group's 4 slots for ITS feature: each 16-lane group streams its own centre
// atom (4 slots per step), accumulates ds / dv in registers and writes the centre once.  No atomics, no
// cross-lane reduction; summation order per centre is the CSR order regardless of batching.
//   A operand (rho) comes from the per-slot table written once per evaluation by k_edge_geom (nbr.hip) in
//   A-fragment order (PMC showed the kernel is instruction-issue bound: recomputing sincos per slice x
//   model x layer cost ~110 of ~360 instructions per step); unit vectors / local neighbor ids likewise.
//   B operand (Wd_ext slice) lives in 18 VGPRs for the whole kernel.
// Chains larger than the LDS capacity (N > ~400) fall back to the gather kernels in painn.hip.
#include "vssr_internal.h"

namespace vssr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FS = 16;           // features per slice
constexpr int NSLICE = F / FS;   // 8
constexpr int EDGE_THREADS = 1024;  // 16 waves: 4 per SIMD at one workgroup per CU

// LDS slice layout: tile[atom][feature f][NSEG] with NSEG = {a, b, c, v_x, v_y, v_z} (layer 0: {b, c}): the values
// a lane needs for its 4 features of one neighbor are 96 contiguous bytes = 6 ds_read_b128 (layer 0: 2).
template <bool L0> struct EdgeLayout {
    static constexpr int NSEC = L0 ? 2 : 3;          // filter sections used (layer 0: b, c; v = 0)
    static constexpr int NSEG = L0 ? 2 : 6;          // values staged per (atom, feature)
    static constexpr int ROW = NSEG * FS + 4;        // LDS row stride (floats); rows stay 16-B aligned
};

// LDS carve-up: tile [max_atoms][ROW] | s slice [max_atoms][FS] | row_start [max_atoms + 1] (ints)
size_t edge_fwd_lds_bytes(int max_atoms, bool l0) {
    const int row = l0 ? EdgeLayout<true>::ROW : EdgeLayout<false>::ROW;
    return sizeof(float) * ((size_t)max_atoms * (row + FS) + max_atoms + 4);
}

// sum over the 4 lanes of a quad (lanes 4q..4q+3), result in every lane: two DPP quad_perm adds
__device__ __forceinline__ float quad_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    return x;
}

// Lane roles ("slot-major"): p = lane & 15 is the slot position inside the step's 16-slot tile, stream = p >> 2
// (4 independent CSR streams per wave, 4 slots each per step), e = p & 3 the slot inside the quad, and
// fq = lane >> 4 selects features 4 fq .. 4 fq + 3 of the slice.  With the WEIGHTS as MFMA A operand
// (A[i = feature][k]) and rho as B (B[k][j = slot]) the filter tile D[feature][slot] puts the 4 feature values of
// slot p into the 4 accumulator registers of lane (p, fq): the lane that loaded rho for slot p (its k-quarter is
// fq) also owns that slot's messages, so one table address serves both and no cross-lane traffic is needed.
template <bool L0>
__global__ void __launch_bounds__(EDGE_THREADS)
k_edge_fwd_mfma(int N, int l, const ModelW *__restrict__ MW, GraphView G, const int *__restrict__ counters,
                int zero_slot, int n_models, int max_atoms, const float *__restrict__ s_in,
                const float *__restrict__ v_in, const float *__restrict__ phi, float *__restrict__ s_msg,
                float *__restrict__ v_msg) {
    using LY = EdgeLayout<L0>;
    extern __shared__ __attribute__((aligned(16))) float tile[];
    if (counters[2]) return;
    // XCD-aware 1-D grid: workgroup id -> XCD id % 8 (observed dispatch rule).  All (slice, model) workgroups of
    // one chain get consecutive ids on ONE XCD, so the chain's rho / record tables (~1.3 MB) and its phi / v
    // rows are fetched from HBM once and then served by that XCD's L2.
    const int wg = blockIdx.x, xcd = wg & 7, rest = wg >> 3;
    const int per_chain = NSLICE * n_models;
    const int t = rest % per_chain, b = (rest / per_chain) * 8 + xcd;
    if (b >= G.n_cfg) return;
    const int fs = t % NSLICE, m = t / NSLICE;
    const int a0 = G.cfg_start[b], Nc = G.cfg_start[b + 1] - a0;
    const size_t mN = (size_t)m * N;
    const int tid = threadIdx.x;

    // ---- stage the chain's feature slice: tile[atom][f][seg], s slice, row_start ------------------------------
    // (NSEG + 1) * 4 float4 per atom; loads are issued in batches of 4 per thread before any LDS store so that
    // the L2 / HBM round trips overlap (one workgroup per CU: nothing else hides them)
    float *s_tile = tile + (size_t)max_atoms * LY::ROW;                      // [atom][FS]
    int *rs = reinterpret_cast<int *>(s_tile + (size_t)max_atoms * FS);      // [Nc + 1] row_start of this chain
    {
        constexpr int PER_ATOM = (LY::NSEG + 1) * 4;   // float4 per atom: NSEG slice segments + the s slice
        const int total = Nc * PER_ATOM;
        auto src_of = [&](int idx) -> const float * {
            int atom = idx / PER_ATOM, rem = idx - atom * PER_ATOM, seg = rem >> 2, q4 = rem & 3;
            const size_t ga = mN + a0 + atom;
            if (seg == LY::NSEG) return s_in + ga * F + fs * FS + q4 * 4;                  // s slice
            if (L0) return phi + ga * F3 + (seg + 1) * F + fs * FS + q4 * 4;               // sections b, c
            if (seg < 3) return phi + ga * F3 + seg * F + fs * FS + q4 * 4;                // sections a, b, c
            return v_in + (ga * 3 + (seg - 3)) * F + fs * FS + q4 * 4;                     // v_x, v_y, v_z
        };
        auto put = [&](int idx, const float4 &val) {
            int atom = idx / PER_ATOM, rem = idx - atom * PER_ATOM, seg = rem >> 2, q4 = rem & 3;
            if (seg == LY::NSEG) {
                *reinterpret_cast<float4 *>(s_tile + atom * FS + q4 * 4) = val;
            } else {
                float *dst = tile + atom * LY::ROW + (q4 * 4) * LY::NSEG + seg;
                dst[0] = val.x; dst[LY::NSEG] = val.y; dst[2 * LY::NSEG] = val.z; dst[3 * LY::NSEG] = val.w;
            }
        };
        for (int base = tid; base < total; base += 4 * EDGE_THREADS) {
            float4 v4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = min(base + u * EDGE_THREADS, total - 1);
                v4[u] = *reinterpret_cast<const float4 *>(src_of(idx));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (base + u * EDGE_THREADS < total) put(base + u * EDGE_THREADS, v4[u]);
        }
        for (int idx = tid; idx <= Nc; idx += EDGE_THREADS) rs[idx] = G.row_start[a0 + idx];
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, p = lane & 15, fq = lane >> 4, e = p & 3;
    const int sid = wave * 4 + (p >> 2);                      // stream id inside the workgroup
    const int nstreams = (EDGE_THREADS / 64) * 4;
    // ---- A operand: Wd_ext[section row = feature (lane & 15)][k = 4 ks + (lane >> 4)] ---------------------------
    const LayerW &W = MW[m].layer[l];
    float wA[LY::NSEC][6];
#pragma unroll
    for (int s = 0; s < LY::NSEC; ++s) {
        const int row = (L0 ? s + 1 : s) * F + fs * FS + p;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            int k = 4 * ks + fq;
            wA[s][ks] = k < 20 ? W.Wd[(size_t)row * 20 + k] : (k == 20 ? W.bd[row] : 0.f);
        }
    }

    // ---- every stream walks a contiguous run of CSR slots, cut at centre boundaries, ~equal slot counts ---------
    const int slot0 = rs[0], slots = rs[Nc] - slot0;
    auto first_centre = [&](int sidx) {   // first centre whose slots start at or after the sidx-th slot quantile
        const int target = slot0 + (int)(((long long)slots * sidx) / nstreams);
        int lo = 0, hi = Nc;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (rs[mid] < target) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const int c_first = sid == 0 ? 0 : first_centre(sid);
    const int c_last = sid == nstreams - 1 ? Nc : first_centre(sid + 1);   // centres [c_first, c_last)
    int c = c_first;
    int pos = rs[c_first];
    const int stream_end = rs[c_last];
    int cend = c < c_last ? rs[c + 1] : stream_end;               // end of the current centre
    float ds[4] = {0.f, 0.f, 0.f, 0.f}, dvx[4] = {0.f, 0.f, 0.f, 0.f}, dvy[4] = {0.f, 0.f, 0.f, 0.f},
          dvz[4] = {0.f, 0.f, 0.f, 0.f};
    const int fcol = fs * FS + 4 * fq;            // first of this lane's 4 global feature columns

    // table entry of this lane's slot; exhausted streams read the reserved all-zero entry (filter = 0)
    const float *rho_lane = G.rho + fq * 6;
    const float4 *erec = G.erec;
    const int last_slot = max(rs[Nc] - 1, 0);     // records are always read from inside the chain (finite values)

    float2 rh[2][3];
    float4 er[2];
    {
        const int sl = pos + e;
        const float2 *rp = reinterpret_cast<const float2 *>(rho_lane + (size_t)(pos < stream_end ? sl : zero_slot) * 24);
        rh[0][0] = rp[0]; rh[0][1] = rp[1]; rh[0][2] = rp[2];
        er[0] = erec[min(sl, last_slot)];
    }

    // A centre that is complete is written exactly once (also covers centres without any neighbor): reduce the 4
    // slot lanes of the quad, add the residual from the staged slices (no global loads), one float4 store per row.
    auto flush_complete = [&]() {
        while (c < c_last && pos >= cend) {
            float4 so, vxo, vyo, vzo;
            so.x = quad_sum(ds[0]); so.y = quad_sum(ds[1]); so.z = quad_sum(ds[2]); so.w = quad_sum(ds[3]);
            vxo.x = quad_sum(dvx[0]); vxo.y = quad_sum(dvx[1]); vxo.z = quad_sum(dvx[2]); vxo.w = quad_sum(dvx[3]);
            vyo.x = quad_sum(dvy[0]); vyo.y = quad_sum(dvy[1]); vyo.z = quad_sum(dvy[2]); vyo.w = quad_sum(dvy[3]);
            vzo.x = quad_sum(dvz[0]); vzo.y = quad_sum(dvz[1]); vzo.z = quad_sum(dvz[2]); vzo.w = quad_sum(dvz[3]);
            if (e == 0) {
                const size_t ga = mN + a0 + c;
                const float4 sr = *reinterpret_cast<const float4 *>(s_tile + c * FS + 4 * fq);
                so.x += sr.x; so.y += sr.y; so.z += sr.z; so.w += sr.w;
                if (!L0) {
                    const float *vc = tile + c * LY::ROW + (4 * fq) * LY::NSEG;
                    vxo.x += vc[3]; vxo.y += vc[9]; vxo.z += vc[15]; vxo.w += vc[21];
                    vyo.x += vc[4]; vyo.y += vc[10]; vyo.z += vc[16]; vyo.w += vc[22];
                    vzo.x += vc[5]; vzo.y += vc[11]; vzo.z += vc[17]; vzo.w += vc[23];
                }
                *reinterpret_cast<float4 *>(s_msg + ga * F + fcol) = so;
                *reinterpret_cast<float4 *>(v_msg + (ga * 3 + 0) * F + fcol) = vxo;
                *reinterpret_cast<float4 *>(v_msg + (ga * 3 + 1) * F + fcol) = vyo;
                *reinterpret_cast<float4 *>(v_msg + (ga * 3 + 2) * F + fcol) = vzo;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { ds[r] = 0.f; dvx[r] = 0.f; dvy[r] = 0.f; dvz[r] = 0.f; }
            ++c;
            cend = c < c_last ? rs[c + 1] : stream_end;
        }
    };
    flush_complete();   // leading centres without neighbors

    const float *trow = tile + (4 * fq) * LY::NSEG;
    while (__any(pos < stream_end)) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {   // two steps per iteration: ping-pong the prefetch registers
            // prefetch the next step's table entries (in flight during this step's math)
            {
                const int sl = pos + 4 + e;
                const float2 *rp =
                    reinterpret_cast<const float2 *>(rho_lane + (size_t)(pos + 4 < stream_end ? sl : zero_slot) * 24);
                rh[ph ^ 1][0] = rp[0]; rh[ph ^ 1][1] = rp[1]; rh[ph ^ 1][2] = rp[2];
                er[ph ^ 1] = erec[min(sl, last_slot)];
            }
            // gather this slot's neighbor row: 4 features x NSEG values, contiguous in LDS
            float tv[4 * LY::NSEG];
            {
                const float4 *row = reinterpret_cast<const float4 *>(trow + __float_as_int(er[ph].w) * LY::ROW);
#pragma unroll
                for (int q = 0; q < LY::NSEG; ++q) {
                    const float4 t4 = row[q];
                    tv[4 * q] = t4.x; tv[4 * q + 1] = t4.y; tv[4 * q + 2] = t4.z; tv[4 * q + 3] = t4.w;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- filter GEMM  D[feature][slot] = Wd_ext[feature][k] rho[k][slot] ------------------------------------
            const float rho[6] = {rh[ph][0].x, rh[ph][0].y, rh[ph][1].x, rh[ph][1].y, rh[ph][2].x, rh[ph][2].y};
            f32x4 acc[LY::NSEC];
#pragma unroll
            for (int s2 = 0; s2 < LY::NSEC; ++s2) acc[s2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 6; ++ks)
#pragma unroll
                for (int s2 = 0; s2 < LY::NSEC; ++s2)
                    acc[s2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[s2][ks], rho[ks], acc[s2], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- messages of this lane's slot for its 4 features (filter = 0 exactly for pads / foreign slots) -----------
            const float ux = er[ph].x, uy = er[ph].y, uz = er[ph].z;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float *tr = tv + r * LY::NSEG;
                if (L0) {
                    const float wb = acc[0][r], wc = acc[1][r];
                    ds[r] = fmaf(tr[0], wb, ds[r]);
                    const float mc = tr[1] * wc;
                    dvx[r] = fmaf(mc, ux, dvx[r]); dvy[r] = fmaf(mc, uy, dvy[r]); dvz[r] = fmaf(mc, uz, dvz[r]);
                } else {
                    const float wa = acc[0][r], wb = acc[1][r], wc = acc[2][r];
                    ds[r] = fmaf(tr[1], wb, ds[r]);
                    const float mc = tr[2] * wc, ma = tr[0] * wa;
                    dvx[r] = fmaf(mc, ux, dvx[r]); dvy[r] = fmaf(mc, uy, dvy[r]); dvz[r] = fmaf(mc, uz, dvz[r]);
                    dvx[r] = fmaf(ma, tr[3], dvx[r]);
                    dvy[r] = fmaf(ma, tr[4], dvy[r]);
                    dvz[r] = fmaf(ma, tr[5], dvz[r]);
                }
            }
            if (pos < stream_end) pos += 4;
            flush_complete();
        }
    }
}

// layer-0 excluded volume: e_excl[i] = sum_e (sigma/d_e)^p  (geometry only, identical for all ensemble
// members that share sigma/p; written per member to keep the readout kernel's indexing)
__global__ void __launch_bounds__(256)
k_excl_vol(int N, int M, GraphView G, const int *__restrict__ counters, float sigma, int power,
           float *__restrict__ e_excl) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || counters[2]) return;
    float tot = 0.f;
    for (int e = G.row_start[i]; e < G.row_start[i + 1]; ++e) {
        float4 ed = G.edge[e];
        if (__float_as_int(ed.w) < 0) continue;
        float d = sqrtf(fmaf(ed.z, ed.z, fmaf(ed.y, ed.y, ed.x * ed.x)));
        tot += powf(sigma / d, (float)power);
    }
    for (int m = 0; m < M; ++m) e_excl[(size_t)m * N + i] = tot;
}

int edge_mfma_init(vssr_handle *h) {
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_edge_fwd_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_edge_fwd_mfma<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
    return VSSR_OK;
}

bool edge_fwd_mfma_fits(int max_atoms) { return edge_fwd_lds_bytes(max_atoms, false) <= 160 * 1024; }

void launch_edge_fwd_mfma(hipStream_t st, int N, int n_cfg, int M, int l, int max_atoms, const ModelW *MW,
                          const GraphView &G, const int *counters, int zero_slot, int excl_vol, float excl_sigma,
                          int excl_power, const float *s_in, const float *v_in, const float *phi, float *s_msg,
                          float *v_msg, float *e_excl) {
    dim3 grid(((n_cfg + 7) / 8) * 8 * NSLICE * M), blk(EDGE_THREADS);
    if (l == 0) {
        hipLaunchKernelGGL(k_edge_fwd_mfma<true>, grid, blk, edge_fwd_lds_bytes(max_atoms, true), st, N, l, MW, G,
                           counters, zero_slot, M, max_atoms, s_in, v_in, phi, s_msg, v_msg);
        if (excl_vol)
            hipLaunchKernelGGL(k_excl_vol, dim3((N + 255) / 256), dim3(256), 0, st, N, M, G, counters, excl_sigma,
                               excl_power, e_excl);
    } else {
        hipLaunchKernelGGL(k_edge_fwd_mfma<false>, grid, blk, edge_fwd_lds_bytes(max_atoms, false), st, N, l, MW, G,
                           counters, zero_slot, M, max_atoms, s_in, v_in, phi, s_msg, v_msg);
    }
}

}  // namespace vssr
