// nbr.hip — neighbor multigraph of a batch of independent configurations (gfx950).
//
// Replaces AtomsBatch.update_nbr_list (nff/io/ase.py; reference call sites
// mcmc/dynamics.py:129, mcmc/utils/misc.py:34-42) + the per-call trim to d <= cutoff
// (SURVEY.md Appendix A item 1).  Output: padded CSR by centre atom; every directed
// (i, j, S) with 0 < |x_j + S.cell - x_i| <= cutoff is one slot; a pair reachable through
// several periodic images occupies several slots.  Distances are decided in
// fp64 from the caller's fp64 positions, so the edge SET equals the oracle's bit for bit;
// the stored edge vector is narrowed to fp32 for the fp32 model.
#include "nbr_dev.h"

namespace vssr {

__global__ void k_wrap(int n, const double *__restrict__ pos, const int *__restrict__ atom_cfg,
                       const double *__restrict__ cell, const double *__restrict__ invcell,
                       const uint8_t *__restrict__ pbc, double *__restrict__ wpos, int *__restrict__ wrap) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    wrap_atom(i, pos, atom_cfg, cell, invcell, pbc, wpos, wrap);
}

// LPC lanes per centre atom (nbr_dev.h: nbr_row)
template <bool FILL, int LPC>
__global__ void __launch_bounds__(256)
k_nbr(int n, const double *__restrict__ wpos, const int *__restrict__ atom_cfg,
      const int *__restrict__ cfg_start, const double *__restrict__ cell, const double *__restrict__ invcell,
      const int *__restrict__ nimg, double rc2, int *__restrict__ deg, const int *__restrict__ row_start, float4 *__restrict__ edge,
      int *__restrict__ edge_S, long long slot_cap, unsigned long long *__restrict__ hits_buf, int hits_stride,
      const unsigned char *__restrict__ active) {
    const int i = blockIdx.x * (256 / LPC) + threadIdx.x / LPC;
    if (i >= n) return;
    nbr_row<FILL, LPC>(i, wpos, atom_cfg, cfg_start, cell, invcell, nimg, rc2, deg, row_start, edge, edge_S, slot_cap, hits_buf, hits_stride,
                       active);
}

// exclusive scan of padded degrees -> row_start ; counters[0] = slots, [1] = real edges.  Tiles of 4096 atoms (4 consecutive
// atoms per thread, int4 loads; wave-level DPP scans + one LDS exchange), one workgroup per tile, two launches: k_scan_tiles
// writes every tile's (padded, real) sums, k_scan_rows adds up the sums of the tiles in front of its own (a few hundred
// values at most) and scans its tile.  (One workgroup walking all tiles took 72 us for the 197 k atoms of 4 096 GaN chains,
// 27 us for the 66 k atoms of the PaiNN bench: 1.5 us of exposed load latency + two barriers per tile.)
struct ScanTile {
    int d[4], pd[4], p, real;
};
__device__ __forceinline__ ScanTile scan_tile_load(int n, const int *__restrict__ deg, int i0) {
    ScanTile s;
    if (i0 + 3 < n) {
        const int4 v = *reinterpret_cast<const int4 *>(deg + i0);   // (deg is 256-byte aligned, i0 a multiple of 4)
        s.d[0] = v.x; s.d[1] = v.y; s.d[2] = v.z; s.d[3] = v.w;
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) s.d[u] = i0 + u < n ? deg[i0 + u] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) s.pd[u] = i0 + u < n ? max((s.d[u] + 3) & ~3, 8) : 0;
    s.p = s.pd[0] + s.pd[1] + s.pd[2] + s.pd[3];
    s.real = s.d[0] + s.d[1] + s.d[2] + s.d[3];
    return s;
}

__global__ void __launch_bounds__(1024)
k_scan_tiles(int n, const int *__restrict__ deg, int *__restrict__ tile_sums /*[tiles][2]*/) {
    __shared__ int wsum[16], wreal[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const ScanTile s = scan_tile_load(n, deg, blockIdx.x * 4096 + 4 * t);
    const int x = wave_incl_scan_i32(s.p), xr = wave_incl_scan_i32(s.real);
    if (lane == 63) { wsum[w] = x; wreal[w] = xr; }
    __syncthreads();
    if (t == 0) {
        int a = 0, b = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { a += wsum[k]; b += wreal[k]; }
        tile_sums[2 * blockIdx.x] = a;
        tile_sums[2 * blockIdx.x + 1] = b;
    }
}

__global__ void __launch_bounds__(1024)
k_scan_rows(int n, const int *__restrict__ deg, int *__restrict__ row_start, int *__restrict__ counters,
            long long slot_cap, const int *__restrict__ tile_sums) {
    __shared__ int wsum[16], wreal[16];
    __shared__ long long wrun[16], wrun_real[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, tile = blockIdx.x;
    const bool last = tile == (int)gridDim.x - 1;
    // slots (and, in the last workgroup, real edges) in front of this tile
    long long run = 0, run_real = 0;
    for (int k = t; k < tile; k += 1024) {
        run += tile_sums[2 * k];
        if (last) run_real += tile_sums[2 * k + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        run += __shfl_xor(run, o);
        run_real += __shfl_xor(run_real, o);
    }
    if (lane == 0) { wrun[w] = run; wrun_real[w] = run_real; }
    const ScanTile s = scan_tile_load(n, deg, tile * 4096 + 4 * t);
    const int x = wave_incl_scan_i32(s.p);   // inclusive scan of the thread sums inside the wave
    if (lane == 63) wsum[w] = x;
    if (!tile_sums) {   // (uniform) single-tile launch: the real edge count comes from here as well
        const int xr = wave_incl_scan_i32(s.real);
        if (lane == 63) wreal[w] = xr;
    }
    __syncthreads();
    run = 0; run_real = 0;
    int woff = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        run += wrun[k];
        run_real += wrun_real[k];
        if (k < w) woff += wsum[k];
    }
    const int i0 = tile * 4096 + 4 * t;
    int o = (int)(run + woff + x - s.p);   // first slot of this thread's first atom
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (i0 + u < n) row_start[i0 + u] = o;
        o += s.pd[u];
    }
    if (last && t == 0) {
        // (a batch of one tile is launched without k_scan_tiles: its sums are this workgroup's own wave sums)
        long long own = 0, own_real = 0;
        if (tile_sums) { own = tile_sums[2 * tile]; own_real = tile_sums[2 * tile + 1]; }
        else {
            for (int k = 0; k < 16; ++k) { own += wsum[k]; own_real += wreal[k]; }
        }
        const long long total = run + own, total_real = run_real + own_real;
        row_start[n] = (int)total;
        counters[0] = (int)total;
        counters[1] = (int)total_real;
        counters[2] = (total > slot_cap - 64 || total > 2147483000LL) ? 1 : 0;
    }
}

// reverse-edge slot (nbr_dev.h: rev_row): LPC lanes per centre, lane per slot
template <int LPC>
__global__ void __launch_bounds__(256)
k_rev(int n, const int *__restrict__ row_start, const float4 *__restrict__ edge,
      const int *__restrict__ edge_S, int *__restrict__ rev, const int *__restrict__ counters, ActiveView av) {
    const int i = blockIdx.x * (256 / LPC) + threadIdx.x / LPC;
    if (i >= n || counters[2] || !av.atom(i)) return;
    rev_row<LPC>(i, row_start, edge, edge_S, rev);
}

// Per-slot geometry tables for the PaiNN edge kernels: unit vector + chain-local neighbor index, edge length,
// and the radial basis rho_k(d) = sin((k+1) pi d / rc)/d * fc(d) (k < 20), rho_20 = fc(d), with its derivative,
// stored in the order the MFMA operand wants them: table[slot][kq][ks] = rho_{kq + 4 ks}  (kq = lane >> 4, ks < 5);
// entry ks = 5 of every quarter holds the envelope fc (resp. its derivative).
// One thread per (slot, kq).  sin/cos of the multiples come from one sincos + a rotation recurrence.
// Operand-ready record of one (slot, quarter) for v_mfma_f32_16x16x32_f16 (write_f16_record).  Every value is split into
// two fp16 pieces x = h + l (h = fp16(x), l = fp16(x - h); 22 mantissa bits, fp16 subnormals are honoured by the matrix
// core -- tools/micro/mfma_f16_denorm.hip).  The edge kernels accumulate Wh.L + Wl.H + Wh.H: the dropped product is 2^-22
// relative, and measured against fp64 the 3-product result is as accurate as a plain fp32 dot product (max 2.0e-7 vs
// 1.9e-7 on the real weights).
__device__ __forceinline__ void split2_f16(float x, unsigned &h, unsigned &l) {
    const _Float16 hh = (_Float16)x;
    const _Float16 ll = (_Float16)(x - (float)hh);
    h = __builtin_bit_cast(unsigned short, hh);
    l = __builtin_bit_cast(unsigned short, ll);
}
// Table layout (rho16 / drho16), in 16-byte units: unit(slot, kq, piece) = (slot >> 2) * 32 + piece * 16 + kq * 4 + (slot & 3).
// The edge kernels always consume an aligned quad of 4 slots with 4 consecutive lanes; with the quad's entries interleaved
// like this those 4 lanes read 64 contiguous bytes (one L1 access) instead of 4 accesses 128 bytes apart: the kernels
// were bound by the L1 tag-lookup rate (~1 access / clock / CU; PMC: TA busy 80-90 %), not by bytes.
__device__ __forceinline__ size_t f16_unit(size_t slot, int kq, int piece) {
    return (slot >> 2) * 32 + (size_t)piece * 16 + (size_t)kq * 4 + (slot & 3);
}
// Record of one (slot, quarter), 32 bytes = two 16-byte units.  The quarter owns the radial indices k = kq + 4 t (t < 5): their
// three products per index (15) and one of the three envelope / bias products fill the 16 K entries a lane holds in TWO matrix
// instructions (painn_edge_mfma.hip, build_wd16 has the weight side):
//   unit 0 = [l0 l1 l2 l3 l4 env h0 h1]                    the first B operand as it stands (env: fc_l | fc_h | fc_h | 0 by quarter)
//   unit 1 = [h2 h3 h4 h4 | scalar (fp32) | neighbor]      the second operand [h2 h3 h4 h4 h0 h1 h2 h3] is put together from both
//                                                          units in registers (two moves)
// scalar = one of {u_x, u_y, u_z, 1 / d} (quarter 0 .. 3) as a plain fp32 word -- an all-gather on the row swaps hands all four to
// every lane of the slot -- and the last word = the chain-local neighbor index.  No separate record loads in the hot loops.
__device__ __forceinline__ void write_f16_record(const float (&v)[5], float env, uint4 *__restrict__ tab, size_t slot, int kq,
                                                 float spare, unsigned jbits) {
    unsigned h[6], l[6];
#pragma unroll
    for (int k = 0; k < 5; ++k) split2_f16(v[k], h[k], l[k]);
    split2_f16(env, h[5], l[5]);
    auto pk = [](unsigned lo, unsigned hi) { return lo | (hi << 16); };
    const unsigned env_piece = kq == 0 ? l[5] : kq == 3 ? 0u : h[5];   // partners: bd_h | bd_l | bd_h | 0 (build_wd16)
    tab[f16_unit(slot, kq, 0)] = make_uint4(pk(l[0], l[1]), pk(l[2], l[3]), pk(l[4], env_piece), pk(h[0], h[1]));
    tab[f16_unit(slot, kq, 1)] = make_uint4(pk(h[2], h[3]), pk(h[4], h[4]), __float_as_uint(spare), jbits);
}

// sum over the 16 lanes of a wave that share lane & 3 (the threads of one table quarter), result in all of them: two
// rotations inside the 16-lane rows (DPP row_ror 4, 8), then the four rows (gfx950 row swaps) -- a fixed order
__device__ __forceinline__ float quarter_class_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xF, 0xF, true));   // row_ror:4
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, true));   // row_ror:8
    const unsigned u = __float_as_uint(x);
    const auto r32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float y = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
    const unsigned v = __float_as_uint(y);
    const auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
}

// NZ > 0: the layer-0 species factorisation (painn_l0.hip) needs, per centre and neighbor species z, the 4 x 24 block
// T[z][comp][kappa] = sum over the centre's slots with species z of rho[kappa] * {1, u_x, u_y, u_z}[comp].  The thread of
// (slot, quarter) holds exactly the six rho values of its quarter, so the block is accumulated here, in registers (a
// separate kernel re-read the fp32 table: 0.19 ms per evaluation), and reduced once per centre over the 16 threads of a
// quarter.  NZ = 0: no accumulation (non-factorised layer 0, or more species than the register budget covers).
template <int NZ>
__global__ void __launch_bounds__(64) k_edge_geom(int n_atoms, const int *__restrict__ row_start, const int *__restrict__ atom_cfg,
                            const int *__restrict__ cfg_start, const float4 *__restrict__ edge,
                            const int *__restrict__ counters, float rc, float excl_sigma, int excl_power,
                            float4 *__restrict__ erec, float *__restrict__ rho,
                            float2 *__restrict__ dist2, uint4 *__restrict__ rho16, uint4 *__restrict__ drho16,
                            const int *__restrict__ Z, const int *__restrict__ zmap, unsigned char *__restrict__ zslot,
                            float *__restrict__ e_excl, const unsigned char *__restrict__ active, float *__restrict__ l0T) {
    if (counters[2]) return;
    float tacc[NZ > 0 ? NZ : 1][4][6];
#pragma unroll
    for (int z = 0; z < (NZ > 0 ? NZ : 1); ++z)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < 6; ++k) tacc[z][c][k] = 0.f;
    const int i = blockIdx.x;                 // centre atom
    if (active && !active[atom_cfg[i]]) return;
    const int a0 = cfg_start[atom_cfg[i]];
    const int e0 = row_start[i], e1 = row_start[i + 1];
    float ex = 0.f;   // excluded-volume energy of the centre: sum over its slots of (sigma / d)^p
    for (int t = threadIdx.x; t < (e1 - e0) * 4; t += blockDim.x) {
        const int slot = e0 + (t >> 2), kq = t & 3;
        const float4 ed = edge[slot];
        const int j = __float_as_int(ed.w);
        float r[6], dr[6], inv;   // this quarter's values (radial_quarter, vssr_internal.h)
        bool valid;
        radial_quarter(ed, kq, rc, r, dr, inv, valid);
        const float fc = r[5], dfc = dr[5];
        if (rho) {   // fp32 table: only the layer-0 kernel of batches with more than 4 species reads it (painn_l0.hip k_l0_table)
            float2 *rg = reinterpret_cast<float2 *>(rho + (size_t)slot * 24 + kq * 6);
#pragma unroll
            for (int q = 0; q < 3; ++q) rg[q] = make_float2(r[2 * q], r[2 * q + 1]);
        }
        {
            const float rv[5] = {r[0], r[1], r[2], r[3], r[4]}, dv[5] = {dr[0], dr[1], dr[2], dr[3], dr[4]};
            const float scal = kq == 0 ? ed.x * inv : kq == 1 ? ed.y * inv : kq == 2 ? ed.z * inv : (valid ? inv : -1.f);
            write_f16_record(rv, fc, rho16, (size_t)slot, kq, scal, (unsigned)min(valid ? j - a0 : 0, 0x7BFF));
            write_f16_record(dv, dfc, drho16, (size_t)slot, kq, 0.f, 0u);
        }
        // species index of the neighbor (layer-0 factorisation, painn_l0.hip); pads / unmapped species: 255
        const int zi = valid ? zmap[Z[j]] : -1;
        if constexpr (NZ > 0) {
            const float uc[4] = {1.f, ed.x * inv, ed.y * inv, ed.z * inv};
#pragma unroll
            for (int z = 0; z < NZ; ++z) {
                const float mz = zi == z ? 1.f : 0.f;   // (pads: rho = 0 and no species matches)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float w = mz * uc[c];
#pragma unroll
                    for (int k = 0; k < 6; ++k) tacc[z][c][k] = fmaf(r[k], w, tacc[z][c][k]);
                }
            }
        }
        if (kq == 0) {
            zslot[slot] = (unsigned char)(zi >= 0 ? zi : 255);
            if (erec) erec[slot] = make_float4(ed.x * inv, ed.y * inv, ed.z * inv, __int_as_float(valid ? j - a0 : 0));
            // (sigma / d)^p with the integer p by squaring (p = 12: 5 multiplications instead of the ~60 instructions of powf, which
            // every lane of the wave executes for the one lane in four that needs it); agrees with pow to ~1e-7 relative
            float rep = 0.f;
            if (valid) {
                float base = excl_sigma * inv;
                rep = 1.f;
                for (int pw = excl_power; pw > 0; pw >>= 1) {   // uniform trip count
                    if (pw & 1) rep *= base;
                    base *= base;
                }
            }
            ex += rep;
            dist2[slot] = valid ? make_float2(inv, -(float)excl_power * rep * inv) : make_float2(-1.f, 0.f);
        }
    }
    ex = wave_sum_f32(ex);   // fixed order (DPP + row swaps), so the sum does not depend on batching
    if (threadIdx.x == 0) e_excl[i] = ex;
    if constexpr (NZ > 0) {
        // Sum over the 16 lanes of a quarter class (lane & 3) as a REDUCE-SCATTER: the 24 NZ values are halved across the two
        // 32-lane halves (v_permlane32_swap: one swap + one add per PAIR of values), halved again across odd / even 16-lane rows
        // (v_permlane16_swap), and only the remaining quarter is summed inside the row (two DPP rotations): 2 x 24 NZ instructions
        // instead of 10 per value, a fixed order.  Row rho ends up with values [rho Q, (rho + 1) Q), Q = 6 NZ, i.e. the (species,
        // component) blocks rho NZ .. rho NZ + NZ - 1, complete in all of its lanes; its four lanes with slot position 0 write them.
        constexpr int CNT = NZ * 24, H = CNT / 2, Q = CNT / 4;
        float val[CNT];
#pragma unroll
        for (int z = 0; z < NZ; ++z)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int k = 0; k < 6; ++k) val[(z * 4 + c) * 6 + k] = tacc[z][c][k];
        float half[H], quart[Q];
#pragma unroll
        for (int v = 0; v < H; ++v) {
            const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(val[v]), __float_as_uint(val[v + H]), false, false);
            half[v] = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
        }
#pragma unroll
        for (int w = 0; w < Q; ++w) {
            const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(half[w]), __float_as_uint(half[w + Q]), false, false);
            float x = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, true));   // row_ror:8
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xF, 0xF, true));   // row_ror:4
            quart[w] = x;
        }
        const int lane = threadIdx.x, rho = lane >> 4, kq = lane & 3;
        if ((lane & 12) == 0) {   // slot position 0 of every row: lane = 16 rho + kq
#pragma unroll
            for (int b = 0; b < NZ; ++b) {   // block zc = rho NZ + b = (species z, component c)
                const int zc = rho * NZ + b, z = zc >> 2, c = zc & 3;
                float2 *dst = reinterpret_cast<float2 *>(l0T + ((size_t)i * NZ + z) * 96 + c * 24 + kq * 6);
                dst[0] = make_float2(quart[6 * b], quart[6 * b + 1]);
                dst[1] = make_float2(quart[6 * b + 2], quart[6 * b + 3]);
                dst[2] = make_float2(quart[6 * b + 4], quart[6 * b + 5]);
            }
        }
    }
}

// Work list of the MFMA edge kernels: per chain, the centres sorted by padded slot count (descending; ties by index) as
// {centre (chain-local), first slot, padded slot count, 0}.  Four consecutive entries form a "bundle" that one wave
// walks in lock step (4 slots per centre and step), so the four centres finish within the same step and their results
// are reduced / written with all lanes active.  One workgroup per chain, O(N^2) ranking on LDS broadcasts.
// Ascending bitonic sort of 2048 keys in LDS by the whole workgroup (any power-of-two width up to 1024 threads x 1 exchange each): the
// bundle tables rank a chain's centres by (length descending, index ascending) = ascending key ((4095 - len) << 11 | index); an
// O(n^2) ranking was 0.2 .. 0.5 ms of the neighbor stage for 1 000 .. 1 400-atom chains.  Unused keys = 0xFFFFFFFF.
constexpr int BSORT_N = 2048;
__device__ __forceinline__ void lds_bitonic_sort(unsigned *keys) {
    for (int k = 2; k <= BSORT_N; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < BSORT_N / 2; t += blockDim.x) {
                const int i = 2 * t - (t & (j - 1)), l = i + j;   // the pair (i, i + j) handled by exchange t
                const unsigned a = keys[i], b2 = keys[l];
                const bool up = (i & k) == 0;
                if ((a > b2) == up) { keys[i] = b2; keys[l] = a; }
            }
            __syncthreads();
        }
}

__global__ void __launch_bounds__(256)
k_bundle_sort(const int *__restrict__ cfg_start, const int *__restrict__ row_start, const int *__restrict__ counters,
              int4 *__restrict__ bundle, const unsigned char *__restrict__ active, int sort_keys) {
    extern __shared__ int sdeg[];
    if (counters[2] || (active && !active[blockIdx.x])) return;
    const int b = blockIdx.x, a0 = cfg_start[b], n = cfg_start[b + 1] - a0;
    for (int c = threadIdx.x; c < n; c += blockDim.x) sdeg[c] = row_start[a0 + c + 1] - row_start[a0 + c];
    __syncthreads();
    if (n <= BSORT_N && sort_keys) {   // (same order as the ranking below: length descending, index ascending)
        unsigned *keys = reinterpret_cast<unsigned *>(sdeg) + n;
        for (int c = threadIdx.x; c < BSORT_N; c += blockDim.x) keys[c] = c < n ? ((unsigned)(4095 - min(sdeg[c], 4095)) << 11) | (unsigned)c : 0xFFFFFFFFu;
        __syncthreads();
        lds_bitonic_sort(keys);
        for (int r = threadIdx.x; r < n; r += blockDim.x) {
            const int c = (int)(keys[r] & 2047u);
            bundle[a0 + r] = make_int4(c, row_start[a0 + c], sdeg[c], 0);
        }
        return;
    }
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        const int d = sdeg[c];
        int rank = 0;
        for (int o = 0; o < n; ++o) {
            const int od = sdeg[o];
            rank += (od > d || (od == d && o < c)) ? 1 : 0;
        }
        bundle[a0 + rank] = make_int4(c, row_start[a0 + c], d, 0);
    }
}

// Per-pass bundle tables of the multi-pass neighbor sum (painn_edge_mfma.hip, k_edge_fwd_mfma<.., SUB>): chains of `list` only.  The
// chain's atoms are cut into P = sub_passes(n) equal ranges; a row is sorted by neighbor, so the slots whose neighbor lies in range p
// are a contiguous piece [k_p, k_{p+1}) of the row.  Windows are cut at quad boundaries -- quads [floor(k_p / 4), ceil(k_{p+1} / 4)) --
// and the kernel sends slots of another range that fall into a window to its zero row.  A centre without a neighbor in range p gets
// an empty window there.  Every table is ranked by window length like the full table by row length; n_entries[p - 1][chain] = centres
// with a non-empty window in pass p >= 1 (the head of that table).  One workgroup per chain; LDS: SUB_MAX_PASSES x n ints.
__global__ void __launch_bounds__(1024)
k_bundle_sort_sub(const int *__restrict__ list, const int *__restrict__ cfg_start, const int *__restrict__ row_start,
                  const float4 *__restrict__ edge, const int *__restrict__ counters, int N, int n_cfg, int chunk_max,
                  int4 *__restrict__ bundle_sub, int *__restrict__ n_entries, const unsigned char *__restrict__ active) {
    extern __shared__ int sdeg[];   // [P][n] window lengths, [P][n] first slots (relative to the row), 2 048 sort keys
    const int b = list[blockIdx.x];
    if (counters[2] || (active && !active[b])) return;
    const int a0 = cfg_start[b], n = cfg_start[b + 1] - a0;
    const int P = sub_passes(n, chunk_max), chunk = sub_chunk(n, chunk_max);
    int *len = sdeg, *first = sdeg + SUB_MAX_PASSES * n;
    __shared__ int cnt[SUB_MAX_PASSES];   // (LDS counters: thousands of global atomics on one address cost more than the rest of the kernel)
    if (threadIdx.x < SUB_MAX_PASSES) cnt[threadIdx.x] = 0;
    __syncthreads();
    // 16 lanes per centre (one DPP row): every lane counts, over its slots, the neighbors below each range boundary; the row is sorted
    // by neighbor, so the count IS the index k_p of the first slot of range p (a binary search per boundary was ten dependent L2 round
    // trips per centre and boundary: most of this kernel's time)
    const int lane = threadIdx.x & 15;
    for (int c = threadIdx.x >> 4; c < n; c += blockDim.x >> 4) {
        const int rs = row_start[a0 + c], re = row_start[a0 + c + 1];
        int below[SUB_MAX_PASSES];   // [p]: real slots with neighbor < (p + 1) chunk; [P - 1]: all real slots
#pragma unroll
        for (int p = 0; p < SUB_MAX_PASSES; ++p) below[p] = 0;
        for (int e = rs + lane; e < re; e += 16) {
            const int j = __float_as_int(edge[e].w);
            if (j < 0) continue;
            const int jl = j - a0;
#pragma unroll
            for (int p = 0; p < SUB_MAX_PASSES; ++p) below[p] += (p + 1 >= P || jl < (p + 1) * chunk) ? 1 : 0;
        }
#pragma unroll
        for (int p = 0; p < SUB_MAX_PASSES; ++p) {   // sum over the 16 lanes of the row (every lane ends with the total)
            int x = below[p];
            x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); x += __shfl_xor(x, 4); x += __shfl_xor(x, 8);
            below[p] = x;
        }
        if (lane == 0) {
            int kprev = 0;
            for (int p = 0; p < P; ++p) {
                const int k = below[p];                          // first slot (relative) beyond range p
                const bool any = k > kprev;
                const int f = kprev & ~3, last = p + 1 < P ? min((k + 3) & ~3, re - rs) : re - rs;   // (the last window ends with the row: pads carry zero entries)
                len[p * n + c] = any ? last - f : 0;
                first[p * n + c] = f;
                if (any && p > 0) atomicAdd(&cnt[p], 1);
                kprev = k;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x >= 1 && threadIdx.x < SUB_MAX_PASSES) n_entries[(size_t)(threadIdx.x - 1) * n_cfg + b] = cnt[threadIdx.x];
    unsigned *keys = reinterpret_cast<unsigned *>(sdeg + 2 * SUB_MAX_PASSES * n);
    for (int p = 0; p < P; ++p) {
        const int *lp = len + p * n;
        __syncthreads();   // (the previous pass is done reading the keys)
        for (int c = threadIdx.x; c < BSORT_N; c += blockDim.x) keys[c] = c < n ? ((unsigned)(4095 - min(lp[c], 4095)) << 11) | (unsigned)c : 0xFFFFFFFFu;
        __syncthreads();
        lds_bitonic_sort(keys);
        for (int r = threadIdx.x; r < n; r += blockDim.x) {
            const int c = (int)(keys[r] & 2047u);
            bundle_sub[(size_t)p * N + a0 + r] = make_int4(c, row_start[a0 + c] + first[p * n + c], lp[c], 0);
        }
    }
}

// Enqueue the neighbor build on the handle's stream (no host synchronisation).  If the slot
// capacity is exceeded, counters[2] is set on the device, every consumer kernel exits early on
// that flag, and the host (vssr_batch_download / vssr_synchronize) grows the buffers and reruns.
int build_neighbors(vssr_handle *h, double cutoff) {
    const int n = h->n_atoms;
    hipStream_t st = h->stream;
    if (h->d_wpos.ensure(sizeof(double) * 3 * n) || h->d_wrap.ensure(sizeof(int) * 3 * n) ||
        h->d_deg.ensure(sizeof(int) * n) || h->d_row_start.ensure(sizeof(int) * (n + 1)) ||
        h->d_counters.ensure(sizeof(int) * 4) || h->d_tile_sums.ensure(sizeof(int) * 2 * ((size_t)n / 4096 + 1)))
        return set_err(h, VSSR_E_NOMEM, "neighbor buffers: out of device memory");
    if (h->slot_cap < (int64_t)n * h->cap_per_atom + 64) h->slot_cap = (int64_t)n * h->cap_per_atom + 64;
    if (h->d_edge.ensure(sizeof(float4) * h->slot_cap) || h->d_edge_S.ensure(sizeof(int) * h->slot_cap) ||
        h->d_rev.ensure(sizeof(int) * h->slot_cap))
        return set_err(h, VSSR_E_NOMEM, "edge buffers: out of device memory");
    // scratch for the hit masks of the counting pass (skipped for very large single configurations / thin cells)
    unsigned long long *hits_buf = nullptr;
    const int hits_stride = (h->max_cfg_atoms + 63) & ~63;
    if (h->max_images <= 64 && (size_t)n * hits_stride * 8 <= ((size_t)1 << 30) &&
        !h->d_hits.ensure((size_t)n * hits_stride * 8))
        hits_buf = h->d_hits.as<unsigned long long>();
    h->prof.begin(KC_NBR, st);
    dim3 blk(128), grd((n + 127) / 128);
    // lanes per centre.  Measured on the PaiNN bench (profiles/r04/ab_nbr_lanes_per_centre.txt) and the GaN workload: 16-lane rows
    // beat one wave per centre in the search at every chain size (260-atom chains: 4 full chunks of 64 + one with 4 lanes; 48-atom
    // chains: 48 of 64 lanes, once), the reverse-slot search likes 32 lanes on PaiNN rows (~41 slots) and 16 on the 8-slot rows of
    // the analytic potentials.  (The 32- and 64-lane search forms gave the same list bit for bit and are not built any more.)
    const int lpc_rev = h->kind != 1 ? 16 : 32;
    dim3 wblk(256);
    auto grid_for = [&](int lpc) { return dim3((n + 256 / lpc - 1) / (256 / lpc)); };
#define NBR_ARGS(FILLING) n, h->d_wpos.as<double>(), h->d_atom_cfg.as<int>(), h->d_cfg_start.as<int>(), h->d_cell.as<double>(),          \
        h->d_invcell.as<double>(), h->d_nimg.as<int>(), cutoff * cutoff, h->d_deg.as<int>(),                                            \
        (FILLING) ? h->d_row_start.as<int>() : (const int *)nullptr, (FILLING) ? h->d_edge.as<float4>() : (float4 *)nullptr,            \
        (FILLING) ? h->d_edge_S.as<int>() : (int *)nullptr, (FILLING) ? (long long)h->slot_cap : (long long)0, hits_buf, hits_stride,   \
        h->active_mask
    hipLaunchKernelGGL(k_wrap, grd, blk, 0, st, n, h->d_pos.as<double>(), h->d_atom_cfg.as<int>(),
                       h->d_cell.as<double>(), h->d_invcell.as<double>(), h->d_pbc.as<uint8_t>(),
                       h->d_wpos.as<double>(), h->d_wrap.as<int>());
    hipLaunchKernelGGL((k_nbr<false, 16>), grid_for(16), wblk, 0, st, NBR_ARGS(false));
    const int n_tiles = n > 0 ? (n + 4095) / 4096 : 1;   // (an empty batch still writes its counters)
    if (n_tiles > 1)
        hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles), dim3(1024), 0, st, n, h->d_deg.as<int>(), h->d_tile_sums.as<int>());
    hipLaunchKernelGGL(k_scan_rows, dim3(n_tiles), dim3(1024), 0, st, n, h->d_deg.as<int>(),
                       h->d_row_start.as<int>(), h->d_counters.as<int>(), (long long)h->slot_cap,
                       n_tiles > 1 ? h->d_tile_sums.as<int>() : (const int *)nullptr);
    hipLaunchKernelGGL((k_nbr<true, 16>), grid_for(16), wblk, 0, st, NBR_ARGS(true));
#undef NBR_ARGS
    const ActiveView rev_av{h->active_mask, h->d_atom_cfg.as<int>()};
#define REV_ARGS n, h->d_row_start.as<int>(), h->d_edge.as<float4>(), h->d_edge_S.as<int>(), h->d_rev.as<int>(), h->d_counters.as<int>(), rev_av
    if (lpc_rev == 16) hipLaunchKernelGGL(k_rev<16>, grid_for(16), wblk, 0, st, REV_ARGS);
    else hipLaunchKernelGGL(k_rev<32>, grid_for(32), wblk, 0, st, REV_ARGS);
#undef REV_ARGS
    if (h->kind == 1) {   // PaiNN: per-slot geometry tables shared by all layers / models / slices
        // layer-0 factorisation with at most 4 species: its T blocks are accumulated by k_edge_geom (h->l0T_by_geom); with more species
        // a separate kernel builds them from the fp32 table and the per-slot unit vectors, which are only written for it
        const int nzf = (h->l0_enabled && h->l0_nz >= 1 && h->l0_nz <= 4) ? h->l0_nz : 0;
        const bool want_f32 = h->l0_enabled && nzf == 0;
        if ((want_f32 && (h->d_erec.ensure(sizeof(float4) * h->slot_cap) || h->d_rho.ensure(sizeof(float) * 24 * h->slot_cap))) ||
            h->d_dist.ensure(sizeof(float2) * h->slot_cap) ||
            h->d_rho16.ensure(sizeof(uint4) * 8 * h->slot_cap) || h->d_drho16.ensure(sizeof(uint4) * 8 * h->slot_cap) ||
            h->d_zslot.ensure((size_t)h->slot_cap) || h->d_bundle.ensure(sizeof(int4) * (size_t)n) ||
            h->d_excl.ensure(sizeof(float) * (size_t)n))
            return set_err(h, VSSR_E_NOMEM, "edge geometry tables: out of device memory");
        // the last slot of the capacity is never used by the CSR (counters[2] flags slots > cap - 64): it is the
        // all-zero table entry that exhausted lanes of the edge kernels read
        // (nothing ever writes them: cleared once per allocation / capacity, not once per evaluation)
        const void *tabs[2] = {h->d_rho16.as<uint4>(), h->d_drho16.as<uint4>()};
        bool cleared = h->zero_entry_cap == h->slot_cap;
        for (int k = 0; k < 2; ++k) cleared = cleared && h->zero_entry_tab[k] == tabs[k];
        if (!cleared) {
            // fp16 tables: the last complete QUAD of the capacity is the all-zero entry (quad-interleaved layout, f16_unit)
            VSSR_HIP(h, hipMemsetAsync(h->d_rho16.as<uint4>() + 32 * (size_t)((h->slot_cap >> 2) - 1), 0, 32 * sizeof(uint4), st));
            VSSR_HIP(h, hipMemsetAsync(h->d_drho16.as<uint4>() + 32 * (size_t)((h->slot_cap >> 2) - 1), 0, 32 * sizeof(uint4), st));
            h->zero_entry_cap = h->slot_cap;
            for (int k = 0; k < 2; ++k) h->zero_entry_tab[k] = tabs[k];
        }
        if (nzf && h->d_l0T.ensure(sizeof(float) * (size_t)n * nzf * 96))
            return set_err(h, VSSR_E_NOMEM, "layer-0 factorisation buffers: out of device memory");
        h->l0T_by_geom = nzf > 0;
#define LAUNCH_GEOM(NZ)                                                                                                          \
        hipLaunchKernelGGL(k_edge_geom<NZ>, dim3(n), dim3(64), 0, st, n, h->d_row_start.as<int>(), h->d_atom_cfg.as<int>(),      \
                           h->d_cfg_start.as<int>(), h->d_edge.as<float4>(), h->d_counters.as<int>(), h->cutoff,               \
                           h->excl_sigma, h->excl_power, want_f32 ? h->d_erec.as<float4>() : (float4 *)nullptr,                \
                           want_f32 ? h->d_rho.as<float>() : (float *)nullptr, h->d_dist.as<float2>(), h->d_rho16.as<uint4>(), \
                           h->d_drho16.as<uint4>(), h->d_Z.as<int>(), h->d_zmap.as<int>(), h->d_zslot.as<unsigned char>(),     \
                           h->d_excl.as<float>(), h->active_mask, nzf ? h->d_l0T.as<float>() : (float *)nullptr)
        switch (nzf) {
            case 1: LAUNCH_GEOM(1); break;
            case 2: LAUNCH_GEOM(2); break;
            case 3: LAUNCH_GEOM(3); break;
            case 4: LAUNCH_GEOM(4); break;
            default: LAUNCH_GEOM(0); break;
        }
#undef LAUNCH_GEOM
        if ((size_t)h->max_cfg_atoms * sizeof(int) <= 48 * 1024) {   // chains of the MFMA edge kernels (LDS slices) are far smaller
            // (large chains: key sort in LDS instead of the O(n^2) ranking -- the same table; rows longer than 4 095 slots do not occur)
            const int sort_keys = h->max_cfg_atoms > 512 && h->max_cfg_atoms <= BSORT_N;
            hipLaunchKernelGGL(k_bundle_sort, dim3(h->n_cfg), dim3(256),
                               (size_t)h->max_cfg_atoms * sizeof(int) + (sort_keys ? BSORT_N * sizeof(unsigned) : 0), st,
                               h->d_cfg_start.as<int>(), h->d_row_start.as<int>(), h->d_counters.as<int>(),
                               h->d_bundle.as<int4>(), h->active_mask, sort_keys);
            // chains of the 4-feature class (788 .. 1 462 atoms): per-pass tables of the two-pass 8-feature forward kernel
            // per-pass bundle tables of the multi-pass forms: forward (chains of the 4-feature class) and reverse (class FS16P); the two
            // cut the chain into ranges of different size, so each has its own tables
            auto sub_tables = [&](vssr::DevBuf &buf, const int *list, int n_list, int max_atoms, int chunk) -> int {
                if (sub_passes(max_atoms, chunk) > SUB_MAX_PASSES) return set_err(h, VSSR_E_BADARG, "neighbor sub-range too small for this chain");
                if (buf.ensure(sizeof(int4) * SUB_MAX_PASSES * (size_t)n + sizeof(int) * (SUB_MAX_PASSES - 1) * (size_t)h->n_cfg))
                    return set_err(h, VSSR_E_NOMEM, "bundle tables: out of device memory");
                hipLaunchKernelGGL(k_bundle_sort_sub, dim3(n_list), dim3(1024),
                                   2 * SUB_MAX_PASSES * (size_t)max_atoms * sizeof(int) + BSORT_N * sizeof(unsigned), st, list,
                                   h->d_cfg_start.as<int>(), h->d_row_start.as<int>(), h->d_edge.as<float4>(), h->d_counters.as<int>(), n, h->n_cfg,
                                   chunk, buf.as<int4>(), reinterpret_cast<int *>(buf.as<int4>() + SUB_MAX_PASSES * (size_t)n), h->active_mask);
                return VSSR_OK;
            };
            if (h->fwd_two_pass && h->n_class[EDGE_CLASS_FS4] > 0) {
                int off = 0;
                for (int c = 0; c < EDGE_CLASS_FS4; ++c) off += h->n_class[c];
                const int rc2 = sub_tables(h->d_bundle_sub, h->d_class_list.as<int>() + off, h->n_class[EDGE_CLASS_FS4], h->max_class_atoms[EDGE_CLASS_FS4],
                                           h->sub_chunk_fwd ? h->sub_chunk_fwd : sub_chunk_max(h->fwd_two_pass));
                if (rc2) return rc2;
            }
            if (h->n_bclass[EDGE_BCLASS_FS16P] > 0) {
                int off = 0;
                for (int c = 0; c < EDGE_MFMA_CLASSES; ++c) off += h->n_class[c];
                for (int c = 0; c < EDGE_BCLASS_FS16P; ++c) off += h->n_bclass[c];
                const int rc2 = sub_tables(h->d_bundle_subb, h->d_class_list.as<int>() + off, h->n_bclass[EDGE_BCLASS_FS16P],
                                           h->max_bclass_atoms[EDGE_BCLASS_FS16P], h->sub_chunk_bwd ? h->sub_chunk_bwd : sub_chunk_max_bwd());
                if (rc2) return rc2;
            }
        }
    }
    h->prof.end(st);
    VSSR_HIP(h, hipMemcpyAsync(h->h_counters, h->d_counters.as<int>(), sizeof(int) * 4,
                               hipMemcpyDeviceToHost, st));
    return VSSR_OK;
}

}  // namespace vssr
