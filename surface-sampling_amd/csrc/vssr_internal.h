// vssr_internal.h — shared declarations of the gfx950 evaluation backend (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vssr_eval.h"

namespace vssr {

constexpr int F = 128;        // feat_dim (compiled value)
constexpr int F3 = 3 * F;
constexpr int MAX_LAYERS = 4;
constexpr int MAX_MODELS = 8;
constexpr int RB_MAX = 24;    // n_rbf + 1 (envelope/bias column) padded to a multiple of 4
constexpr int NODE_TILE = 8;  // atoms per workgroup in the node kernels
constexpr int L0_MAX_SPECIES = 8;  // layer-0 species factorisation is used when the batch has at most this many species

// ---- device-side views -------------------------------------------------------------------
struct LayerW {
    const float *W1, *W1t, *b1;  // [F][F], transposed [F][F]
    const float *W2, *W2t, *b2;  // [3F][F], transposed [F][3F]
    const float *Wd, *bd;        // [3F][R], [3F]
    const float *U, *Ut, *V, *Vt;  // [F][F]
    const float *W3, *W3t, *b3;  // [F][2F], transposed [2F][F]
    const float *W4, *W4t, *b4;  // [3F][F], transposed [F][3F]
    // W1, W2, U, V, W3, W4 (forward) and W1^T, W2^T, W4^T, W3^T, [U;V]^T (reverse) as 2-way fp16 pieces in
    // v_mfma_f32_16x16x32_f16 B-fragment order (pack_mfma_tiles16), 16-column tiles
    const uint4 *qW1, *qW2, *qU, *qV, *qW3, *qW4, *qW1t, *qW2t, *qW4t, *qW3t, *qUVt;
    const uint4 *wd16;           // radial-filter weights, 2-way fp16 split in MFMA A-operand order: [3F rows][4 quarters][operand 1, 2] (build_wd16)
};
struct ModelW {
    const float *embed;  // [n_embed][F]
    LayerW layer[MAX_LAYERS];
    const float *W5, *W5t, *b5, *w6, *b6;  // [H][F], [F][H], [H], [H], [1]
    const uint4 *qW5, *qW5t;               // readout matrices as fp16 pieces in MFMA fragment order (painn_node_mfma.hip)
};

// Chains that take part in an evaluation.  During a lock-step relaxation converged chains are switched off: every kernel
// tests its chain (or the chains of its atom tile) and leaves at once, so an iteration costs what the unconverged chains cost;
// the results of a switched-off chain stay those of its last evaluation.  mask == nullptr: all chains.
struct ActiveView {
    const unsigned char *mask;   // [n_cfg] 1 = evaluate
    const int *atom_cfg;         // [n_atoms]
    unsigned *sat = nullptr;     // [n_cfg] raised by the node kernels when an activation left the fp16-split range (mfma16.h SatTrack)
    __device__ __forceinline__ bool chain(int c) const { return !mask || mask[c]; }
    __device__ __forceinline__ bool atom(int i) const { return !mask || mask[atom_cfg[i]]; }
    // any active chain among atoms [a0, a1]?  (chains are contiguous atom ranges)
    __device__ __forceinline__ bool tile(int a0, int a1) const {
        if (!mask) return true;
        for (int c = atom_cfg[a0], c1 = atom_cfg[a1]; c <= c1; ++c)
            if (mask[c]) return true;
        return false;
    }
};

struct GraphView {  // neighbor multigraph of the resident batch (padded CSR by centre)
    int n_atoms;             // total atoms in the batch
    int n_cfg;
    const int *atom_cfg;     // [n_atoms] configuration (chain) of each atom
    const int *cfg_start;    // [n_cfg+1]
    const int *row_start;    // [n_atoms+1] first slot of each centre (slot counts are multiples of 4)
    const int *deg;          // [n_atoms] real degree (unpadded)
    const float4 *edge;      // [slots] {r_x, r_y, r_z, bitcast(j)} ; j < 0 marks a pad slot
    const int *rev;          // [slots] slot of the reverse edge (j -> i, -S)
    // per-slot geometry tables, computed once per evaluation and shared by every layer / model / feature slice
    const float4 *erec;      // [slots] {u_x, u_y, u_z, bitcast(j local to its chain)} ; pads: u = 0, j = 0 (same condition as rho, else null)
    const float *rho;        // [slots][4][6]  radial basis * envelope, [kq][ks] = rho_{kq+4ks}: only for the layer-0 kernel of batches with > 4 species (else null)
    const uint4 *rho16;      // operand-ready 2-way fp16 split of rho + the slot's scalars, quad-interleaved: [slot / 4][unit 0, 1][quarter][slot % 4] x 16 B (nbr.hip f16_unit, write_f16_record)
    const uint4 *drho16;     // same for d rho / d d
    const unsigned char *zslot;   // [slots] species index (zmap) of the neighbor, 255 for pads / unmapped
    const float2 *dist2;     // [slots] {1 / edge length (pads: -1), excluded-volume dE/dd = -p (sigma/d)^p / d (pads: 0)}
    const int4 *bundle;      // [n_atoms] per chain: centres sorted by padded degree (descending): {centre (chain-local), first slot, padded slot count, 0}
    const unsigned char *chain_class;   // [n_cfg] EDGE_BCLASS_*: which reverse neighbor kernels serve the chain (layers >= 1); a chain
                                        // gathers in the reverse pass exactly when it gathers in the forward pass
    ActiveView act;          // chains switched off by the relaxation driver
};

// Wave-wide integer prefix sum and float sum on the DPP network / the gfx950 row swaps (no LDS permutes: a __shfl is a ds_bpermute,
// ~100 cycles, and a scan / butterfly is six of them in a dependent chain).  Fixed orders, independent of what else runs.
#if defined(__HIPCC__)
__device__ __forceinline__ int wave_incl_scan_i32(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);    // row_shr:1 (lanes without a source add 0)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);    // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);    // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);    // row_shr:8   -> inclusive inside every 16-lane row
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return x;
}
__device__ __forceinline__ float xrow_sum_f32(float x) {   // sum over the four 16-lane rows (lanes l, l^16, l^32, l^48), result in every lane
    const unsigned u = __float_as_uint(x);
    const auto r32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float y = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
    const unsigned v = __float_as_uint(y);
    const auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
}
__device__ __forceinline__ float wave_sum_f32(float x) {   // result in every lane
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xF, 0xF, true));   // row_ror:4
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, true));   // row_ror:8
    const unsigned u = __float_as_uint(x);
    const auto r32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float y = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
    const unsigned v = __float_as_uint(y);
    const auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
}
#endif

// Radial basis of one slot for one table quarter kq (radial indices n = kq + 1 + 4 ks, ks = 0..4, then the envelope):
//   r[ks] = sin(n pi d / rc) / d * fc(d),  dr[ks] = d r[ks] / d d,  r[5] = fc,  dr[5] = fc'   
// ONE definition for the kernel that writes the per-slot tables (k_edge_geom, nbr.hip) and for the kernel that rebuilds the fp32
// values instead of reading them (k_l0_bwd, painn_l0.hip): both see bit-identical numbers (no contraction, same operation order).
// ed: {r_x, r_y, r_z, bitcast(j)}, j < 0 marks a pad slot (everything zero, inv = 0).
#if defined(__HIPCC__)
__device__ __forceinline__ void radial_quarter(const float4 &ed, int kq, float rc, float (&r)[6], float (&dr)[6], float &inv, bool &valid) {
    const float alpha = 3.14159265358979323846f / rc;
    valid = __float_as_int(ed.w) >= 0;
    const float d2 = fmaf(ed.z, ed.z, fmaf(ed.y, ed.y, ed.x * ed.x));
    const float d = valid ? sqrtf(d2) : 1.f;
    inv = valid ? 1.f / d : 0.f;
    float s1, c1;
    sincosf(alpha * d, &s1, &c1);
    const bool inside = valid && d < rc;
    const float fc = inside ? 0.5f * (c1 + 1.f) : 0.f;
    const float dfc = inside ? -0.5f * alpha * s1 : 0.f;
    const float s2 = 2.f * s1 * c1, c2 = fmaf(c1, c1, -s1 * s1);
    const float s3 = fmaf(s2, c1, c2 * s1), c3 = fmaf(c2, c1, -s2 * s1);
    const float s4 = 2.f * s2 * c2, c4 = fmaf(c2, c2, -s2 * s2);
    float sn = kq == 0 ? s1 : kq == 1 ? s2 : kq == 2 ? s3 : s4;
    float cn = kq == 0 ? c1 : kq == 1 ? c2 : kq == 2 ? c3 : c4;
    float nf = (float)(kq + 1);
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const float rb = sn * inv;
        r[ks] = rb * fc;
        dr[ks] = fmaf(nf * alpha * cn * inv - rb * inv, fc, rb * dfc);
        const float sn2 = fmaf(sn, c4, cn * s4), cn2 = fmaf(cn, c4, -sn * s4);
        sn = sn2; cn = cn2; nf += 4.f;
    }
    r[5] = fc;      // envelope (bias column) replicated in every quarter
    dr[5] = dfc;
}
#endif

struct StateView {  // activations of all models: index [m][atom][...]
    int n_atoms;
    int n_models;
    // forward
    float *s_in[MAX_LAYERS + 1];  // [M][N][F]   state entering layer l (l = L: final)
    float *v_in[MAX_LAYERS + 1];  // [M][N][3][F]
    float *phi[MAX_LAYERS];       // [M][N][3F]
    float *s_msg[MAX_LAYERS];     // after message block l
    float *v_msg[MAX_LAYERS];
    float *e_atom;                // [M][N]
    const float *e_excl;          // [N] excluded-volume part (geometry only: written by k_edge_geom, nbr.hip)
    // reverse
    float *sbar;                  // [M][N][F]     adjoint of s_in[l+1] / s_in[l]
    float *vbar;                  // [M][N][3][F]
    float *sbar_msg;              // adjoint of s_msg[l] (the fused reverse kernels alternate between sbar_msg and sbar)
    float *sbar_msg_l0;           // the buffer that holds the adjoint of s_msg[0] after a reverse pass
    float *vbar_msg;
    float *phibar;                // [M][N][3F]
    float4 *gbar;                 // [M][slots] dE/d r for the edge (j -> i) stored at slot (i, j)
};

// ---- host-side helpers ----------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {  // grow-only; contents are NOT preserved
        if (need <= bytes) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        size_t want = need + need / 4 + 256;
        if (hipMalloc(&p, want) != hipSuccess) return -1;
        bytes = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T *as() const { return reinterpret_cast<T *>(p); }
};

enum KernelClass {
    KC_NBR = 0,
    KC_EMBED,
    KC_MSG_MLP,
    KC_EDGE_FWD,
    KC_UPDATE_FWD,
    KC_READOUT,
    KC_UPDATE_BWD,
    KC_EDGE_BWD,
    KC_MSG_MLP_BWD,
    KC_FINALIZE,
    KC_TERSOFF,
    KC_L0_FWD,
    KC_L0_BWD,
    KC_COUNT
};
extern const char *const kKernelClassNames[KC_COUNT];

struct Profiler {
    bool enabled = false;
    struct Rec { int kc; hipEvent_t a, b; };
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    int64_t launches[KC_COUNT] = {0};
    double total_ms[KC_COUNT] = {0};
    hipEvent_t get_event();
    void begin(int kc, hipStream_t s);
    void end(hipStream_t s);
    void collect();  // stream must be synchronised
    void reset();
    void destroy();
};

}  // namespace vssr

struct vssr_handle {
    int kind = 0;  // 1 = PaiNN ensemble, 2 = Tersoff, 3 = EAM (funcfl)
    vssr_eam_grid eam_grid = {0, 0, 0.0, 0.0, 0.0};   // EAM: grids; spline tables live in ters_params
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    vssr::Profiler prof;

    int edge_impl = 1;  // 1 = LDS-slice + MFMA edge kernels for chains that fit LDS; 0 forces the gather kernels
                        // (VSSR_EDGE_IMPL=gather: debug knob, also the automatic path for very large chains)
    int max_cfg_atoms = 0;
    // chains by neighbor-sum path (EDGE_CLASS_*), fixed at upload from every chain's own atom count: class of every chain,
    // the chain lists of the matrix-pipe classes (concatenated in class order), counts and largest chain per class
    vssr::DevBuf d_chain_class, d_class_list;
    int n_class[5] = {0, 0, 0, 0, 0}, max_class_atoms[5] = {0, 0, 0, 0, 0};   // forward classes (EDGE_CLASS_*)
    int n_bclass[5] = {0, 0, 0, 0, 0}, max_bclass_atoms[5] = {0, 0, 0, 0, 0};   // reverse classes (EDGE_BCLASS_*); lists follow the forward lists
    int fs16_max_atoms = -1, fs8_max_atoms = -1, fs4_max_atoms = -1;   // test knobs (VSSR_EDGE_FS16_MAX / _FS8_MAX): lower the class limits (fs4: no knob, always the LDS limit)
    int max_images = 0;          // largest number of periodic images any configuration of the batch scans per pair

    // configuration
    int n_models = 0, n_rbf = 20, num_conv = 3, n_embed = 100, readout_hidden = 64;
    float cutoff = 5.f;
    int excl_vol = 1, excl_power = 12;
    float excl_sigma = 1.5f;
    double units_per_ev = 1.0;
    bool has_offset = false;
    double offset_const = 0.0;

    // weights
    vssr::DevBuf weights;        // all model blobs + transposed copies
    vssr::DevBuf wd16;           // fp16-split radial-filter weights in MFMA operand order
    vssr::DevBuf node16;         // fp16-split node-GEMM weights in MFMA fragment order
    vssr::DevBuf model_table;    // ModelW[n_models]
    vssr::DevBuf offset_per_z;   // double[n_embed]

    // tersoff
    int n_types = 0;
    double ters_cutmax = 0;
    vssr::DevBuf ters_params;    // double[nt^3][14]

    // resident batch
    bool batch_valid = false, ran = false;
    int n_cfg = 0, n_atoms = 0;
    std::vector<int> h_n_atoms, h_cfg_start;
    vssr::DevBuf d_pos, d_wpos, d_wrap, d_Z, d_atom_cfg, d_cfg_start, d_cell, d_invcell, d_nimg, d_pbc;
    vssr::DevBuf d_deg, d_row_start, d_edge, d_edge_S, d_rev, d_counters, d_tile_sums;
    vssr::DevBuf d_erec, d_rho, d_dist, d_rho16, d_drho16, d_zslot, d_bundle, d_excl;
    vssr::DevBuf d_bundle_sub;   // [2][n_atoms] per-pass bundle tables of the two-pass forward neighbor sum (chains of the 4-feature class)
    int bwd_multi_pass = 1;      // VSSR_EDGE_BWD_MPASS=0: chains of > 557 atoms take the 8- / 4-feature reverse kernels (round 4) instead of the
                                 // 16-feature kernel in several passes; 2: every chain of the matrix-pipe classes takes the multi-pass form (tests)
    int fwd_mpass_fs8 = 1;       // chains of 406 .. 787 atoms take the 16-feature multi-pass forward form (-9 % against the single-pass
                                 // 8-feature kernel of round 4, profiles/r05/NOTES_large_chains.md; 0 = that kernel, no knob any more)
    int sub_chunk_fwd = 0, sub_chunk_bwd = 0;   // VSSR_EDGE_SUB_CHUNK=n (tests): atoms per neighbor sub-range instead of what LDS holds
    vssr::DevBuf d_bundle_subb;  // per-pass bundle tables of the reverse multi-pass form (its ranges are larger than the forward's)
    int fwd_two_pass = 16;       // VSSR_EDGE_FWD_2PASS = 0 | 8 | 16: chains of 788 .. 1 462 atoms take the 4-feature forward kernel, or the 8- / 16-feature
                                 // kernel in several passes over neighbor sub-ranges
    vssr::DevBuf d_hits;         // neighbor search: per (centre, candidate) 64-bit hit masks of the counting pass
    // layer-0 species factorisation (painn_l0.hip)
    int l0_enabled = 1, l0_nz = 0;
    bool l0_used = false;                // last run used the factorised layer 0
    vssr::DevBuf d_l0A, d_l0At;          // [M][n_embed][2][24][F] and [M][n_embed][2][F][24], built at create
    vssr::DevBuf d_zmap, d_zlist;        // species index of Z (or -1), distinct Z of the resident batch
    vssr::DevBuf d_l0T, d_l0Q;
    // lock-step relaxation (relax.hip)
    vssr::DevBuf d_vel, d_fire, d_fixed, d_relax_steps, d_relax_conv, d_active, d_bfgs_q, d_bfgs_b;
    const unsigned char *active_mask = nullptr;   // set by relax_run for the duration of a relaxation
    int relax_regrows = 0;
    long long relax_lockstep = 0;   // lock-step evaluations of the batch launched by the last relaxation (vssr_batch_relax_counts)
    int relax_compactions = 0;      // live-chain compactions of the last CG relaxation
    vssr::DevBuf d_cmp;            // arena of the live-chain compaction (relax.hip)
    vssr::DevBuf d_cm;             // flags + per-chain evaluation counters of the chain-resident minimiser (chain_min.hip)
    long long relax_chain_evals = 0;   // chain-evaluations those launches actually dispatched (live-chain compaction: < lockstep x B)
    // trajectory recording of the lock-step relaxations (relax.hip k_traj_record): every traj_interval optimizer steps
    int traj_interval = 0, traj_records = 0, traj_B = 0, traj_N = 0;
    vssr::DevBuf d_traj_pos, d_traj_f, d_traj_e, d_traj_n;
    bool graph_partial = false;   // the resident neighbor graph / activations cover only the chains of the last relaxation iteration
    bool l0T_by_geom = false;     // this evaluation's layer-0 T blocks were written by k_edge_geom (<= 4 species)
    int64_t zero_entry_cap = -1;  // capacity / table addresses for which the all-zero table entries were last cleared
    const void *zero_entry_tab[2] = {nullptr, nullptr};
    int cap_per_atom = 64;        // initial neighbor capacity (slots per atom); vssr_debug_capacity
    int cm_cap_per_atom = 0;      // slots per atom the chain-resident CG minimiser's per-chain pools grew to (chain_min.hip)
    bool cap_tight = false;       // regrow to the exact need only (tests: forces repeated overflows)
    uint32_t last_want = 0;                       // outputs produced by the last run
    int64_t slot_cap = 0;
    int *h_counters = nullptr;   // pinned: [0] total slots, [1] total real edges, [2] overflow flag

    // state
    vssr::DevBuf d_state;        // one arena for all activations
    vssr::StateView sv;
    vssr::DevBuf d_gbar;
    vssr::DevBuf d_upd_save;     // forward intermediates of every update block for the reverse pass (update_save_bytes per layer)
    // Partial edge gradients of the reverse neighbor pass (fixed to mode 2 since round 6; modes 0 / 1 are what rounds 2 / 3 measured,
    // profiles/r03/NOTES_edge_traffic.md, and still serve the gather kernels' float4 records):
    //   0  one set of float4 buffers per model, the second reverse layer adds to it (read-modify-write), group 0 is reduced in place
    //   1  one set per reverse layer (no read-modify-write: TA relief in the second launch), float4 records
    //   2  one set per reverse layer, 12-byte records in a separate buffer, reduced into the final float4 buffer  (default)
    int gbar_mode = 2;
    vssr::DevBuf d_gpart;        // mode 2: [M][groups][slot_cap][3] floats
    int debug_keep = 0;          // VSSR_DEBUG_KEEP=1: also materialise what only vssr_debug_read looks at (the last block's vector output)
    int upd_save = 0;            // VSSR_UPD_SAVE=1: update_fwd stores its intermediates and the reverse kernel loads them instead of
                                 // recomputing (measured: update_bwd 2.45 -> 2.21, update_fwd 1.30 -> 1.57..1.60 ms / step: no gain;
                                 // profiles/r03/NOTES_node_kernels.md)
    // results (device)
    vssr::DevBuf d_energy, d_energy_std, d_energy_models, d_forces, d_forces_std, d_e_atoms;
    vssr::DevBuf d_energy64;   // double [B] E | [B] sigma_E | [B][M] per model: the results before narrowing to float32
    vssr::DevBuf d_ters_e, d_ters_ea, d_ters_f;  // fp64 Tersoff results
    vssr::DevBuf d_sat, d_sat_out;   // [n_cfg] unsigned: saturation flags raised during a run / reported for the last evaluation of every chain
    std::vector<unsigned> h_sat;     // host copy of d_sat_out taken by vssr_batch_download (vssr_batch_saturated serves it: no second
    bool h_sat_valid = false;        // synchronisation / copy per download, advisor r3); void after every run
    vssr::DevBuf d_stress;           // [2][n_cfg][6] double: virial stress (mean, std over models) of the last evaluation (vssr_batch_stress)
};

namespace vssr {

int set_err(vssr_handle *h, int code, const char *fmt, ...);
#define VSSR_HIP(h, expr)                                                                    \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return vssr::set_err(h, VSSR_E_DEVICE, "%s failed: %s (%s:%d)", #expr,          \
                                 hipGetErrorString(_e), __FILE__, __LINE__);                 \
    } while (0)

// neighbor list (nbr.hip)
int build_neighbors(vssr_handle *h, double cutoff);
// PaiNN pipeline (painn.hip)
int painn_alloc_state(vssr_handle *h);
int painn_run(vssr_handle *h, uint32_t want);
int painn_stress(vssr_handle *h);   // enqueues k_stress: d_stress from the edge gradients of the last run (forces wanted)
// Tersoff (tersoff.hip)
int tersoff_run(vssr_handle *h, uint32_t want);
// EAM (eam.hip)
int eam_run(vssr_handle *h, uint32_t want);
void eam_build_spline(const double *f, int n, double delta, double *spl /*[n + 1][7]*/);
// lock-step FIRE relaxation (relax.hip)
// method 0: FIRE (fp), 1: BFGS (bp)
int relax_run(vssr_handle *h, int method, const vssr_fire_params *fp, const vssr_bfgs_params *bp,
              const uint8_t *fixed_host, uint32_t want);
// LAMMPS-style conjugate gradients for the fp64 potentials (relax.hip); results in d_relax_steps [B][3] = {iterations, evaluations, stop reason}
int relax_cg(vssr_handle *h, const vssr_cg_params *cp, const uint8_t *fixed_host, uint32_t want);
// chain_min.hip: the same minimisation with one workgroup per chain (Tersoff handles, chains of <= 256 atoms; VSSR_CG_FUSED=0 disables)
bool chain_min_supported(const vssr_handle *h);
int chain_min_cg(vssr_handle *h, const vssr_cg_params *cp, const uint8_t *fixed_host, uint32_t want);
// MFMA node stages (painn_node_mfma.hip)
int node_mfma_init(vssr_handle *h);
void launch_msg_mlp_mfma(hipStream_t st, int N, int M, int l, const ActiveView &av, const ModelW *MW, const float *s_in,
                         float *phi);
void launch_msg_mlp_bwd_mfma(hipStream_t st, int N, int M, int l, const ActiveView &av, const ModelW *MW, const float *s_in,
                             const float *phibar, const float *sbar_msg, float *sbar_in);
void launch_update_fwd_mfma(hipStream_t st, int N, int M, int l, const ActiveView &av, const ModelW *MW, const float *s_msg,
                            const float *v_msg, float *s_out, float *v_out, float *phi_next, void *save);
size_t update_save_bytes(int N, int M);   // per layer: forward intermediates of the update block kept for its reverse
bool readout_mfma_supported(int hidden);
void launch_readout_mfma(hipStream_t st, int N, int M, const ActiveView &av, const ModelW *MW, const float *s, const float *e_excl,
                         float *e_atom);
void launch_update_bwd_mfma(hipStream_t st, int N, int M, int l, int mode, int vbar_is_zero, const ActiveView &av, const ModelW *MW,
                            const float *s_msg, const float *v_msg, const float *sbar_src, const float *vbar,
                            const float *s_next, const float *phibar, const float *e_excl, float *e_atom,
                            float *sbar_msg, float *vbar_msg, const void *save);
// layer-0 species factorisation (painn_l0.hip)
void pack_mfma_tiles16(const float *Wsrc, int rows, int K, unsigned *dst /*[rows/32][K/16][2][64][4]*/);
void build_wd16(const float *Wd, const float *bd, unsigned *dst /*[3F][4][2][4]*/);
void l0_build_tables(const float *blob_embed, const float *W1, const float *b1, const float *W2, const float *b2,
                     const float *Wd, const float *bd, int n_embed, float *A, float *At);
void l0_pack_tables(const float *A, int n_embed, unsigned *A16, unsigned *At16);
size_t l0_packed_dwords(int n_embed);
int l0_mfma_init(vssr_handle *h);
int l0_run_forward(vssr_handle *h, const GraphView &G, float *s_msg, float *v_msg);
// fresh_mfma: chains of the matrix-pipe classes have nothing in the final buffer yet (overwrite), gather-class chains accumulate
int l0_run_reverse(vssr_handle *h, const GraphView &G, int first_write, int fresh_mfma, const float *sbar_msg, const float *vbar_msg,
                   float4 *gbar, long long gbar_stride, int n_groups);
// LDS-slice + MFMA edge stages (painn_edge_mfma.hip)
int edge_mfma_init(vssr_handle *h);
// forward neighbor-sum paths: 16-feature slices with / without the scalar residual in LDS, 8- and 4-feature slices, gather kernels
enum { EDGE_CLASS_FS16 = 0, EDGE_CLASS_FS16M = 1, EDGE_CLASS_FS8 = 2, EDGE_CLASS_FS4 = 3, EDGE_CLASS_GATHER = 4, EDGE_CLASSES = 5,
       EDGE_MFMA_CLASSES = 4 };
// reverse paths (the reverse tile is smaller: 16-feature slices serve chains up to 557 atoms): what GraphView::chain_class holds
// (FS16P: 16-feature slices in several passes over neighbor sub-ranges -- chains beyond the 557 atoms of the single-pass form, round 5)
enum { EDGE_BCLASS_FS16 = 0, EDGE_BCLASS_FS8 = 1, EDGE_BCLASS_FS4 = 2, EDGE_BCLASS_FS16P = 3, EDGE_BCLASS_GATHER = 4, EDGE_BCLASSES = 5,
       EDGE_MFMA_BCLASSES = 4 };
// partial edge-gradient buffers (feature slices) a chain of reverse class c writes per layer set and model
__host__ __device__ constexpr int edge_bclass_slices(int c) { return (c == EDGE_BCLASS_FS16 || c == EDGE_BCLASS_FS16P) ? 8 : c == EDGE_BCLASS_FS8 ? 16 : c == EDGE_BCLASS_FS4 ? 32 : 1; }
int edge_bclass_of(int n_atoms);
int edge_class_of(int n_atoms);       // path of a chain by its own atom count
// multi-pass forward neighbor sum (chains of the 4-feature class): the chain's atoms in P equal ranges of at most chunk_max atoms
constexpr int SUB_MAX_PASSES = 4;
__host__ __device__ constexpr int sub_chunk_max(int width) { return width == 16 ? 400 : 784; }   // staged rows that fit 160 KB (400 / 208 B per row) minus the zero row
__host__ __device__ constexpr int sub_chunk_max_bwd() { return 550; }   // reverse tile: 272 B per row + the current-centre records (12 KB) in 160 KB, minus the zero row
__host__ __device__ constexpr int sub_passes(int n_atoms, int chunk_max) { return (n_atoms + chunk_max - 1) / chunk_max; }
__host__ __device__ constexpr int sub_chunk(int n_atoms, int chunk_max) { return (n_atoms + sub_passes(n_atoms, chunk_max) - 1) / sub_passes(n_atoms, chunk_max); }
int edge_class_groups(int cls);       // partial edge-gradient buffers a chain of that class writes per model
void launch_edge_bwd_mfma(hipStream_t st, int cls, int N, const int *list, int n_list, int M, int l, int layer_first, int max_atoms,
                          const ModelW *MW, const GraphView &G, const int *counters, int zero_slot,
                          const float *v_in, const float *phi, const float *sbar_msg, const float *vbar_msg,
                          float *phibar, float *vbar_in, float *gbar, long long gbar_stride, int n_groups, int group_off, int rec,
                          const int4 *bundle_sub, int sub_chunk);
void launch_edge_fwd_mfma(hipStream_t st, int cls, int N, const int *list, int n_list, int M, int l, int max_atoms, const ModelW *MW,
                          const GraphView &G, const int *counters, int zero_slot, const float *s_in, const float *v_in,
                          const float *phi, float *s_msg, float *v_msg, const int4 *bundle_sub, int sub_width, int sub_chunk);

}  // namespace vssr
