// painn.hip — PaiNN ensemble forward + hand-derived reverse pass on gfx950, batched over
// independent configurations (Markov chains) and ensemble members.
//
// Replaces `for model in models: model(batch)` + torch.autograd.grad inside
// EnsembleNFF.calculate (nff/io/ase_calcs.py; reference call site
// mcmc/calculators/calculators.py:484) — math per SURVEY.md Appendix A items 2-10:
//   message block  (nff/nn/modules/painn.py MessageBlock / InvariantMessage / DistanceEmbed)
//   update block   (nff/nn/modules/painn.py UpdateBlock)
//   readout + excluded volume + sum pool (nff/nn/models/painn.py)
// The reverse pass is written out by hand (the reference gets it from autograd):
// every edge quantity is accumulated by the workgroup that owns the CENTRE atom, so there are
// no float atomics anywhere and results are run-to-run deterministic.
//
// Data layout in HBM (fp32): s [M][N][F], v [M][N][3][F] (Cartesian-major so that a wave reads
// 64 consecutive features), phi [M][N][3F]; M = ensemble members, N = atoms of the whole batch.
// grid.y = ensemble member for every kernel.
#include "vssr_internal.h"

namespace vssr {

constexpr int T = NODE_TILE;   // atoms per workgroup of the readout kernel
constexpr int RB = 21;       // n_rbf (20) radial functions * envelope, + the envelope itself (bias column)
constexpr int ECHUNK = 16;   // edges staged per step in the edge kernels
constexpr float PI_F = 3.14159265358979323846f;

__device__ inline float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ inline float swishf_(float x) { return x * sigmoidf_(x); }
__device__ inline float dswishf_(float x) {
    float sg = sigmoidf_(x);
    return sg * fmaf(x, 1.f - sg, 1.f);
}

// ---- embedding: s0 = Emb[Z], v0 = 0 ------------------------------------------------------------
__global__ void __launch_bounds__(128) k_embed(int N, const int *__restrict__ Z, const ModelW *__restrict__ MW,
                                               float *__restrict__ s0, float *__restrict__ v0) {
    int i = blockIdx.x, m = blockIdx.y, f = threadIdx.x;
    const ModelW &W = MW[m];
    size_t a = (size_t)m * N + i;
    s0[a * F + f] = W.embed[(size_t)Z[i] * F + f];   // (cheap: runs for switched-off chains as well)
    v0[(a * 3 + 0) * F + f] = 0.f;
    v0[(a * 3 + 1) * F + f] = 0.f;
    v0[(a * 3 + 2) * F + f] = 0.f;
}

// ---- per-chunk edge geometry shared by the two edge kernels ----------------------------------------
struct EdgeChunk {
    float rho[ECHUNK][RB_MAX];
    float drho[ECHUNK][RB_MAX];
    float u[ECHUNK][4];  // unit vector centre -> neighbor, [3] = distance
    int j[ECHUNK];
    float fc[ECHUNK], dfc[ECHUNK];
};

template <bool DERIV>
__device__ inline void stage_chunk(EdgeChunk &S, const float4 *__restrict__ edge, int e0, int ne, float rc) {
    const int tid = threadIdx.x;
    if (tid < ECHUNK) {
        int j = -1;
        float d = 1.f, fc = 0.f, dfc = 0.f, ux = 0.f, uy = 0.f, uz = 0.f;
        if (tid < ne) {
            float4 ed = edge[e0 + tid];
            j = __float_as_int(ed.w);
            if (j >= 0) {
                d = sqrtf(fmaf(ed.z, ed.z, fmaf(ed.y, ed.y, ed.x * ed.x)));
                float inv = 1.f / d;
                ux = ed.x * inv; uy = ed.y * inv; uz = ed.z * inv;
                if (d < rc) {
                    fc = 0.5f * (cosf(PI_F * d / rc) + 1.f);
                    dfc = -0.5f * PI_F / rc * sinf(PI_F * d / rc);
                }
            }
        }
        S.j[tid] = j;
        S.u[tid][0] = ux; S.u[tid][1] = uy; S.u[tid][2] = uz; S.u[tid][3] = d;
        S.fc[tid] = fc; S.dfc[tid] = dfc;
        S.rho[tid][RB - 1] = fc;
        if (DERIV) S.drho[tid][RB - 1] = dfc;
    }
    __syncthreads();
    for (int p = tid; p < ECHUNK * (RB - 1); p += blockDim.x) {
        int e = p / (RB - 1), k = p % (RB - 1);
        float r = 0.f, dr = 0.f;
        if (S.j[e] >= 0) {
            float d = S.u[e][3], a = (float)(k + 1) * PI_F / rc;
            float sn, cs;
            sincosf(a * d, &sn, &cs);
            float rb = sn / d;
            r = rb * S.fc[e];
            if (DERIV) dr = fmaf(a * cs / d - sn / (d * d), S.fc[e], rb * S.dfc[e]);
        }
        S.rho[e][k] = r;
        if (DERIV) S.drho[e][k] = dr;
    }
    __syncthreads();
}

// ---- message block, forward: one workgroup per (centre atom, model); thread = feature --------------
// s_msg_i = s_i + sum_e phi_j[b] w_e[b];  v_msg_i = v_i + sum_e (phi_j[c] w_e[c] u_e + phi_j[a] w_e[a] v_j)
template <bool L0>
__global__ void __launch_bounds__(128)
k_edge_fwd(int N, int l, const ModelW *__restrict__ MW, GraphView G, const int *__restrict__ counters,
           float rc, int excl_vol, float excl_sigma, int excl_power, const float *__restrict__ s_in,
           const float *__restrict__ v_in, const float *__restrict__ phi, float *__restrict__ s_msg,
           float *__restrict__ v_msg, int only_class) {
    __shared__ EdgeChunk S;
    if (counters[2] || !G.act.atom(blockIdx.x)) return;
    if (only_class >= 0 && G.chain_class[G.atom_cfg[blockIdx.x]] != only_class) return;   // (chains of the matrix-pipe classes)
    const int i = blockIdx.x, m = blockIdx.y, f = threadIdx.x;
    const LayerW &W = MW[m].layer[l];
    float wa[RB], wb[RB], wc[RB];
    for (int k = 0; k < RB - 1; ++k) {
        wa[k] = L0 ? 0.f : W.Wd[(size_t)(f) * (RB - 1) + k];
        wb[k] = W.Wd[(size_t)(F + f) * (RB - 1) + k];
        wc[k] = W.Wd[(size_t)(2 * F + f) * (RB - 1) + k];
    }
    wa[RB - 1] = L0 ? 0.f : W.bd[f];
    wb[RB - 1] = W.bd[F + f];
    wc[RB - 1] = W.bd[2 * F + f];

    float acc_s = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    const int e_begin = G.row_start[i], e_end = G.row_start[i + 1];
    const size_t mN = (size_t)m * N;
    for (int e0 = e_begin; e0 < e_end; e0 += ECHUNK) {
        int ne = min(ECHUNK, e_end - e0);
        stage_chunk<false>(S, G.edge, e0, ne, rc);
        for (int e = 0; e < ne; ++e) {
            int j = S.j[e];
            if (j < 0) continue;
            float wA = 0.f, wB = 0.f, wC = 0.f;
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                float r = S.rho[e][k];
                if (!L0) wA = fmaf(wa[k], r, wA);
                wB = fmaf(wb[k], r, wB);
                wC = fmaf(wc[k], r, wC);
            }
            const float *pj = phi + (mN + j) * F3;
            acc_s = fmaf(pj[F + f], wB, acc_s);
            float mc = pj[2 * F + f] * wC;
            ax = fmaf(mc, S.u[e][0], ax);
            ay = fmaf(mc, S.u[e][1], ay);
            az = fmaf(mc, S.u[e][2], az);
            if (!L0) {
                float ma = pj[f] * wA;
                const float *vj = v_in + (mN + j) * 3 * F;
                ax = fmaf(ma, vj[f], ax);
                ay = fmaf(ma, vj[F + f], ay);
                az = fmaf(ma, vj[2 * F + f], az);
            }
        }
        __syncthreads();
    }
    size_t a = mN + i;
    s_msg[a * F + f] = s_in[a * F + f] + acc_s;
    if (L0) {
        v_msg[(a * 3 + 0) * F + f] = ax;
        v_msg[(a * 3 + 1) * F + f] = ay;
        v_msg[(a * 3 + 2) * F + f] = az;
    } else {
        v_msg[(a * 3 + 0) * F + f] = v_in[(a * 3 + 0) * F + f] + ax;
        v_msg[(a * 3 + 1) * F + f] = v_in[(a * 3 + 1) * F + f] + ay;
        v_msg[(a * 3 + 2) * F + f] = v_in[(a * 3 + 2) * F + f] + az;
    }
}

// ---- readout (+ its own reverse): e_i = w6.swish(W5 s + b5) + b6 ; sbar = W5^T (w6 * swish'(h5)) ---------
__global__ void __launch_bounds__(128)
k_readout(int N, int H, ActiveView av, const ModelW *__restrict__ MW, const float *__restrict__ s,
          const float *__restrict__ e_excl, int excl_vol, float *__restrict__ e_atom, float *__restrict__ sbar) {
    __shared__ float xs[F][T];
    __shared__ float hb[F][T];
    __shared__ float es[F][T];
    const int tid = threadIdx.x, m = blockIdx.y, a0 = blockIdx.x * T;
    if (!av.tile(min(a0, N - 1), min(a0 + T - 1, N - 1))) return;
    const ModelW &W = MW[m];
    const size_t mN = (size_t)m * N;
    for (int t = 0; t < T; ++t) {
        int atom = a0 + t;
        xs[tid][t] = atom < N ? s[(mN + atom) * F + tid] : 0.f;
    }
    __syncthreads();
    float acc[T];
    if (tid < H) {
        float b = W.b5[tid], w6 = W.w6[tid];
        for (int t = 0; t < T; ++t) acc[t] = b;
        for (int k = 0; k < F; ++k) {
            float w = W.W5t[k * H + tid];
            for (int t = 0; t < T; ++t) acc[t] = fmaf(w, xs[k][t], acc[t]);
        }
        for (int t = 0; t < T; ++t) {
            es[tid][t] = w6 * swishf_(acc[t]);
            hb[tid][t] = w6 * dswishf_(acc[t]);
        }
    }
    __syncthreads();
    if (tid < T) {
        int atom = a0 + tid;
        if (atom < N) {
            float e = W.b6[0];
            for (int o = 0; o < H; ++o) e += es[o][tid];
            if (excl_vol) e += e_excl[atom];
            e_atom[mN + atom] = e;
        }
    }
    for (int t = 0; t < T; ++t) acc[t] = 0.f;
    for (int o = 0; o < H; ++o) {
        float w = W.W5[o * F + tid];
        for (int t = 0; t < T; ++t) acc[t] = fmaf(w, hb[o][t], acc[t]);
    }
    for (int t = 0; t < T; ++t) {
        int atom = a0 + t;
        if (atom < N) sbar[(mN + atom) * F + tid] = acc[t];
    }
}

// ---- message block, reverse: one workgroup per (atom c, model), c in its role as SOURCE j ------------------
// For every neighbor n of c the edge (n -> c) carried phi_c, v_c into n.  Gathers the output
// adjoints of n, accumulates phibar_c and vbar_c without scatter, and produces dE/d r for the
// edge (n -> c), stored at the slot (c, n).
template <bool L0>
__global__ void __launch_bounds__(128)
k_edge_bwd(int N, int l, int accumulate, const ModelW *__restrict__ MW, GraphView G,
           const int *__restrict__ counters, float rc, int excl_vol, float excl_sigma, int excl_power,
           const float *__restrict__ v_in, const float *__restrict__ phi, const float *__restrict__ sbar_msg,
           const float *__restrict__ vbar_msg, float *__restrict__ phibar, float *__restrict__ vbar_in,
           float4 *__restrict__ gbar, long long gbar_stride, int only_class, int fresh_mfma) {
    __shared__ EdgeChunk S;
    __shared__ float red[ECHUNK][4][F + 1];
    __shared__ float tots[ECHUNK][4];
    if (counters[2] || !G.act.atom(blockIdx.x)) return;
    if (only_class >= 0 && G.chain_class[G.atom_cfg[blockIdx.x]] != only_class) return;
    // (compact partial buffers: the final buffer of a matrix-pipe chain holds nothing yet when layer 0 arrives)
    if (fresh_mfma && G.chain_class[G.atom_cfg[blockIdx.x]] != EDGE_BCLASS_GATHER) accumulate = 0;
    const int c = blockIdx.x, m = blockIdx.y, f = threadIdx.x;
    const LayerW &W = MW[m].layer[l];
    float wa[RB], wb[RB], wc[RB];
    for (int k = 0; k < RB - 1; ++k) {
        wa[k] = L0 ? 0.f : W.Wd[(size_t)(f) * (RB - 1) + k];
        wb[k] = W.Wd[(size_t)(F + f) * (RB - 1) + k];
        wc[k] = W.Wd[(size_t)(2 * F + f) * (RB - 1) + k];
    }
    wa[RB - 1] = L0 ? 0.f : W.bd[f];
    wb[RB - 1] = W.bd[F + f];
    wc[RB - 1] = W.bd[2 * F + f];

    const size_t mN = (size_t)m * N;
    const size_t ac = mN + c;
    const float pc_a = L0 ? 0.f : phi[ac * F3 + f];
    const float pc_b = phi[ac * F3 + F + f];
    const float pc_c = phi[ac * F3 + 2 * F + f];
    float vc0 = 0.f, vc1 = 0.f, vc2 = 0.f;
    if (!L0) {
        vc0 = v_in[(ac * 3 + 0) * F + f];
        vc1 = v_in[(ac * 3 + 1) * F + f];
        vc2 = v_in[(ac * 3 + 2) * F + f];
    }
    float accb = 0.f, accc = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    float4 *gb = gbar + (size_t)m * gbar_stride;
    const int e_begin = G.row_start[c], e_end = G.row_start[c + 1];
    for (int e0 = e_begin; e0 < e_end; e0 += ECHUNK) {
        int ne = min(ECHUNK, e_end - e0);
        stage_chunk<true>(S, G.edge, e0, ne, rc);
        for (int e = 0; e < ne; ++e) {
            int n = S.j[e];
            if (n < 0) {
                red[e][0][f] = 0.f; red[e][1][f] = 0.f; red[e][2][f] = 0.f; red[e][3][f] = 0.f;
                continue;
            }
            float wA = 0.f, wB = 0.f, wC = 0.f, dA = 0.f, dB = 0.f, dC = 0.f;
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                float r = S.rho[e][k], dr = S.drho[e][k];
                if (!L0) { wA = fmaf(wa[k], r, wA); dA = fmaf(wa[k], dr, dA); }
                wB = fmaf(wb[k], r, wB); dB = fmaf(wb[k], dr, dB);
                wC = fmaf(wc[k], r, wC); dC = fmaf(wc[k], dr, dC);
            }
            const size_t an = mN + n;
            float sbn = sbar_msg[an * F + f];
            float vb0 = vbar_msg[(an * 3 + 0) * F + f];
            float vb1 = vbar_msg[(an * 3 + 1) * F + f];
            float vb2 = vbar_msg[(an * 3 + 2) * F + f];
            // unit vector of the edge (n -> c) = -(c -> n)
            float u0 = -S.u[e][0], u1 = -S.u[e][1], u2 = -S.u[e][2];
            float p = fmaf(vb2, u2, fmaf(vb1, u1, vb0 * u0));
            accb = fmaf(wB, sbn, accb);
            accc = fmaf(wC, p, accc);
            float dpart = fmaf(pc_b * sbn, dB, pc_c * p * dC);
            if (!L0) {
                float q = fmaf(vb2, vc2, fmaf(vb1, vc1, vb0 * vc0));
                ax = fmaf(wA, vb0, ax);
                ay = fmaf(wA, vb1, ay);
                az = fmaf(wA, vb2, az);
                dpart = fmaf(pc_a * q, dA, dpart);
            }
            float mc = pc_c * wC;
            red[e][0][f] = dpart;
            red[e][1][f] = mc * vb0;
            red[e][2][f] = mc * vb1;
            red[e][3][f] = mc * vb2;
        }
        __syncthreads();
        if (f < ne * 4) {
            int e = f >> 2, comp = f & 3;
            float tot = 0.f;
            for (int k = 0; k < F; ++k) tot += red[e][comp][k];
            tots[e][comp] = tot;
        }
        __syncthreads();
        if (f < ne && S.j[f] >= 0) {
            float d = S.u[f][3];
            float u0 = -S.u[f][0], u1 = -S.u[f][1], u2 = -S.u[f][2];
            float db = tots[f][0];
            if (L0 && excl_vol) db -= (float)excl_power * powf(excl_sigma / d, (float)excl_power) / d;
            float b0 = tots[f][1], b1 = tots[f][2], b2 = tots[f][3];
            float dot = fmaf(b2, u2, fmaf(b1, u1, b0 * u0));
            float g0 = fmaf(db, u0, (b0 - dot * u0) / d);
            float g1 = fmaf(db, u1, (b1 - dot * u1) / d);
            float g2 = fmaf(db, u2, (b2 - dot * u2) / d);
            if (accumulate) {
                float4 old = gb[e0 + f];
                g0 += old.x; g1 += old.y; g2 += old.z;
            }
            gb[e0 + f] = make_float4(g0, g1, g2, 0.f);
        }
        __syncthreads();
    }
    if (!L0) {
        phibar[ac * F3 + f] = fmaf(vc2, az, fmaf(vc1, ay, vc0 * ax));
        phibar[ac * F3 + F + f] = accb;
        phibar[ac * F3 + 2 * F + f] = accc;
        vbar_in[(ac * 3 + 0) * F + f] = fmaf(pc_a, ax, vbar_msg[(ac * 3 + 0) * F + f]);
        vbar_in[(ac * 3 + 1) * F + f] = fmaf(pc_a, ay, vbar_msg[(ac * 3 + 1) * F + f]);
        vbar_in[(ac * 3 + 2) * F + f] = fmaf(pc_a, az, vbar_msg[(ac * 3 + 2) * F + f]);
    }
}

// ---- ensemble reduction -----------------------------------------------------------------------------------------
// forces: dE/dx_c = sum_{slots (c,n)} ( G[(n->c)] - G[(c->n)] ), G[(c->n)] lives at rev[slot].
// Partial edge gradients of the feature slices (one buffer per workgroup group) -> one buffer per model (group 0, in place): a streaming,
// coalesced pass in fixed group order, so that the gather through `rev` below touches one buffer per model only.  A chain's
// slots carry as many partial buffers as its class has slices (grid.x = chain).
// Compact partial buffers (gbar_mode 2): P [M][n_groups][slot_cap][3] floats, one set of `per_set` groups per reverse layer;
// a chain's slots carry `slices` groups per set.  final G [M][slot_cap] float4 already holds the layer-0 part.
__global__ void __launch_bounds__(256)
k_reduce_gpart(int M, int n_groups, int layer_sets, GraphView G, const int *__restrict__ counters, const float *__restrict__ P,
               float4 *__restrict__ gbar, long long slot_cap, int uniform_slices) {
    if (counters[2]) return;
    const int per_set = n_groups / layer_sets;
    long long s0, s1;
    int slices;
    if (uniform_slices) {   // every chain in one class: grid.x walks the slots of the whole batch
        s0 = 0; s1 = counters[0]; slices = uniform_slices;
    } else {                // grid (chains, blocks per chain)
        const int b = blockIdx.x;
        if (!G.act.chain(b)) return;
        slices = edge_bclass_slices(G.chain_class[b]);
        if (slices == 1) return;   // gather-class chains accumulate in the final buffer directly
        s0 = G.row_start[G.cfg_start[b]]; s1 = G.row_start[G.cfg_start[b + 1]];
    }
    const long long first = uniform_slices ? (long long)blockIdx.x * blockDim.x : (long long)blockIdx.y * blockDim.x;
    const long long step = uniform_slices ? (long long)gridDim.x * blockDim.x : (long long)gridDim.y * blockDim.x;
    for (long long slot = s0 + first + threadIdx.x; slot < s1; slot += step)
        for (int m = 0; m < M; ++m) {
            float4 acc = gbar[(size_t)m * slot_cap + slot];
            for (int k = 0; k < layer_sets; ++k)
                for (int grp = 0; grp < slices; ++grp) {
                    const float *p = P + ((size_t)(m * n_groups + k * per_set + grp) * slot_cap + slot) * 3;
                    acc.x += p[0]; acc.y += p[1]; acc.z += p[2];
                }
            gbar[(size_t)m * slot_cap + slot] = acc;
        }
}

// (every chain of the batch in one class -- the usual case: one thread per slot of the whole batch)
__global__ void k_reduce_gbar_groups_uniform(int M, int n_groups, const int *__restrict__ counters, float4 *__restrict__ gbar,
                                             long long gbar_stride) {
    const long long slot = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (counters[2] || slot >= counters[0]) return;
    for (int m = 0; m < M; ++m) {
        float4 *g0 = gbar + (size_t)(m * n_groups) * gbar_stride + slot;
        float4 acc = *g0;
        for (int grp = 1; grp < n_groups; ++grp) {
            const float4 v = g0[(size_t)grp * gbar_stride];
            acc.x += v.x; acc.y += v.y; acc.z += v.z;
        }
        *g0 = acc;
    }
}

__global__ void __launch_bounds__(256)
k_reduce_gbar_groups(int M, int n_groups, int layer_sets, GraphView G, const int *__restrict__ counters, float4 *__restrict__ gbar,
                     long long gbar_stride) {
    if (counters[2]) return;
    const int b = blockIdx.x;   // (x: no 65 535 limit on the chain count)
    if (!G.act.chain(b)) return;
    const int slices = edge_bclass_slices(G.chain_class[b]);   // buffers the chain's reverse kernels wrote per layer set
    if (slices == 1) return;
    const int per_set = n_groups / layer_sets;   // group index of layer set k, slice f: k * per_set + f
    const int s0 = G.row_start[G.cfg_start[b]], s1 = G.row_start[G.cfg_start[b + 1]];
    for (int slot = s0 + blockIdx.y * blockDim.x + threadIdx.x; slot < s1; slot += gridDim.y * blockDim.x)
        for (int m = 0; m < M; ++m) {
            float4 *g0 = gbar + (size_t)(m * n_groups) * gbar_stride + slot;
            float4 acc = *g0;
            for (int k = 0; k < layer_sets; ++k)
                for (int grp = k ? 0 : 1; grp < slices; ++grp) {
                    const float4 v = g0[(size_t)(k * per_set + grp) * gbar_stride];
                    acc.x += v.x; acc.y += v.y; acc.z += v.z;
                }
            *g0 = acc;
        }
}

// forces = - sum_slots (G[slot] - G[rev[slot]]) per model, then ensemble mean / std.  gbar_model_stride: distance between
// the (group-reduced) buffers of consecutive models.
// One wave per centre, lanes over its slots: the row's gradients are read as contiguous float4 (thread-per-atom reads were
// one L1 access per lane), the reverse slots are gathered, and the 3 x M partial sums are combined by a fixed shuffle tree.
__global__ void __launch_bounds__(256)
k_finalize_forces(int N, int M, GraphView G, const int *__restrict__ counters,
                  const float4 *__restrict__ gbar, long long gbar_model_stride, double units_per_ev,
                  float *__restrict__ forces, float *__restrict__ forces_std) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= N || counters[2] || !G.act.atom(c)) return;
    const int e0 = G.row_start[c], e1 = G.row_start[c + 1];
    double fm[3] = {0, 0, 0}, f2[3] = {0, 0, 0};
    for (int m = 0; m < M; ++m) {
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        const float4 *gb = gbar + (size_t)m * gbar_model_stride;
        for (int e = e0 + lane; e < e1; e += 64) {
            const int r = G.rev[e];
            if (r < 0) continue;
            const float4 a = gb[e], b = gb[r];
            g0 += a.x - b.x; g1 += a.y - b.y; g2 += a.z - b.z;
        }
        g0 = wave_sum_f32(g0); g1 = wave_sum_f32(g1); g2 = wave_sum_f32(g2);   // DPP + row swaps (vssr_internal.h), fixed order
        double f[3] = {-(double)g0 / units_per_ev, -(double)g1 / units_per_ev, -(double)g2 / units_per_ev};
        for (int x = 0; x < 3; ++x) { fm[x] += f[x]; f2[x] += f[x] * f[x]; }
    }
    if (lane < 3) {
        const int x = lane;
        const double fmx = x == 0 ? fm[0] : x == 1 ? fm[1] : fm[2], f2x = x == 0 ? f2[0] : x == 1 ? f2[1] : f2[2];
        double mu = fmx / M;
        double var = f2x / M - mu * mu;
        forces[3 * c + x] = (float)mu;
        if (forces_std) forces_std[3 * c + x] = (float)sqrt(var > 0 ? var : 0.0);
    }
}

// Virial stress of every chain from the per-slot edge gradients the reverse pass left in `gbar` (nothing is re-evaluated):
// the energy depends on positions and cell only through the edge vectors r_e = x_j + S cell - x_i, and gbar[e] holds
// dE/d(-r_e) (the gradient for the edge j -> i, stored at slot (i, j)), so under a homogeneous strain r -> (1 + eps) r
//   dE/d eps_ab = - sum_e gbar[e]_a r_e,b ,      sigma = (1 / V) dE/d eps  (ASE sign convention: tensile positive),
// symmetrised, Voigt order xx yy zz yz xz xy, eV / A^3; ensemble mean and population std over the models (like the forces).
// The stoichiometric offset does not depend on the strain.  One workgroup per chain, fp64 accumulation, fixed order.
__global__ void __launch_bounds__(256)
k_stress(int M, GraphView G, const int *__restrict__ counters, const float4 *__restrict__ gbar, long long gbar_model_stride,
         const double *__restrict__ cell, double units_per_ev, double *__restrict__ stress, double *__restrict__ stress_std) {
    __shared__ double red[9][256];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (counters[2]) return;
    const int s0 = G.row_start[G.cfg_start[b]], s1 = G.row_start[G.cfg_start[b + 1]];
    const double *c = cell + 9 * (size_t)b;
    const double vol = fabs(c[0] * (c[4] * c[8] - c[5] * c[7]) - c[1] * (c[3] * c[8] - c[5] * c[6]) + c[2] * (c[3] * c[7] - c[4] * c[6]));
    double mean[6] = {0, 0, 0, 0, 0, 0}, sq[6] = {0, 0, 0, 0, 0, 0};
    for (int m = 0; m < M; ++m) {
        const float4 *gb = gbar + (size_t)m * gbar_model_stride;
        double w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int e = s0 + tid; e < s1; e += 256) {
            const float4 r = G.edge[e];
            if (__float_as_int(r.w) < 0) continue;   // pad slot
            const float4 g = gb[e];
            const double gx = g.x, gy = g.y, gz = g.z, rx = r.x, ry = r.y, rz = r.z;
            w[0] -= gx * rx; w[1] -= gx * ry; w[2] -= gx * rz;
            w[3] -= gy * rx; w[4] -= gy * ry; w[5] -= gy * rz;
            w[6] -= gz * rx; w[7] -= gz * ry; w[8] -= gz * rz;
        }
        for (int k = 0; k < 9; ++k) red[k][tid] = w[k];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s)
                for (int k = 0; k < 9; ++k) red[k][tid] += red[k][tid + s];
            __syncthreads();
        }
        if (tid == 0) {
            const double f = 1.0 / (units_per_ev * vol);
            const double v[6] = {red[0][0] * f, red[4][0] * f, red[8][0] * f, 0.5 * (red[5][0] + red[7][0]) * f,
                                 0.5 * (red[2][0] + red[6][0]) * f, 0.5 * (red[1][0] + red[3][0]) * f};
            for (int k = 0; k < 6; ++k) { mean[k] += v[k]; sq[k] += v[k] * v[k]; }
        }
        __syncthreads();
    }
    if (tid == 0)
        for (int k = 0; k < 6; ++k) {
            const double mu = mean[k] / M, var = sq[k] / M - mu * mu;
            stress[6 * (size_t)b + k] = mu;
            if (stress_std) stress_std[6 * (size_t)b + k] = sqrt(var > 0 ? var : 0.0);
        }
}

__global__ void __launch_bounds__(256)
k_finalize_energy(int N, int M, const unsigned char *__restrict__ active, const int *__restrict__ cfg_start, const int *__restrict__ Z,
                  const float *__restrict__ e_atom, double units_per_ev, const double *__restrict__ offset_per_z,
                  double offset_const, float *__restrict__ energy, float *__restrict__ energy_std,
                  float *__restrict__ energy_models, double *__restrict__ energy64, float *__restrict__ e_atoms_mean,
                  unsigned *__restrict__ sat, unsigned *__restrict__ sat_out) {
    __shared__ double red[256];
    __shared__ double em[MAX_MODELS];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (active && !active[b]) return;
    const int a0 = cfg_start[b], a1 = cfg_start[b + 1];
    double off = 0.0;
    if (offset_per_z) {
        for (int i = a0 + tid; i < a1; i += blockDim.x) off += offset_per_z[Z[i]];
    }
    red[tid] = off;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    double offset = offset_per_z ? red[0] + offset_const : 0.0;
    __syncthreads();
    for (int m = 0; m < M; ++m) {
        double acc = 0.0;
        for (int i = a0 + tid; i < a1; i += blockDim.x) acc += (double)e_atom[(size_t)m * N + i];
        red[tid] = acc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) em[m] = red[0] / units_per_ev + offset;
        __syncthreads();
    }
    if (tid == 0) {
        double mu = 0, var = 0;
        for (int m = 0; m < M; ++m) mu += em[m];
        mu /= M;
        for (int m = 0; m < M; ++m) var += (em[m] - mu) * (em[m] - mu);
        energy[b] = (float)mu;
        energy_std[b] = (float)sqrt(var / M);
        // the same values before the narrowing to the reference's float32 result word (vssr_batch_energy_f64): [E | sigma | models]
        const int B = gridDim.x;
        energy64[b] = mu;
        energy64[B + b] = sqrt(var / M);
        for (int m = 0; m < M; ++m) energy64[(size_t)2 * B + (size_t)b * M + m] = em[m];
        // saturation report of this evaluation (mfma16.h SatTrack): the node kernels' flag, or a non-finite energy; the run
        // flag is cleared for the chain's next evaluation
        sat_out[b] = (sat[b] != 0u || !isfinite((float)mu)) ? 1u : 0u;
        sat[b] = 0u;
        for (int m = 0; m < M; ++m) energy_models[(size_t)b * M + m] = (float)em[m];
    }
    for (int i = a0 + tid; i < a1; i += blockDim.x) {
        double acc = 0;
        for (int m = 0; m < M; ++m) acc += (double)e_atom[(size_t)m * N + i];
        e_atoms_mean[i] = (float)(acc / M / units_per_ev);
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
// partial edge-gradient buffers per model: as many as the widest class present writes (one without the reverse edge kernels)
static int gbar_layer_sets(const vssr_handle *h) { return h->gbar_mode == 2 ? h->num_conv - 1 : h->gbar_mode == 1 && h->num_conv > 2 ? h->num_conv - 1 : 1; }
int painn_gbar_groups(const vssr_handle *h) {
    if (h->num_conv < 2) return 1;
    const int per_set = h->n_bclass[EDGE_BCLASS_FS4] ? edge_class_groups(EDGE_BCLASS_FS4)
                        : h->n_bclass[EDGE_BCLASS_FS8] ? edge_class_groups(EDGE_BCLASS_FS8)
                        : (h->n_bclass[EDGE_BCLASS_FS16] || h->n_bclass[EDGE_BCLASS_FS16P]) ? edge_class_groups(EDGE_BCLASS_FS16) : 1;
    return per_set > 1 ? per_set * gbar_layer_sets(h) : 1;
}

int painn_alloc_state(vssr_handle *h) {
    const size_t N = h->n_atoms, M = h->n_models, L = h->num_conv;
    const size_t nS = M * N * F, nV = 3 * nS, nP = 3 * nS;
    size_t floats = (L + 1) * (nS + nV) + L * nP + L * (nS + nV) + M * N  // forward
                    + 2 * (nS + nV) + nP;                                          // reverse
    if (h->d_state.ensure(floats * sizeof(float)))
        return set_err(h, VSSR_E_NOMEM, "activation arena (%zu MB): out of device memory", floats * 4 >> 20);
    float *p = h->d_state.as<float>();
    StateView &sv = h->sv;
    sv.n_atoms = (int)N;
    sv.n_models = (int)M;
    for (size_t l = 0; l <= L; ++l) { sv.s_in[l] = p; p += nS; sv.v_in[l] = p; p += nV; }
    for (size_t l = 0; l < L; ++l) {
        sv.phi[l] = p; p += nP;
        sv.s_msg[l] = p; p += nS;
        sv.v_msg[l] = p; p += nV;
    }
    sv.e_atom = p; p += M * N;
    sv.sbar = p; p += nS;
    sv.vbar = p; p += nV;
    sv.sbar_msg = p; p += nS;
    sv.vbar_msg = p; p += nV;
    sv.phibar = p; p += nP;
    if (h->upd_save && readout_mfma_supported(h->readout_hidden) &&
        h->d_upd_save.ensure(update_save_bytes((int)N, (int)M) * L))
        return set_err(h, VSSR_E_NOMEM, "forward intermediates of the update blocks: out of device memory");
    const size_t groups = (size_t)painn_gbar_groups(h);
    const bool compact = h->gbar_mode == 2 && groups > 1;
    if (h->d_gbar.ensure(sizeof(float4) * M * (compact ? 1 : groups) * (size_t)h->slot_cap) ||
        (compact && h->d_gpart.ensure(sizeof(float) * 3 * M * groups * (size_t)h->slot_cap + 256)))
        return set_err(h, VSSR_E_NOMEM, "edge-gradient buffer: out of device memory");
    sv.gbar = h->d_gbar.as<float4>();
    if (h->d_energy.ensure(sizeof(float) * h->n_cfg) || h->d_energy_std.ensure(sizeof(float) * h->n_cfg) ||
        h->d_energy_models.ensure(sizeof(float) * h->n_cfg * M) || h->d_energy64.ensure(sizeof(double) * h->n_cfg * (2 + M)) ||
        h->d_forces.ensure(sizeof(float) * 3 * N) ||
        h->d_forces_std.ensure(sizeof(float) * 3 * N) || h->d_e_atoms.ensure(sizeof(float) * N))
        return set_err(h, VSSR_E_NOMEM, "result buffers: out of device memory");
    if (h->d_sat.bytes < sizeof(unsigned) * h->n_cfg || h->d_sat_out.bytes < sizeof(unsigned) * h->n_cfg) {
        if (h->d_sat.ensure(sizeof(unsigned) * h->n_cfg) || h->d_sat_out.ensure(sizeof(unsigned) * h->n_cfg))
            return set_err(h, VSSR_E_NOMEM, "saturation flags: out of device memory");
        VSSR_HIP(h, hipMemsetAsync(h->d_sat.p, 0, h->d_sat.bytes, h->stream));
        VSSR_HIP(h, hipMemsetAsync(h->d_sat_out.p, 0, h->d_sat_out.bytes, h->stream));
    }
    return VSSR_OK;
}

int painn_run(vssr_handle *h, uint32_t want) {
    const int N = h->n_atoms, M = h->n_models, L = h->num_conv, H = h->readout_hidden;
    hipStream_t st = h->stream;
    h->h_sat_valid = false;   // (the host copy of the saturation report belongs to the previous evaluation)
    int rc = build_neighbors(h, (double)h->cutoff);
    if (rc) return rc;
    rc = painn_alloc_state(h);
    if (rc) return rc;
    StateView &sv = h->sv;
    sv.e_excl = h->d_excl.as<float>();
    GraphView G;
    G.n_atoms = N;
    G.n_cfg = h->n_cfg;
    G.atom_cfg = h->d_atom_cfg.as<int>();
    G.cfg_start = h->d_cfg_start.as<int>();
    G.row_start = h->d_row_start.as<int>();
    G.deg = h->d_deg.as<int>();
    G.edge = h->d_edge.as<float4>();
    G.rev = h->d_rev.as<int>();
    G.erec = h->d_erec.as<float4>();
    G.rho = h->d_rho.as<float>();
    G.dist2 = h->d_dist.as<float2>();
    G.rho16 = h->d_rho16.as<uint4>();
    G.drho16 = h->d_drho16.as<uint4>();
    G.zslot = h->d_zslot.as<unsigned char>();
    G.bundle = h->d_bundle.as<int4>();
    G.chain_class = h->d_chain_class.as<unsigned char>();
    G.act = ActiveView{h->active_mask, h->d_atom_cfg.as<int>(), h->d_sat.as<unsigned>()};
    const ActiveView &av = G.act;
    const ModelW *MW = h->model_table.as<ModelW>();
    const int *counters = h->d_counters.as<int>();
    const int *Z = h->d_Z.as<int>();
    dim3 blk(128);
    dim3 g_atom(N, M), g_tile((N + T - 1) / T, M);
    Profiler &P = h->prof;
    const bool l0_fact = h->l0_enabled && h->l0_nz > 0;   // layer 0 by species factorisation (any chain size)
    // The MFMA edge kernels serve layers >= 1, every chain through the instantiation of its own class (16- / 8-feature
    // slices; larger chains: the gather kernels) -- up to three launches per layer and direction, each over the chains of one
    // class.  Layer 0 is factorised by species (painn_l0.hip) or, with more than 8 species / VSSR_L0_FACTORISE=0, runs the
    // gather kernels for every chain (its v input is zero and only two filter sections matter).
    const int n_groups = painn_gbar_groups(h);   // partial gbar buffers per model
    const int layer_sets = n_groups > 1 ? gbar_layer_sets(h) : 1;
    const bool compact = h->gbar_mode == 2 && n_groups > 1;   // 12-byte partial records in d_gpart, final float4 buffer per model
    const int fin_groups = compact ? 1 : n_groups;             // buffers between two models in the final float4 array
    const int *cls_list[EDGE_MFMA_CLASSES];
    const int *bcls_list[EDGE_MFMA_BCLASSES];
    {
        int o = 0;
        for (int c = 0; c < EDGE_MFMA_CLASSES; o += h->n_class[c], ++c) cls_list[c] = h->d_class_list.as<int>() + o;
        for (int c = 0; c < EDGE_MFMA_BCLASSES; o += h->n_bclass[c], ++c) bcls_list[c] = h->d_class_list.as<int>() + o;
    }
    const int n_gather = h->n_class[EDGE_CLASS_GATHER];
    // forward intermediates of the update blocks for the reverse pass (fused reverse kernels and forces wanted only)
    const bool keep = h->upd_save && (want & VSSR_WANT_FORCES) && readout_mfma_supported(H);
    auto save_of = [&](int l) -> void * { return keep ? (char *)h->d_upd_save.p + update_save_bytes(N, M) * (size_t)l : nullptr; };
    h->l0_used = l0_fact;

    if (!l0_fact) {   // s0 = Emb[Z], v0 = 0 (the factorised layer 0 reads the embedding directly)
        P.begin(KC_EMBED, st);
        hipLaunchKernelGGL(k_embed, g_atom, blk, 0, st, N, Z, MW, sv.s_in[0], sv.v_in[0]);
        P.end(st);
    }
    for (int l = 0; l < L; ++l) {
        if (l == 0 && l0_fact) {   // phi0 is a per-species constant: no message MLP, no per-edge filter
            P.begin(KC_L0_FWD, st);
            rc = l0_run_forward(h, G, sv.s_msg[0], sv.v_msg[0]);
            if (rc) return rc;
            P.end(st);
            P.begin(KC_UPDATE_FWD, st);
            launch_update_fwd_mfma(st, N, M, l, av, MW, sv.s_msg[l], sv.v_msg[l], sv.s_in[l + 1],
                                   (l + 1 < L || h->debug_keep) ? sv.v_in[l + 1] : nullptr, l + 1 < L ? sv.phi[l + 1] : nullptr, save_of(l));
            P.end(st);
            continue;
        }
        if (l == 0) {   // (phi of layers >= 1 is the tail of the previous layer's update kernel)
            P.begin(KC_MSG_MLP, st);
            launch_msg_mlp_mfma(st, N, M, l, av, MW, sv.s_in[l], sv.phi[l]);
            P.end(st);
        }
        P.begin(KC_EDGE_FWD, st);
        if (l == 0)
            hipLaunchKernelGGL(k_edge_fwd<true>, g_atom, blk, 0, st, N, l, MW, G, counters, h->cutoff, h->excl_vol,
                               h->excl_sigma, h->excl_power, sv.s_in[l], sv.v_in[l], sv.phi[l], sv.s_msg[l],
                               sv.v_msg[l], -1);
        else {
            for (int cls = 0; cls < EDGE_MFMA_CLASSES; ++cls)
                launch_edge_fwd_mfma(st, cls, N, cls_list[cls], h->n_class[cls], M, l, h->max_class_atoms[cls], MW, G, counters,
                                     (int)(h->slot_cap - 1), sv.s_in[l], sv.v_in[l], sv.phi[l], sv.s_msg[l], sv.v_msg[l],
                                     (cls == EDGE_CLASS_FS4 && h->fwd_two_pass) ? h->d_bundle_sub.as<int4>() : (const int4 *)nullptr,
                                     h->fwd_two_pass, h->sub_chunk_fwd ? h->sub_chunk_fwd : sub_chunk_max(h->fwd_two_pass));
            if (n_gather)
                hipLaunchKernelGGL(k_edge_fwd<false>, g_atom, blk, 0, st, N, l, MW, G, counters, h->cutoff, h->excl_vol,
                                   h->excl_sigma, h->excl_power, sv.s_in[l], sv.v_in[l], sv.phi[l], sv.s_msg[l],
                                   sv.v_msg[l], (int)EDGE_BCLASS_GATHER);
        }
        P.end(st);
        P.begin(KC_UPDATE_FWD, st);
        // (nothing reads the vector output of the LAST block -- the readout takes s only: it is written for vssr_debug_read only)
        launch_update_fwd_mfma(st, N, M, l, av, MW, sv.s_msg[l], sv.v_msg[l], sv.s_in[l + 1],
                               (l + 1 < L || h->debug_keep) ? sv.v_in[l + 1] : nullptr, l + 1 < L ? sv.phi[l + 1] : nullptr, save_of(l));
        P.end(st);
    }
    // Readout.  With forces wanted (and the compiled readout width) it runs as the head of the last layer's reverse kernel;
    // energy-only evaluations use the stand-alone kernel built from the same code (identical per-atom energies).
    const bool fused = readout_mfma_supported(H);
    const bool want_forces = (want & VSSR_WANT_FORCES) != 0;
    const float *e_excl = h->excl_vol ? sv.e_excl : (const float *)nullptr;
    if (!(fused && want_forces)) {
        P.begin(KC_READOUT, st);
        if (fused) launch_readout_mfma(st, N, M, av, MW, sv.s_in[L], e_excl, sv.e_atom);
        else
            hipLaunchKernelGGL(k_readout, g_tile, blk, 0, st, N, H, av, MW, sv.s_in[L], sv.e_excl, h->excl_vol, sv.e_atom,
                               sv.sbar);
        P.end(st);
    }

    if (want_forces) {
        // Reverse pass.  update_bwd(l) carries the producer of its sbar input as its head: the readout (l = L - 1) or the
        // reverse of the message MLP of layer l + 1 (painn_node_mfma.hip); the adjoint of s_msg alternates between two
        // buffers so that a launch never reads the buffer it writes.
        float *sb_buf[2] = {sv.sbar_msg, sv.sbar};
        for (int l = L - 1; l >= 0; --l) {
            float *sbar_msg_l = fused ? sb_buf[(L - 1 - l) & 1] : sv.sbar_msg;
            const float *sbar_msg_up = sb_buf[(L - l) & 1];   // adjoint of s_msg[l + 1] (fused path, l < L - 1)
            P.begin(KC_UPDATE_BWD, st);
            if (!fused)
                launch_update_bwd_mfma(st, N, M, l, 0, (int)(l == L - 1), av, MW, sv.s_msg[l], sv.v_msg[l], sv.sbar, sv.vbar,
                                       nullptr, nullptr, nullptr, nullptr, sbar_msg_l, sv.vbar_msg, nullptr);
            else if (l == L - 1)
                launch_update_bwd_mfma(st, N, M, l, 1, 1, av, MW, sv.s_msg[l], sv.v_msg[l], nullptr, sv.vbar, sv.s_in[L], nullptr,
                                       e_excl, sv.e_atom, sbar_msg_l, sv.vbar_msg, save_of(l));
            else
                launch_update_bwd_mfma(st, N, M, l, 2, 0, av, MW, sv.s_msg[l], sv.v_msg[l], sbar_msg_up, sv.vbar, sv.s_in[l + 1],
                                       sv.phibar, nullptr, nullptr, sbar_msg_l, sv.vbar_msg, save_of(l));
            P.end(st);
            sv.sbar_msg_l0 = sbar_msg_l;
            P.begin((l == 0 && l0_fact) ? KC_L0_BWD : KC_EDGE_BWD, st);
            int accumulate = (l != L - 1);
            if (l == 0 && l0_fact) {
                rc = l0_run_reverse(h, G, (int)(L == 1), (int)compact, sbar_msg_l, sv.vbar_msg, sv.gbar, (long long)h->slot_cap,
                                    fin_groups);
                if (rc) return rc;
            } else if (l == 0)   // adds into partial buffer 0 of every model (model stride = n_groups buffers)
                hipLaunchKernelGGL(k_edge_bwd<true>, g_atom, blk, 0, st, N, l, accumulate, MW, G, counters,
                                   h->cutoff, h->excl_vol, h->excl_sigma, h->excl_power, sv.v_in[l], sv.phi[l],
                                   sbar_msg_l, sv.vbar_msg, sv.phibar, sv.vbar, sv.gbar,
                                   (long long)h->slot_cap * fin_groups, -1, (int)compact);
            else {
                // (a launch that writes a partial-gradient set for the first time writes compact records: k_edge_bwd_mfma<.., FIRST = true>
                //  has their stride compiled in)
                if ((l == L - 1 || layer_sets > 1) && !compact && h->n_cfg > n_gather)
                    return set_err(h, VSSR_E_STATE, "matrix-pipe reverse kernels need the compact partial-gradient records");
                for (int cls = 0; cls < EDGE_MFMA_BCLASSES; ++cls)
                    launch_edge_bwd_mfma(st, cls, N, bcls_list[cls], h->n_bclass[cls], M, l, (int)(l == L - 1 || layer_sets > 1),
                                         h->max_bclass_atoms[cls], MW, G, counters, (int)(h->slot_cap - 1), sv.v_in[l], sv.phi[l],
                                         sbar_msg_l, sv.vbar_msg, sv.phibar, sv.vbar,
                                         compact ? h->d_gpart.as<float>() : reinterpret_cast<float *>(sv.gbar), (long long)h->slot_cap,
                                         n_groups, layer_sets > 1 ? (L - 1 - l) * (n_groups / layer_sets) : 0, compact ? 3 : 4,
                                         cls == EDGE_BCLASS_FS16P ? h->d_bundle_subb.as<int4>() : (const int4 *)nullptr,
                                         h->sub_chunk_bwd ? h->sub_chunk_bwd : sub_chunk_max_bwd());
                if (n_gather)   // partial buffer 0 of every model
                    hipLaunchKernelGGL(k_edge_bwd<false>, g_atom, blk, 0, st, N, l, accumulate, MW, G, counters,
                                       h->cutoff, h->excl_vol, h->excl_sigma, h->excl_power, sv.v_in[l], sv.phi[l],
                                       sbar_msg_l, sv.vbar_msg, sv.phibar, sv.vbar, sv.gbar,
                                       (long long)h->slot_cap * fin_groups, (int)EDGE_BCLASS_GATHER, 0);
            }
            P.end(st);
            if (l > 0 && !fused) {
                P.begin(KC_MSG_MLP_BWD, st);
                launch_msg_mlp_bwd_mfma(st, N, M, l, av, MW, sv.s_in[l], sv.phibar, sv.sbar_msg, sv.sbar);
                P.end(st);
            }
        }
    }
    P.begin(KC_FINALIZE, st);
    if (want & VSSR_WANT_FORCES) {
#ifdef ABL_LDS_FORCE   // ablation build (profiles/r04/NOTES_force_accum.md): no per-slot partial records, nothing to reduce (results incomplete)
        if (compact) {   // (skipped)
        } else if (false) {
#else
        if (compact) {
#endif
            const int uni = h->active_mask ? 0 : h->n_bclass[EDGE_BCLASS_FS16] + h->n_bclass[EDGE_BCLASS_FS16P] == h->n_cfg ? 8 : h->n_bclass[EDGE_BCLASS_FS8] == h->n_cfg ? 16
                            : h->n_bclass[EDGE_BCLASS_FS4] == h->n_cfg ? 32 : 0;
            const dim3 grid = uni ? dim3((unsigned)((h->slot_cap + 255) / 256), 1) : dim3(h->n_cfg, 12);
            hipLaunchKernelGGL(k_reduce_gpart, grid, dim3(256), 0, st, M, n_groups, layer_sets, G, counters, h->d_gpart.as<float>(),
                               sv.gbar, (long long)h->slot_cap, uni);
        } else if (n_groups > 1) {
            const int cls_only = h->n_bclass[EDGE_BCLASS_FS16] + h->n_bclass[EDGE_BCLASS_FS16P] == h->n_cfg ? EDGE_BCLASS_FS16
                                 : h->n_bclass[EDGE_BCLASS_FS8] == h->n_cfg ? EDGE_BCLASS_FS8
                                 : h->n_bclass[EDGE_BCLASS_FS4] == h->n_cfg ? EDGE_BCLASS_FS4 : -1;
            if (cls_only >= 0 && !h->active_mask && (layer_sets == 1 || n_gather == 0))   // (switched-off chains keep their reduced gradients: the per-chain form skips them)
                hipLaunchKernelGGL(k_reduce_gbar_groups_uniform, dim3((unsigned)((h->slot_cap + 255) / 256)), dim3(256), 0, st, M,
                                   n_groups, counters, sv.gbar, (long long)h->slot_cap);
            else
                hipLaunchKernelGGL(k_reduce_gbar_groups, dim3(h->n_cfg, 12), dim3(256), 0, st, M, n_groups, layer_sets, G, counters, sv.gbar,
                                   (long long)h->slot_cap);
        }
        hipLaunchKernelGGL(k_finalize_forces, dim3((N + 3) / 4), dim3(256), 0, st, N, M, G, counters, sv.gbar,
                           (long long)h->slot_cap * fin_groups, h->units_per_ev, h->d_forces.as<float>(),
                           h->d_forces_std.as<float>());
    }
    hipLaunchKernelGGL(k_finalize_energy, dim3(h->n_cfg), dim3(256), 0, st, N, M, h->active_mask, G.cfg_start, Z, sv.e_atom,
                       h->units_per_ev, h->has_offset ? h->offset_per_z.as<double>() : (const double *)nullptr,
                       h->offset_const, h->d_energy.as<float>(), h->d_energy_std.as<float>(),
                       h->d_energy_models.as<float>(), h->d_energy64.as<double>(), h->d_e_atoms.as<float>(),
                       h->d_sat.as<unsigned>(), h->d_sat_out.as<unsigned>());
    P.end(st);
    VSSR_HIP(h, hipGetLastError());
    return VSSR_OK;
}

// vssr_batch_stress: the virial of the last evaluation (forces must have been computed: gbar holds the reduced edge gradients)
int painn_stress(vssr_handle *h) {
    if (h->d_stress.ensure(sizeof(double) * 12 * (size_t)h->n_cfg)) return set_err(h, VSSR_E_DEVICE, "out of device memory (stress)");
    GraphView G{};
    G.n_atoms = h->n_atoms;
    G.n_cfg = h->n_cfg;
    G.cfg_start = h->d_cfg_start.as<int>();
    G.row_start = h->d_row_start.as<int>();
    G.edge = h->d_edge.as<float4>();
    const int fin_groups = h->gbar_mode == 2 ? 1 : painn_gbar_groups(h);
    double *out = h->d_stress.as<double>();
    hipLaunchKernelGGL(k_stress, dim3(h->n_cfg), dim3(256), 0, h->stream, h->n_models, G, h->d_counters.as<int>(), h->sv.gbar,
                       (long long)h->slot_cap * fin_groups, h->d_cell.as<double>(), h->units_per_ev, out, out + 6 * (size_t)h->n_cfg);
    VSSR_HIP(h, hipGetLastError());
    return VSSR_OK;
}

}  // namespace vssr
