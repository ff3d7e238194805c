// painn_l0.hip — exact species factorisation of the FIRST message block (forward and reverse).
//
// At layer 0 the message inputs depend on the neighbor only through its species: s0_j = Emb[Z_j], v0_j = 0, so
// phi0_j = MLP_0(Emb[Z_j]) =: phi0[Z_j] is a per-species constant.  With
// w_e = Wd_ext . rho_e the layer-0 message sums regroup EXACTLY (only the fp32 summation order changes):
//   s_msg_i[f]    = s0_i[f] + sum_z  sum_k  Ab_z[f][k] T_i,z[0][k]          Ab_z[f][k] = phi0_z[F+f]  Wd_ext[F+f][k]
//   v_msg_i[x][f] =           sum_z  sum_k  Ac_z[f][k] T_i,z[1+x][k]        Ac_z[f][k] = phi0_z[2F+f] Wd_ext[2F+f][k]
//   T_i,z[0][k] = sum_{e in i, Z_j = z} rho_e[k] ,  T_i,z[1+x][k] = sum_{e in i, Z_j = z} rho_e[k] u_e[x]
// i.e. a 4 x 24 block per (centre, neighbor species) instead of a 384-wide filter per edge.  Reverse pass, for the
// edge (n -> c) stored at slot (c, n), with z = Z_c:
//   Q_n,z[0][k]   = sum_f Ab_z[f][k] sbar_n[f] ,  Q_n,z[1+x][k] = sum_f Ac_z[f][k] vbar_n[x][f]
//   dE/dd = drho . Q[0] + sum_x u_nc[x] (drho . Q[1+x]) (+ excluded volume) ;  dE/du[x] = rho . Q[1+x]
// The per-edge work drops from 2 x 384 x 24 MACs per model (forward) to 7 x 24 (reverse) / 0 (forward): the layer-0
// launches of the neighbor-sum kernels disappear.  Ab / Ac depend only on the weights: built once at vssr_create.
// k runs in the table order kappa = kq * 6 + ks  <->  k = kq + 4 ks  used by the rho / drho tables (nbr.hip).
#include "vssr_internal.h"

namespace vssr {

constexpr int KP = 24;              // padded radial index
constexpr int TBLK = 4 * KP;        // floats per (atom, species) block: [scalar, x, y, z][24]

// ---- T_i,z : one block of 96 threads per centre; thread = (component, kappa) ------------------------------------------
__global__ void __launch_bounds__(96)
k_l0_accum(int nz, GraphView G, const int *__restrict__ counters, const int *__restrict__ Z,
           const int *__restrict__ zmap, float *__restrict__ T) {
    if (counters[2]) return;
    const int i = blockIdx.x, comp = threadIdx.x / KP, kap = threadIdx.x % KP;
    const int a0 = G.cfg_start[G.atom_cfg[i]];
    float acc[L0_MAX_SPECIES];
#pragma unroll
    for (int z = 0; z < L0_MAX_SPECIES; ++z) acc[z] = 0.f;
    for (int e = G.row_start[i]; e < G.row_start[i + 1]; ++e) {
        const float4 er = G.erec[e];
        const float r = G.rho[(size_t)e * KP + kap];          // pads: rho = 0
        const int zi = zmap[Z[a0 + __float_as_int(er.w)]];
        const float u = comp == 0 ? 1.f : comp == 1 ? er.x : comp == 2 ? er.y : er.z;
        const float val = r * u;
#pragma unroll
        for (int z = 0; z < L0_MAX_SPECIES; ++z) acc[z] += (z == zi) ? val : 0.f;
    }
    for (int z = 0; z < nz; ++z) T[((size_t)i * nz + z) * TBLK + comp * KP + kap] = acc[z];
}

constexpr int L0T = 8;   // atoms per block in the per-atom kernels (each table element is loaded once per L0T atoms)

// ---- forward: s_msg0, v_msg0 for every model; block = (tile of L0T atoms, model), thread = feature ---------------------
__global__ void __launch_bounds__(128)
k_l0_fwd(int N, int nz, const int *__restrict__ counters, const int *__restrict__ Z, const int *__restrict__ zlist,
         const ModelW *__restrict__ MW, const float *__restrict__ l0A /*[M][n_embed][2][24][F]*/, int n_embed,
         const float *__restrict__ T, float *__restrict__ s_msg, float *__restrict__ v_msg) {
    __shared__ float Ts[L0T][L0_MAX_SPECIES * TBLK];
    if (counters[2]) return;
    const int i0 = blockIdx.x * L0T, m = blockIdx.y, f = threadIdx.x;
    for (int t = f; t < L0T * nz * TBLK; t += 128) {
        const int a = t / (nz * TBLK), r = t % (nz * TBLK);
        Ts[a][r] = T[(size_t)min(i0 + a, N - 1) * nz * TBLK + r];
    }
    __syncthreads();
    float s[L0T], vx[L0T], vy[L0T], vz[L0T];
#pragma unroll
    for (int a = 0; a < L0T; ++a) {
        s[a] = MW[m].embed[(size_t)Z[min(i0 + a, N - 1)] * F + f];
        vx[a] = 0.f; vy[a] = 0.f; vz[a] = 0.f;
    }
    for (int z = 0; z < nz; ++z) {
        const float *Ab = l0A + (((size_t)m * n_embed + zlist[z]) * 2 + 0) * KP * F + f;
        const float *Ac = Ab + (size_t)KP * F;
        for (int k = 0; k < KP; ++k) {
            const float ab = Ab[k * F], ac = Ac[k * F];
#pragma unroll
            for (int a = 0; a < L0T; ++a) {
                const float *t0 = Ts[a] + z * TBLK;
                s[a] = fmaf(ab, t0[k], s[a]);
                vx[a] = fmaf(ac, t0[KP + k], vx[a]);
                vy[a] = fmaf(ac, t0[2 * KP + k], vy[a]);
                vz[a] = fmaf(ac, t0[3 * KP + k], vz[a]);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < L0T; ++a) {
        if (i0 + a >= N) break;
        const size_t g = (size_t)m * N + i0 + a;
        s_msg[g * F + f] = s[a];
        v_msg[(g * 3 + 0) * F + f] = vx[a];
        v_msg[(g * 3 + 1) * F + f] = vy[a];
        v_msg[(g * 3 + 2) * F + f] = vz[a];
    }
}

// ---- reverse, per atom: Q_n,z[comp][kappa] = sum_f A_z[f][kappa] X_comp[f], X = [sbar; vbar_x; vbar_y; vbar_z] --------------
__global__ void __launch_bounds__(96)
k_l0_q(int N, int nz, const int *__restrict__ counters, const int *__restrict__ zlist,
       const float *__restrict__ l0At /*[M][n_embed][2][F][24]*/, int n_embed, const float *__restrict__ sbar_msg,
       const float *__restrict__ vbar_msg, float *__restrict__ Q /*[M][N][nz][4][24]*/) {
    __shared__ float X[4][F][L0T];
    if (counters[2]) return;
    const int i0 = blockIdx.x * L0T, m = blockIdx.y, comp = threadIdx.x / KP, kap = threadIdx.x % KP;
    for (int t = threadIdx.x; t < L0T * 4 * F; t += 96) {
        const int a = t / (4 * F), c4 = (t / F) % 4, f = t % F;
        const size_t g = (size_t)m * N + min(i0 + a, N - 1);
        X[c4][f][a] = c4 == 0 ? sbar_msg[g * F + f] : vbar_msg[(g * 3 + (c4 - 1)) * F + f];
    }
    __syncthreads();
    for (int z = 0; z < nz; ++z) {
        const float *At = l0At + (((size_t)m * n_embed + zlist[z]) * 2 + (comp == 0 ? 0 : 1)) * F * KP + kap;
        float acc[L0T];
#pragma unroll
        for (int a = 0; a < L0T; ++a) acc[a] = 0.f;
#pragma unroll 4
        for (int f = 0; f < F; ++f) {
            const float w = At[f * KP];
#pragma unroll
            for (int a = 0; a < L0T; ++a) acc[a] = fmaf(w, X[comp][f][a], acc[a]);
        }
#pragma unroll
        for (int a = 0; a < L0T; ++a)
            if (i0 + a < N) Q[((((size_t)m * N + i0 + a) * nz + z) * 4 + comp) * KP + kap] = acc[a];
    }
}

// ---- reverse, per slot: dE/dr of the edge (n -> c) stored at slot (c, n); one wave per centre, lane = (slot, kq) ------------
__global__ void __launch_bounds__(256)
k_l0_bwd(int N, int M, int nz, int first_write, int excl_vol, GraphView G, const int *__restrict__ counters,
         const int *__restrict__ Z, const int *__restrict__ zmap, const float *__restrict__ Q,
         float4 *__restrict__ gbar, long long gbar_group_stride, int n_groups) {
    const int lane = threadIdx.x & 63, sl = lane >> 2, kq = lane & 3;
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= N || counters[2]) return;
    const int a0 = G.cfg_start[G.atom_cfg[c]];
    const int zc = zmap[Z[c]];
    const int e0 = G.row_start[c], e1 = G.row_start[c + 1];
    for (int eb = e0; eb < e1; eb += 16) {
        const int e = min(eb + sl, e1 - 1);
        const bool live = eb + sl < e1;
        const float4 er = G.erec[e];
        const float2 dd = G.dist2[e];
        const float *rp = G.rho + (size_t)e * KP + kq * 6, *dp = G.drho + (size_t)e * KP + kq * 6;
        float rho[6], drho[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { rho[k] = rp[k]; drho[k] = dp[k]; }
        const int n = a0 + __float_as_int(er.w);
        for (int m = 0; m < M; ++m) {
            const float *q = Q + ((((size_t)m * N + n) * nz + zc) * 4) * KP + kq * 6;
            float d0 = 0.f, dx = 0.f, dy = 0.f, dz = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float q0 = q[k], qx = q[KP + k], qy = q[2 * KP + k], qz = q[3 * KP + k];
                d0 = fmaf(drho[k], q0, d0);
                dx = fmaf(drho[k], qx, dx); dy = fmaf(drho[k], qy, dy); dz = fmaf(drho[k], qz, dz);
                bx = fmaf(rho[k], qx, bx); by = fmaf(rho[k], qy, by); bz = fmaf(rho[k], qz, bz);
            }
            // reduce the 4 radial quarters (lanes 4 sl .. 4 sl + 3): fixed order, DPP
            auto qs = [](float x) {
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
                return x;
            };
            d0 = qs(d0); dx = qs(dx); dy = qs(dy); dz = qs(dz); bx = qs(bx); by = qs(by); bz = qs(bz);
            if (live && kq == 0 && dd.x > 0.f) {
                const float ux = er.x, uy = er.y, uz = er.z;   // unit vector c -> n; the edge (n -> c) has -u
                float db = d0 - (dx * ux + dy * uy + dz * uz);
                if (excl_vol) db += dd.y;
                const float invd = dd.x;
                const float dotu = fmaf(bz, uz, fmaf(by, uy, bx * ux));
                float g0 = fmaf(-db, ux, (bx - dotu * ux) * invd);
                float g1 = fmaf(-db, uy, (by - dotu * uy) * invd);
                float g2 = fmaf(-db, uz, (bz - dotu * uz) * invd);
                float4 *gb = gbar + (size_t)m * n_groups * gbar_group_stride;   // group 0 of model m
                if (!first_write) {
                    const float4 old = gb[e];
                    g0 += old.x; g1 += old.y; g2 += old.z;
                }
                gb[e] = make_float4(g0, g1, g2, 0.f);
            }
        }
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------------
// Build Ab / Ac (both layouts) for every species index from the weights of one model (host, once per handle).
void l0_build_tables(const float *blob_embed, const float *W1, const float *b1, const float *W2, const float *b2,
                     const float *Wd, const float *bd, int n_embed, float *A /*[n_embed][2][24][F]*/,
                     float *At /*[n_embed][2][F][24]*/) {
    std::vector<float> h(F), phi(F3);
    for (int z = 0; z < n_embed; ++z) {
        const float *s = blob_embed + (size_t)z * F;
        for (int o = 0; o < F; ++o) {
            float acc = b1[o];
            for (int k = 0; k < F; ++k) acc = fmaf(W1[(size_t)o * F + k], s[k], acc);
            h[o] = acc / (1.f + expf(-acc));
        }
        for (int o = 0; o < F3; ++o) {
            float acc = b2[o];
            for (int k = 0; k < F; ++k) acc = fmaf(W2[(size_t)o * F + k], h[k], acc);
            phi[o] = acc;
        }
        for (int sec = 0; sec < 2; ++sec)           // 0: b-section (scalar message), 1: c-section (vector message)
            for (int kap = 0; kap < KP; ++kap) {
                // table order: kappa = kq * 6 + ks ; ks < 5 -> radial index k = kq + 4 ks ; ks = 5 -> envelope (bias
                // column), which the table replicates in every quarter: only quarter 0 carries the bias weight
                const int kq = kap / 6, ks = kap % 6;
                const int k = ks < 5 ? kq + 4 * ks : (kq == 0 ? 20 : 99);
                for (int f = 0; f < F; ++f) {
                    const int row = (sec + 1) * F + f;
                    const float w = k < 20 ? Wd[(size_t)row * 20 + k] : (k == 20 ? bd[row] : 0.f);
                    const float val = phi[row] * w;
                    A[(((size_t)z * 2 + sec) * KP + kap) * F + f] = val;
                    At[(((size_t)z * 2 + sec) * F + f) * KP + kap] = val;
                }
            }
    }
}

int l0_run_forward(vssr_handle *h, const GraphView &G, float *s_msg, float *v_msg) {
    const int N = h->n_atoms, M = h->n_models, nz = h->l0_nz;
    hipStream_t st = h->stream;
    if (h->d_l0T.ensure(sizeof(float) * (size_t)N * nz * TBLK) ||
        h->d_l0Q.ensure(sizeof(float) * (size_t)M * N * nz * TBLK))
        return set_err(h, VSSR_E_NOMEM, "layer-0 factorisation buffers: out of device memory");
    hipLaunchKernelGGL(k_l0_accum, dim3(N), dim3(96), 0, st, nz, G, h->d_counters.as<int>(), h->d_Z.as<int>(),
                       h->d_zmap.as<int>(), h->d_l0T.as<float>());
    hipLaunchKernelGGL(k_l0_fwd, dim3((N + L0T - 1) / L0T, M), dim3(128), 0, st, N, nz, h->d_counters.as<int>(), h->d_Z.as<int>(),
                       h->d_zlist.as<int>(), h->model_table.as<ModelW>(), h->d_l0A.as<float>(), h->n_embed,
                       h->d_l0T.as<float>(), s_msg, v_msg);
    return VSSR_OK;
}

int l0_run_reverse(vssr_handle *h, const GraphView &G, int first_write, const float *sbar_msg, const float *vbar_msg,
                   float4 *gbar, long long gbar_stride, int n_groups) {
    const int N = h->n_atoms, M = h->n_models, nz = h->l0_nz;
    hipStream_t st = h->stream;
    hipLaunchKernelGGL(k_l0_q, dim3((N + L0T - 1) / L0T, M), dim3(96), 0, st, N, nz, h->d_counters.as<int>(), h->d_zlist.as<int>(),
                       h->d_l0At.as<float>(), h->n_embed, sbar_msg, vbar_msg, h->d_l0Q.as<float>());
    hipLaunchKernelGGL(k_l0_bwd, dim3((N + 3) / 4), dim3(256), 0, st, N, M, nz, first_write, h->excl_vol, G,
                       h->d_counters.as<int>(), h->d_Z.as<int>(), h->d_zmap.as<int>(), h->d_l0Q.as<float>(), gbar,
                       gbar_stride, n_groups);
    return VSSR_OK;
}

}  // namespace vssr
