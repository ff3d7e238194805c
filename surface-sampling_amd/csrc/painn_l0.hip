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
#include "mfma16.h"

namespace vssr {

constexpr int KP = 24;              // padded radial index
constexpr int TBLK = 4 * KP;        // floats per (atom, species) block: [scalar, x, y, z][24]

// ---- T_i,z : one block of 96 threads per centre; thread = (component, kappa) ------------------------------------------
template <int NZ>   // number of species of the batch (the per-slot select runs over exactly these)
__global__ void __launch_bounds__(96)
k_l0_accum(GraphView G, const int *__restrict__ counters, float *__restrict__ T) {
    constexpr int nz = NZ;
    if (counters[2] || !G.act.atom(blockIdx.x)) return;
    const int i = blockIdx.x, comp = threadIdx.x / KP, kap = threadIdx.x % KP;
    float acc[NZ];
#pragma unroll
    for (int z = 0; z < NZ; ++z) acc[z] = 0.f;
    // slot counts are multiples of 4: four slots per iteration, all loads independent (species index from the per-slot
    // table written by k_edge_geom); the sum order over slots stays ascending
    // the per-slot record and species index are the same for every thread of the block: read through the constant
    // address space (scalar loads) instead of 96 identical vector accesses -- the kernel is bound by the L1 access rate
    auto erec_c = [&](int idx) -> float4 {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef const f32x4 __attribute__((address_space(4))) *cp;
        const f32x4 v = reinterpret_cast<cp>(reinterpret_cast<uintptr_t>(G.erec))[idx];
        return make_float4(v.x, v.y, v.z, v.w);
#else
        return G.erec[idx];
#endif
    };
    auto zslot_c = [&](int word) -> unsigned {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef const unsigned __attribute__((address_space(4))) *cp;
        return reinterpret_cast<cp>(reinterpret_cast<uintptr_t>(G.zslot))[word];
#else
        return reinterpret_cast<const unsigned *>(G.zslot)[word];
#endif
    };
    const int e_begin = __builtin_amdgcn_readfirstlane(G.row_start[i]), e_end = __builtin_amdgcn_readfirstlane(G.row_start[i + 1]);
    for (int e = e_begin; e < e_end; e += 4) {   // (rows start at multiples of 4: the 4 species bytes are one aligned word)
        float4 er[4];
        float r[4];
        int zi[4];
        const unsigned zw = zslot_c(e >> 2);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            er[u] = erec_c(e + u);
            r[u] = G.rho[(size_t)(e + u) * KP + kap];          // pads: rho = 0
            zi[u] = (zw >> (8 * u)) & 255;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float uu = comp == 0 ? 1.f : comp == 1 ? er[u].x : comp == 2 ? er[u].y : er[u].z;
            const float val = r[u] * uu;
#pragma unroll
            for (int z = 0; z < NZ; ++z) acc[z] += (z == zi[u]) ? val : 0.f;
        }
    }
#pragma unroll
    for (int z = 0; z < NZ; ++z) T[((size_t)i * nz + z) * TBLK + comp * KP + kap] = acc[z];
}

// ---- per-atom contractions on the matrix pipe (fp16 2-way split, see painn_node_mfma.hip / mfma16.h) -------------------------
// Species z is one K-chunk of 32: its 24 table entries + 8 zeros.  Weight tiles per (model, species, section) were packed
// at vssr_create (l0_pack_tables): forward  B = A_z[f][kappa]  (8 column tiles of 16 features, 1 chunk),
//                                  reverse  B = A_z[kappa][f]  (2 column tiles of 16 kappas, 4 chunks of 32 features).
constexpr int L0_TILE_U4 = 1024;   // uint4 per (model, species, section) in either packed table

// forward: s_msg0 = Emb[Z] + T[:, 0] . Ab ,  v_msg0[x] = T[:, 1 + x] . Ac  for every model; workgroup = 32 atoms (the T
// planes are model independent and are reused for all M models), wave w = features 16 w .. 16 w + 15
__global__ void __launch_bounds__(NTHREADS)
k_l0_fwd16(int N, int M, int nz, ActiveView av, const int *__restrict__ counters, const int *__restrict__ Z,
           const int *__restrict__ zlist, const ModelW *__restrict__ MW, const uint4 *__restrict__ A16, int n_embed,
           const float *__restrict__ T, float *__restrict__ s_msg, float *__restrict__ v_msg) {
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
    if (counters[2]) return;
    const int K = 32 * nz, a0 = blockIdx.x * TA;
    if (!av.tile(min(a0, N - 1), min(a0 + TA - 1, N - 1))) return;
    const Planes Ts = make_planes(ldsh, TA, K), Tv = make_planes(ldsh + plane_halves(TA, K), 3 * TA, K);
    const LaneGeo L;
    SatTrack sat;
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    // T[(i nz + z) 96 + comp 24 + kappa] -> planes; the 8 pad entries of every species chunk are zero
    // (four entries per thread and pass: 16-byte loads, 8-byte plane stores; KP = 24 is a multiple of 4)
    for (int t = threadIdx.x; t < TA * nz * 4 * 8; t += NTHREADS) {
        const int k4 = t & 7, comp = (t >> 3) & 3, z = (t >> 5) % nz, a = t / (32 * nz);
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (4 * k4 < KP)
            val = *reinterpret_cast<const float4 *>(T + ((size_t)min(a0 + a, N - 1) * nz + z) * TBLK + comp * KP + 4 * k4);
        if (comp == 0) store_split4<SAT_ANY>(Ts, a, 32 * z + 4 * k4, val, sat);
        else store_split4<SAT_ANY>(Tv, (comp - 1) * TA + a, 32 * z + 4 * k4, val, sat);
    }
    sat.commit(av, a0, N);
    __syncthreads();
    // weight pieces of step (m, q + 1) are requested before the matrix instructions of step (m, q)
    u32x4 bn[2][2];
    auto load_w = [&](int mm, int qq) {
        const uint4 *wb = A16 + (((size_t)mm * n_embed + zlist[qq]) * 2) * L0_TILE_U4 + (size_t)L.w * 128 + lane;
#pragma unroll
        for (int sec = 0; sec < 2; ++sec)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) bn[sec][pc] = gload4u(wb + sec * L0_TILE_U4 + pc * 64);
    };
    load_w(0, 0);
    for (int m = 0; m < M; ++m) {
        f32x4 acc[8];   // tiles 0, 1: scalar rows ; 2 + 2 x + t: component x, rows 16 t ..
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int q = 0; q < nz; ++q) {
            u32x4 b[2][2], a[8][2];
#pragma unroll
            for (int sec = 0; sec < 2; ++sec)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) b[sec][pc] = bn[sec][pc];
            {
                const int qn = q + 1 < nz ? q + 1 : 0, mn = q + 1 < nz ? m : m + 1;
                if (mn < M) load_w(mn, qn);   // (workgroup-uniform)
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const Planes &P = t < 2 ? Ts : Tv;
                const int row = (t < 2 ? 16 * t : 16 * (t - 2)) + r;
                a[t][0] = *reinterpret_cast<const u32x4 *>(P.h + row * P.ld + 32 * q + 8 * g);
                a[t][1] = *reinterpret_cast<const u32x4 *>(P.l + row * P.ld + 32 * q + 8 * g);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    acc[t] = mfma16(b[t < 2 ? 0 : 1][k == 0 ? 1 : 0], a[t][k == 1 ? 1 : 0], acc[t]);   // w_l a_h, w_h a_l, w_h a_h: D[feature][atom] (LaneGeo)
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int a_ = a0 + L.row(t);
            if (a_ >= N) continue;
            const size_t gI = (size_t)m * N + a_;
            *reinterpret_cast<f32x4 *>(s_msg + gI * F + L.col0) = gload4f(MW[m].embed + (size_t)Z[a_] * F + L.col0) + acc[t];
#pragma unroll
            for (int x = 0; x < 3; ++x) *reinterpret_cast<f32x4 *>(v_msg + (gI * 3 + x) * F + L.col0) = acc[2 + 2 * x + t];
        }
    }
}

// reverse, per atom: Q_n,z[comp][kappa] = sum_f A_z[f][kappa] X_comp[f], X = [sbar; vbar_x; vbar_y; vbar_z].  Workgroup =
// (32 atoms, model); the 2 nz column tiles (species x kappa half) are spread over the 8 waves.
__global__ void __launch_bounds__(NTHREADS)
k_l0_q16(int N, int nz, ActiveView av, const int *__restrict__ counters, const int *__restrict__ zlist, const uint4 *__restrict__ At16,
         int n_embed, const float *__restrict__ sbar_msg, const float *__restrict__ vbar_msg,
         float *__restrict__ Q /*[M][N][nz][4][24]*/) {
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
    if (counters[2]) return;
    const int a0 = blockIdx.x * TA, m = blockIdx.y;
    if (!av.tile(min(a0, N - 1), min(a0 + TA - 1, N - 1))) return;
    const size_t mN = (size_t)m * N;
    const Planes Xs = make_planes(ldsh, TA, F), Xv = make_planes(ldsh + plane_halves(TA, F), 3 * TA, F);
    SatTrack sat;
    load_rows_split<TA>(Xs, 0, [&](int row) { return sbar_msg + (mN + min(a0 + row, N - 1)) * F; }, sat);
    load_rows_split<3 * TA>(Xv, 0, [&](int row) {
        int x = row / TA, a = min(a0 + (row % TA), N - 1);
        return vbar_msg + ((mN + a) * 3 + x) * F;
    }, sat);
    sat.commit(av, a0, N);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, w = threadIdx.x >> 6;
    for (int ct = w; ct < 2 * nz; ct += NW) {
        const int q = ct >> 1, hf = ct & 1;
        const uint4 *wb = At16 + (((size_t)m * n_embed + zlist[q]) * 2) * L0_TILE_U4 + (size_t)hf * 512 + lane;
        f32x4 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // weight pieces one chunk ahead (ping-pong registers; the loop is unrolled so that the halves are static)
        u32x4 bb[2][2][2];
        auto load_w = [&](int buf, int c) {
#pragma unroll
            for (int sec = 0; sec < 2; ++sec)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) bb[buf][sec][pc] = gload4u(wb + sec * L0_TILE_U4 + (c * 2 + pc) * 64);
        };
        load_w(0, 0);
#pragma unroll
        for (int c = 0; c < F / 32; ++c) {
            u32x4 (&b)[2][2] = bb[c & 1];
            u32x4 a[8][2];
            if (c + 1 < F / 32) load_w((c + 1) & 1, c + 1);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const Planes &P = t < 2 ? Xs : Xv;
                const int row = (t < 2 ? 16 * t : 16 * (t - 2)) + r;
                a[t][0] = *reinterpret_cast<const u32x4 *>(P.h + row * P.ld + 32 * c + 8 * g);
                a[t][1] = *reinterpret_cast<const u32x4 *>(P.l + row * P.ld + 32 * c + 8 * g);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    acc[t] = mfma16(a[t][k == 1 ? 1 : 0], b[t < 2 ? 0 : 1][k == 0 ? 1 : 0], acc[t]);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        const int kap = 16 * hf + r;
        if (kap < KP) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int comp = t < 2 ? 0 : 1 + (t - 2) / 2, tt = t < 2 ? t : (t - 2) & 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int a_ = a0 + 16 * tt + 4 * g + i;
                    if (a_ < N) Q[((((size_t)m * N + a_) * nz + q) * 4 + comp) * KP + kap] = acc[t][i];
                }
            }
        }
    }
}

// ---- reverse, per slot: dE/dr of the edge (n -> c) stored at slot (c, n); one wave per centre, lane = (slot, kq) ------------
// Walks the row of atom n and produces the gradient of the REVERSE edges (c' -> n) of its slots (n -> c'): they need
// Q[n][species of c'], i.e. only the row owner's own blocks (M x nz x 384 B, staged once per wave in LDS) instead of a
// 1.15 KB gather per slot from a different atom (the previous formulation: 3.3 GB of L2 gathers per launch).  Same
// distance, hence the same radial values; unit vector negated; the result is written to slot rev[e'].
__global__ void __launch_bounds__(256)
k_l0_bwd(int N, int M, int nz, int first_write, int fresh_mfma, int excl_vol, float rc, GraphView G, const int *__restrict__ counters,
         const float *__restrict__ Q, float4 *__restrict__ gbar, long long gbar_group_stride, int n_groups) {
    extern __shared__ __attribute__((aligned(16))) float qs_all[];   // [wave][m][species][4][24]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sl = lane >> 2, kq = lane & 3;
    const int n = blockIdx.x * (blockDim.x >> 6) + wave;
    if (n >= N || counters[2] || !G.act.atom(n)) return;
    const int per_model = nz * TBLK;
    // nothing to add to: the first write of the launch sequence, or (compact partial buffers) a chain whose reverse neighbor
    // kernels wrote elsewhere -- all slots of a row belong to the row owner's chain, and so do their reverse slots
    const bool fw = first_write || (fresh_mfma && G.chain_class[G.atom_cfg[n]] != EDGE_BCLASS_GATHER);
    float *qs = qs_all + (size_t)wave * M * per_model;
    for (int m = 0; m < M; ++m) {
        const float4 *src = reinterpret_cast<const float4 *>(Q + ((size_t)m * N + n) * per_model);
        for (int i = lane; i < per_model / 4; i += 64) reinterpret_cast<float4 *>(qs + m * per_model)[i] = src[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // one wave writes and reads its own block: LDS operations of a wave execute in order
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int e0 = G.row_start[n], e1 = G.row_start[n + 1];
    for (int eb = e0; eb < e1; eb += 16) {
        const int e = min(eb + sl, e1 - 1);
        const bool live = eb + sl < e1;
        const float4 ed = G.edge[e];
        const float2 dd = G.dist2[e];
        const int zi = G.zslot[e];          // species index of c' (255: pad)
        const int re = G.rev[e];            // slot of (c' -> n)
        // this quarter's radial values, rebuilt from the edge vector exactly as k_edge_geom computed them (radial_quarter): 16 bytes
        // per slot instead of the 192 bytes of two fp32 tables that no other kernel reads
        float rho[6], drho[6], inv;
        bool valid;
        radial_quarter(ed, kq, rc, rho, drho, inv, valid);
        const float4 er = make_float4(ed.x * inv, ed.y * inv, ed.z * inv, 0.f);
        const bool real = live && zi < nz && re >= 0 && valid;
        const int zq = zi < nz ? zi : 0;
        for (int m = 0; m < M; ++m) {
            const float *q = qs + m * per_model + zq * TBLK + kq * 6;
            float qv[4][6];
#pragma unroll
            for (int cq = 0; cq < 4; ++cq)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float2 t2 = reinterpret_cast<const float2 *>(q + cq * KP)[k];
                    qv[cq][2 * k] = t2.x; qv[cq][2 * k + 1] = t2.y;
                }
            float d0 = 0.f, dx = 0.f, dy = 0.f, dz = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float q0 = qv[0][k], qx = qv[1][k], qy = qv[2][k], qz = qv[3][k];
                d0 = fmaf(drho[k], q0, d0);
                dx = fmaf(drho[k], qx, dx); dy = fmaf(drho[k], qy, dy); dz = fmaf(drho[k], qz, dz);
                bx = fmaf(rho[k], qx, bx); by = fmaf(rho[k], qy, by); bz = fmaf(rho[k], qz, bz);
            }
            // reduce the 4 radial quarters (lanes 4 sl .. 4 sl + 3): fixed order, DPP
            auto qsum = [](float x) {
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
                return x;
            };
            d0 = qsum(d0); dx = qsum(dx); dy = qsum(dy); dz = qsum(dz); bx = qsum(bx); by = qsum(by); bz = qsum(bz);
            if (real && kq == 0) {
                const float ux = -er.x, uy = -er.y, uz = -er.z;   // unit vector c' -> n; the edge (n -> c') has -u
                float db = d0 - (dx * ux + dy * uy + dz * uz);
                if (excl_vol) db += dd.y;
                const float invd = inv;
                const float dotu = fmaf(bz, uz, fmaf(by, uy, bx * ux));
                float g0 = fmaf(-db, ux, (bx - dotu * ux) * invd);
                float g1 = fmaf(-db, uy, (by - dotu * uy) * invd);
                float g2 = fmaf(-db, uz, (bz - dotu * uz) * invd);
                float4 *gb = gbar + (size_t)m * n_groups * gbar_group_stride;   // group 0 of model m
                if (!fw) {
                    const float4 old = gb[re];
                    g0 += old.x; g1 += old.y; g2 += old.z;
                }
                gb[re] = make_float4(g0, g1, g2, 0.f);
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

// fp16-split fragment-order copies of one model's tables (pack_mfma_tiles16, painn_node_mfma.hip):
//   A16 [n_embed][2][1024 uint4]: matrix [F features][32 kappa (24 + zeros)]  -> forward B operand
//   At16[n_embed][2][1024 uint4]: matrix [32 kappa (24 + zeros)][F features]  -> reverse B operand
void l0_pack_tables(const float *A /*[n_embed][2][24][F]*/, int n_embed, unsigned *A16, unsigned *At16) {
    std::vector<float> fk((size_t)F * 32), kf((size_t)32 * F);
    for (int z = 0; z < n_embed; ++z)
        for (int sec = 0; sec < 2; ++sec) {
            std::fill(fk.begin(), fk.end(), 0.f);
            std::fill(kf.begin(), kf.end(), 0.f);
            for (int kap = 0; kap < KP; ++kap)
                for (int f = 0; f < F; ++f) {
                    const float v = A[(((size_t)z * 2 + sec) * KP + kap) * F + f];
                    fk[(size_t)f * 32 + kap] = v;
                    kf[(size_t)kap * F + f] = v;
                }
            const size_t o = ((size_t)z * 2 + sec) * L0_TILE_U4 * 4;
            pack_mfma_tiles16(fk.data(), F, 32, A16 + o);
            pack_mfma_tiles16(kf.data(), 32, F, At16 + o);
        }
}
size_t l0_packed_dwords(int n_embed) { return (size_t)n_embed * 2 * L0_TILE_U4 * 4; }

int l0_mfma_init(vssr_handle *h) {
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_l0_fwd16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VSSR_HIP(h, hipFuncSetAttribute((const void *)k_l0_q16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return VSSR_OK;
}

int l0_run_forward(vssr_handle *h, const GraphView &G, float *s_msg, float *v_msg) {
    const int N = h->n_atoms, M = h->n_models, nz = h->l0_nz;
    hipStream_t st = h->stream;
    if (h->d_l0T.ensure(sizeof(float) * (size_t)N * nz * TBLK) ||
        h->d_l0Q.ensure(sizeof(float) * (size_t)M * N * nz * TBLK))
        return set_err(h, VSSR_E_NOMEM, "layer-0 factorisation buffers: out of device memory");
    if (!h->l0T_by_geom) {   // (at most 4 species: k_edge_geom has already accumulated T, nbr.hip)
        const int *cnt = h->d_counters.as<int>();
        float *T = h->d_l0T.as<float>();
        switch (nz) {
            case 1: hipLaunchKernelGGL(k_l0_accum<1>, dim3(N), dim3(96), 0, st, G, cnt, T); break;
            case 2: hipLaunchKernelGGL(k_l0_accum<2>, dim3(N), dim3(96), 0, st, G, cnt, T); break;
            case 3: hipLaunchKernelGGL(k_l0_accum<3>, dim3(N), dim3(96), 0, st, G, cnt, T); break;
            case 4: hipLaunchKernelGGL(k_l0_accum<4>, dim3(N), dim3(96), 0, st, G, cnt, T); break;
            case 5: hipLaunchKernelGGL(k_l0_accum<5>, dim3(N), dim3(96), 0, st, G, cnt, T); break;
            case 6: hipLaunchKernelGGL(k_l0_accum<6>, dim3(N), dim3(96), 0, st, G, cnt, T); break;
            case 7: hipLaunchKernelGGL(k_l0_accum<7>, dim3(N), dim3(96), 0, st, G, cnt, T); break;
            default: hipLaunchKernelGGL(k_l0_accum<L0_MAX_SPECIES>, dim3(N), dim3(96), 0, st, G, cnt, T); break;
        }
    }
    const size_t lds_fwd = sizeof(_Float16) * (plane_halves(TA, 32 * nz) + plane_halves(3 * TA, 32 * nz));
    hipLaunchKernelGGL(k_l0_fwd16, dim3((N + TA - 1) / TA), dim3(NTHREADS), lds_fwd, st, N, M, nz, G.act, h->d_counters.as<int>(),
                       h->d_Z.as<int>(), h->d_zlist.as<int>(), h->model_table.as<ModelW>(), h->d_l0A.as<uint4>(), h->n_embed,
                       h->d_l0T.as<float>(), s_msg, v_msg);
    return VSSR_OK;
}

int l0_run_reverse(vssr_handle *h, const GraphView &G, int first_write, int fresh_mfma, const float *sbar_msg, const float *vbar_msg,
                   float4 *gbar, long long gbar_stride, int n_groups) {
    const int N = h->n_atoms, M = h->n_models, nz = h->l0_nz;
    hipStream_t st = h->stream;
    const size_t lds_q = sizeof(_Float16) * (plane_halves(TA, F) + plane_halves(3 * TA, F));
    hipLaunchKernelGGL(k_l0_q16, dim3((N + TA - 1) / TA, M), dim3(NTHREADS), lds_q, st, N, nz, G.act, h->d_counters.as<int>(),
                       h->d_zlist.as<int>(), h->d_l0At.as<uint4>(), h->n_embed, sbar_msg, vbar_msg, h->d_l0Q.as<float>());
    if (sizeof(float) * 4 * M * nz * TBLK > 48 * 1024)   // (3 models x 3 species: 13.8 KB)
        VSSR_HIP(h, hipFuncSetAttribute((const void *)k_l0_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k_l0_bwd, dim3((N + 3) / 4), dim3(256), sizeof(float) * 4 * M * nz * TBLK, st, N, M, nz, first_write, fresh_mfma,
                       h->excl_vol, h->cutoff, G, h->d_counters.as<int>(), h->d_l0Q.as<float>(), gbar, gbar_stride, n_groups);
    return VSSR_OK;
}

}  // namespace vssr
