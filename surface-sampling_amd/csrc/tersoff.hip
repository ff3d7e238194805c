// tersoff.hip — Tersoff energy / per-atom energy / forces on gfx950 (fp64), batched over
// independent configurations.
//
// Replaces LAMMMPSCalc.run_lammps_calc with `pair_style tersoff` (reference
// mcmc/calculators/calculators.py:507-640; potential mcmc/potentials/GaN.tersoff; template
// tutorials/data/GaN_0001/GaN_0001_lammps_energy_template.txt).  Semantics follow LAMMPS
// pair_tersoff: entry (i,j,k); two-body terms and fc(r_ij) from (i,j,j),
// three-body terms and fc(r_ik) from (i,j,k); b_ij with the LAMMPS asymptotic branches;
// pe/atom splits every directed pair term half/half between i and j.
//
// The site energy E_i depends only on the vectors r_ij from i to its neighbors, so one thread owns
// one centre: it writes G_slot = dE_i/d r_ij for its own slots, and the force on atom c is
// gathered as sum_{slots of c} (G[slot] - G[rev[slot]]) — no atomics, deterministic.
#include "vssr_internal.h"

namespace vssr {

struct TersP { double m, gamma, lam3, c, d, h, n, beta, lam2, B, R, D, lam1, A; };

__device__ inline double t_fc(double r, const TersP &p) {
    if (r < p.R - p.D) return 1.0;
    if (r > p.R + p.D) return 0.0;
    return 0.5 * (1.0 - sin(M_PI_2 * (r - p.R) / p.D));
}
__device__ inline double t_fc_d(double r, const TersP &p) {
    if (r < p.R - p.D || r > p.R + p.D) return 0.0;
    return -(M_PI_4 / p.D) * cos(M_PI_2 * (r - p.R) / p.D);
}
__device__ inline void t_gijk(double cs, const TersP &p, double &g, double &dg) {
    double c2 = p.c * p.c, d2 = p.d * p.d, hc = p.h - cs;
    double den = d2 + hc * hc;
    g = p.gamma * (1.0 + c2 / d2 - c2 / den);
    dg = p.gamma * (-2.0 * c2 * hc) / (den * den);
}
__device__ inline void t_ex(double rij, double rik, const TersP &p, double &ex, double &dex) {
    double arg = p.lam3 * (rij - rik), darg = p.lam3;
    if ((int)p.m == 3) {
        darg = 3.0 * p.lam3 * arg * arg;
        arg = arg * arg * arg;
    }
    if (arg > 69.0776) { ex = 1.e30; dex = 0.0; }
    else if (arg < -69.0776) { ex = 0.0; dex = 0.0; }
    else { ex = exp(arg); dex = ex * darg; }
}
__device__ inline void t_bij(double zeta, const TersP &p, double &b, double &db) {
    double tmp = p.beta * zeta, n = p.n;
    double c1 = pow(2.0 * n * 1.0e-16, -1.0 / n), c2 = pow(2.0 * n * 1.0e-8, -1.0 / n);
    double c3 = 1.0 / c2, c4 = 1.0 / c1;
    if (tmp > c1) { b = 1.0 / sqrt(tmp); db = p.beta * -0.5 * pow(tmp, -1.5); return; }
    if (tmp > c2) {
        b = (1.0 - pow(tmp, -n) / (2.0 * n)) / sqrt(tmp);
        db = p.beta * (-0.5 * pow(tmp, -1.5) * (1.0 - (1.0 + 1.0 / (2.0 * n)) * pow(tmp, -n)));
        return;
    }
    if (tmp < c4) { b = 1.0; db = 0.0; return; }
    if (tmp < c3) { b = 1.0 - pow(tmp, n) / (2.0 * n); db = -0.5 * p.beta * pow(tmp, n - 1.0); return; }
    double tn = pow(tmp, n);
    b = pow(1.0 + tn, -1.0 / (2.0 * n));
    db = -0.5 * pow(1.0 + tn, -1.0 - 1.0 / (2.0 * n)) * tn / zeta;
}

__device__ inline void edge_vec(const double *__restrict__ wpos, const double *C, int i, int j, int packedS,
                                double r[3]) {
    int s0 = (packedS & 255) - 128, s1 = ((packedS >> 8) & 255) - 128, s2 = ((packedS >> 16) & 255) - 128;
    for (int x = 0; x < 3; ++x)
        r[x] = wpos[3 * j + x] - wpos[3 * i + x] + s0 * C[x] + s1 * C[3 + x] + s2 * C[6 + x];
}

__global__ void k_tersoff_site(int N, int nt, const TersP *__restrict__ P, const int *__restrict__ type,
                               const int *__restrict__ atom_cfg, const double *__restrict__ cell,
                               const double *__restrict__ wpos, const int *__restrict__ row_start,
                               const float4 *__restrict__ edge, const int *__restrict__ edge_S,
                               const int *__restrict__ counters, double *__restrict__ eps /*[slots]*/,
                               double *__restrict__ gslot /*[slots][3]*/, ActiveView av) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || counters[2] || !av.atom(i)) return;
    const double *C = cell + 9 * atom_cfg[i];
    const int ti = type[i];
    const int e0 = row_start[i], e1 = row_start[i + 1];
    for (int e = e0; e < e1; ++e) {
        eps[e] = 0.0;
        gslot[3 * e] = 0.0; gslot[3 * e + 1] = 0.0; gslot[3 * e + 2] = 0.0;
    }
    for (int e = e0; e < e1; ++e) {
        int j = __float_as_int(edge[e].w);
        if (j < 0) continue;
        const int tj = type[j];
        const TersP pij = P[(ti * nt + tj) * nt + tj];
        double rij[3];
        edge_vec(wpos, C, i, j, edge_S[e], rij);
        double r = sqrt(rij[0] * rij[0] + rij[1] * rij[1] + rij[2] * rij[2]);
        if (r > pij.R + pij.D) continue;
        double fc = t_fc(r, pij), dfc = t_fc_d(r, pij);
        double fR = pij.A * exp(-pij.lam1 * r), fA = -pij.B * exp(-pij.lam2 * r);
        double zeta = 0.0;
        for (int e2 = e0; e2 < e1; ++e2) {
            int k = __float_as_int(edge[e2].w);
            if (e2 == e || k < 0) continue;
            const TersP pk = P[(ti * nt + tj) * nt + type[k]];
            double rik[3];
            edge_vec(wpos, C, i, k, edge_S[e2], rik);
            double r2 = sqrt(rik[0] * rik[0] + rik[1] * rik[1] + rik[2] * rik[2]);
            if (r2 > pk.R + pk.D) continue;
            double cs = (rij[0] * rik[0] + rij[1] * rik[1] + rij[2] * rik[2]) / (r * r2);
            double g, dg, ex, dex;
            t_gijk(cs, pk, g, dg);
            t_ex(r, r2, pk, ex, dex);
            zeta += t_fc(r2, pk) * g * ex;
        }
        double bij, dbij;
        t_bij(zeta, pij, bij, dbij);
        eps[e] = 0.5 * fc * (fR + bij * fA);
        double dV_dr = 0.5 * (dfc * (fR + bij * fA) + fc * (-pij.lam1 * fR - pij.lam2 * bij * fA));
        double pref = 0.5 * fc * fA * dbij;
        double gij[3] = {dV_dr * rij[0] / r, dV_dr * rij[1] / r, dV_dr * rij[2] / r};
        if (pref != 0.0) {
            for (int e2 = e0; e2 < e1; ++e2) {
                int k = __float_as_int(edge[e2].w);
                if (e2 == e || k < 0) continue;
                const TersP pk = P[(ti * nt + tj) * nt + type[k]];
                double rik[3];
                edge_vec(wpos, C, i, k, edge_S[e2], rik);
                double r2 = sqrt(rik[0] * rik[0] + rik[1] * rik[1] + rik[2] * rik[2]);
                if (r2 > pk.R + pk.D) continue;
                double cs = (rij[0] * rik[0] + rij[1] * rik[1] + rij[2] * rik[2]) / (r * r2);
                double g, dg, ex, dex;
                t_gijk(cs, pk, g, dg);
                t_ex(r, r2, pk, ex, dex);
                double fck = t_fc(r2, pk), dfck = t_fc_d(r2, pk);
                for (int x = 0; x < 3; ++x) {
                    double dcs_drij = (rik[x] / r2 - cs * rij[x] / r) / r;
                    double dcs_drik = (rij[x] / r - cs * rik[x] / r2) / r2;
                    double dz_drij = fck * (dg * dcs_drij * ex + g * dex * rij[x] / r);
                    double dz_drik = dfck * rik[x] / r2 * g * ex + fck * (dg * dcs_drik * ex - g * dex * rik[x] / r2);
                    gij[x] += pref * dz_drij;
                    gslot[3 * e2 + x] += pref * dz_drik;
                }
            }
        }
        for (int x = 0; x < 3; ++x) gslot[3 * e + x] += gij[x];
    }
}

__global__ void k_tersoff_gather(int N, const int *__restrict__ row_start, const int *__restrict__ rev,
                                 const int *__restrict__ counters, const double *__restrict__ eps,
                                 const double *__restrict__ gslot, double *__restrict__ e_atom,
                                 double *__restrict__ forces, ActiveView av) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N || counters[2] || !av.atom(c)) return;
    double ea = 0.0, f0 = 0.0, f1 = 0.0, f2 = 0.0;
    for (int e = row_start[c]; e < row_start[c + 1]; ++e) {
        int r = rev[e];
        if (r < 0) continue;
        ea += 0.5 * (eps[e] + eps[r]);
        f0 += gslot[3 * e] - gslot[3 * r];
        f1 += gslot[3 * e + 1] - gslot[3 * r + 1];
        f2 += gslot[3 * e + 2] - gslot[3 * r + 2];
    }
    e_atom[c] = ea;
    forces[3 * c] = f0; forces[3 * c + 1] = f1; forces[3 * c + 2] = f2;
}

__global__ void __launch_bounds__(256)
k_tersoff_energy(const int *__restrict__ cfg_start, const double *__restrict__ e_atom, double *__restrict__ energy,
                 const unsigned char *__restrict__ active) {
    __shared__ double red[256];
    int b = blockIdx.x, tid = threadIdx.x;
    if (active && !active[b]) return;
    double acc = 0.0;
    for (int i = cfg_start[b] + tid; i < cfg_start[b + 1]; i += blockDim.x) acc += e_atom[i];
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) energy[b] = red[0];
}

int tersoff_run(vssr_handle *h, uint32_t want) {
    (void)want;
    const int N = h->n_atoms;
    hipStream_t st = h->stream;
    int rc = build_neighbors(h, h->ters_cutmax);
    if (rc) return rc;
    if (h->d_ters_e.ensure(sizeof(double) * h->n_cfg) || h->d_ters_ea.ensure(sizeof(double) * N) ||
        h->d_ters_f.ensure(sizeof(double) * 3 * N) || h->d_gbar.ensure(sizeof(double) * 4 * (size_t)h->slot_cap))
        return set_err(h, VSSR_E_NOMEM, "tersoff buffers: out of device memory");
    double *eps = h->d_gbar.as<double>();
    double *gslot = eps + h->slot_cap;
    h->prof.begin(KC_TERSOFF, st);
    dim3 blk(64), grd((N + 63) / 64);
    const ActiveView av{h->active_mask, h->d_atom_cfg.as<int>()};
    hipLaunchKernelGGL(k_tersoff_site, grd, blk, 0, st, N, h->n_types, h->ters_params.as<TersP>(), h->d_Z.as<int>(),
                       h->d_atom_cfg.as<int>(), h->d_cell.as<double>(), h->d_wpos.as<double>(),
                       h->d_row_start.as<int>(), h->d_edge.as<float4>(), h->d_edge_S.as<int>(),
                       h->d_counters.as<int>(), eps, gslot, av);
    hipLaunchKernelGGL(k_tersoff_gather, grd, blk, 0, st, N, h->d_row_start.as<int>(), h->d_rev.as<int>(),
                       h->d_counters.as<int>(), eps, gslot, h->d_ters_ea.as<double>(), h->d_ters_f.as<double>(), av);
    hipLaunchKernelGGL(k_tersoff_energy, dim3(h->n_cfg), dim3(256), 0, st, h->d_cfg_start.as<int>(),
                       h->d_ters_ea.as<double>(), h->d_ters_e.as<double>(), h->active_mask);
    h->prof.end(st);
    VSSR_HIP(h, hipGetLastError());
    return VSSR_OK;
}

}  // namespace vssr
