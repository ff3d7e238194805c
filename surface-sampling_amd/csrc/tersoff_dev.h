// tersoff_dev.h — device bodies of the Tersoff kernels (tersoff.hip), shared with the chain-resident minimiser (chain_min.hip): one
// implementation of the site energy / gradient arithmetic, so both drivers produce the same bits.
#ifndef VSSR_TERSOFF_DEV_H
#define VSSR_TERSOFF_DEV_H
#include "vssr_internal.h"

namespace vssr {

struct TersP { double m, gamma, lam3, c, d, h, n, beta, lam2, B, R, D, lam1, A; };

// sin and cos of an argument in [-pi/2, pi/2] (the cutoff shell maps onto exactly that interval): Taylor series to x^21 / x^22,
// truncation error < 2e-18.  The library's fp64 sin() + cos() were ~300 instructions per three-body term once any lane of a
// wave sat in a cutoff shell -- together with exp() most of what k_tersoff_site4 executed (profiles/r04/NOTES_tersoff.md).
__device__ __forceinline__ void t_sincos_half_pi(double x, double &sn, double &cs) {
    const double z = x * x;
    double ps = -1.0 / 51090942171709440000.0;                 // -1/21!
    ps = fma(ps, z, 1.0 / 121645100408832000.0);               //  1/19!
    ps = fma(ps, z, -1.0 / 355687428096000.0);                 // -1/17!
    ps = fma(ps, z, 1.0 / 1307674368000.0);                    //  1/15!
    ps = fma(ps, z, -1.0 / 6227020800.0);                      // -1/13!
    ps = fma(ps, z, 1.0 / 39916800.0);                         //  1/11!
    ps = fma(ps, z, -1.0 / 362880.0);                          // -1/9!
    ps = fma(ps, z, 1.0 / 5040.0);                             //  1/7!
    ps = fma(ps, z, -1.0 / 120.0);                             // -1/5!
    ps = fma(ps, z, 1.0 / 6.0);                                //  1/3!
    sn = fma(-x * z, ps, x);
    double pc = 1.0 / 1124000727777607680000.0;                //  1/22!
    pc = fma(pc, z, -1.0 / 2432902008176640000.0);             // -1/20!
    pc = fma(pc, z, 1.0 / 6402373705728000.0);                 //  1/18!
    pc = fma(pc, z, -1.0 / 20922789888000.0);                  // -1/16!
    pc = fma(pc, z, 1.0 / 87178291200.0);                      //  1/14!
    pc = fma(pc, z, -1.0 / 479001600.0);                       // -1/12!
    pc = fma(pc, z, 1.0 / 3628800.0);                          //  1/10!
    pc = fma(pc, z, -1.0 / 40320.0);                           // -1/8!
    pc = fma(pc, z, 1.0 / 720.0);                              //  1/6!
    pc = fma(pc, z, -1.0 / 24.0);                              // -1/4!
    pc = fma(pc, z, 0.5);                                      //  1/2!
    cs = fma(-z, pc, 1.0);
}
// cutoff function and its derivative (LAMMPS ters_fc / ters_fc_d)
__device__ inline void t_fc_both(double r, const TersP &p, double &fc, double &dfc) {
    if (r < p.R - p.D) { fc = 1.0; dfc = 0.0; return; }
    if (r > p.R + p.D) { fc = 0.0; dfc = 0.0; return; }
    double sn, cs;
    t_sincos_half_pi(M_PI_2 * (r - p.R) / p.D, sn, cs);
    fc = 0.5 * (1.0 - sn);
    dfc = -(M_PI_4 / p.D) * cs;
}
__device__ inline double t_fc(double r, const TersP &p) {
    double fc, dfc;
    t_fc_both(r, p, fc, dfc);
    return fc;
}
__device__ inline double t_fc_d(double r, const TersP &p) {
    double fc, dfc;
    t_fc_both(r, p, fc, dfc);
    return dfc;
}
__device__ inline void t_gijk(double cs, const TersP &p, double &g, double &dg) {
    double c2 = p.c * p.c, d2 = p.d * p.d, hc = p.h - cs;
    double inv = 1.0 / (d2 + hc * hc);
    g = p.gamma * (1.0 + c2 / d2 - c2 * inv);
    dg = p.gamma * (-2.0 * c2 * hc) * (inv * inv);
}
__device__ inline void t_ex(double rij, double rik, const TersP &p, double &ex, double &dex) {
    if (p.lam3 == 0.0) { ex = 1.0; dex = 0.0; return; }   // exp(0) = 1, derivative lam3 (or 3 lam3 arg^2) = 0: the same values, no exp()
    double arg = p.lam3 * (rij - rik), darg = p.lam3;
    if ((int)p.m == 3) {
        darg = 3.0 * p.lam3 * arg * arg;
        arg = arg * arg * arg;
    }
    if (arg > 69.0776) { ex = 1.e30; dex = 0.0; }
    else if (arg < -69.0776) { ex = 0.0; dex = 0.0; }
    else { ex = exp(arg); dex = ex * darg; }
}
// thresholds of the asymptotic branches of b_ij (LAMMPS pair_tersoff.cpp ters_bij): functions of the entry's n only
__device__ inline void t_bij_limits(double n, double &c1, double &c2) {
    if (n == 1.0) { c1 = 1.0 / 2.0e-16; c2 = 1.0 / 2.0e-8; return; }
    c1 = pow(2.0 * n * 1.0e-16, -1.0 / n);
    c2 = pow(2.0 * n * 1.0e-8, -1.0 / n);
}
__device__ inline void t_bij(double zeta, const TersP &p, double c1, double c2, double &b, double &db) {
    double tmp = p.beta * zeta, n = p.n;
    double c3 = 1.0 / c2, c4 = 1.0 / c1;
    if (tmp > c1) { b = 1.0 / sqrt(tmp); db = p.beta * -0.5 * pow(tmp, -1.5); return; }
    if (tmp > c2) {
        b = (1.0 - pow(tmp, -n) / (2.0 * n)) / sqrt(tmp);
        db = p.beta * (-0.5 * pow(tmp, -1.5) * (1.0 - (1.0 + 1.0 / (2.0 * n)) * pow(tmp, -n)));
        return;
    }
    if (tmp < c4) { b = 1.0; db = 0.0; return; }
    if (tmp < c3) { b = 1.0 - pow(tmp, n) / (2.0 * n); db = -0.5 * p.beta * pow(tmp, n - 1.0); return; }
    if (n == 1.0) {   // (1 + x)^(-1/2) and (1 + x)^(-3/2) without pow(): the GaN entries, and every potential with n = 1
        const double s1 = 1.0 + tmp;
        b = 1.0 / sqrt(s1);
        db = -0.5 * (b / s1) * tmp / zeta;
        return;
    }
    double tn = pow(tmp, n);
    b = pow(1.0 + tn, -1.0 / (2.0 * n));
    db = -0.5 * (b / (1.0 + tn)) * tn / zeta;   // (1 + tn)^(-1 - 1/(2n)) = b / (1 + tn)
}
__device__ inline void t_bij(double zeta, const TersP &p, double &b, double &db) {
    double c1, c2;
    t_bij_limits(p.n, c1, c2);
    t_bij(zeta, p, c1, c2, b, db);
}

__device__ inline void edge_vec(const double *__restrict__ wpos, const double *C, int i, int j, int packedS,
                                double r[3]) {
    int s0 = (packedS & 255) - 128, s1 = ((packedS >> 8) & 255) - 128, s2 = ((packedS >> 16) & 255) - 128;
    for (int x = 0; x < 3; ++x)
        r[x] = wpos[3 * j + x] - wpos[3 * i + x] + s0 * C[x] + s1 * C[3 + x] + s2 * C[6 + x];
}

// one thread per centre i (rows longer than `longer_than` slots; -1: every row).  row_start: indexed [i], [i + 1]
__device__ inline void tersoff_site_atom(int i, int nt, const TersP *__restrict__ P, const int *__restrict__ type,
                               const int *__restrict__ atom_cfg, const double *__restrict__ cell,
                               const double *__restrict__ wpos, const int *__restrict__ row_start,
                               const float4 *__restrict__ edge, const int *__restrict__ edge_S,
                               double *__restrict__ eps /*[slots]*/, double *__restrict__ gslot /*[slots][3]*/, int longer_than) {
    const double *C = cell + 9 * atom_cfg[i];
    const int ti = type[i];
    const int e0 = row_start[i], e1 = row_start[i + 1];
    if (e1 - e0 <= longer_than) return;   // rows k_tersoff_site4 has done (-1: every row)
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


#ifndef TS_MARK   // phase clocks of the chain-resident minimiser's debug build (chain_min.hip, -DCM_PHASE_TIMING)
#define TS_MARK_INIT
#define TS_MARK(k)
#endif

constexpr int TS_MAXD = 16, TS_CENTRES = 64, TS_LANES = 4, TS_MAXP = 64;   // slots per row in LDS; centres / workgroup; lanes / centre; 4^3 entries

// A parameter entry as k_tersoff_site4 keeps it in LDS: what every three-body term would otherwise recompute from the file's
// fields is done once per workgroup -- gamma (1 + c^2/d^2), gamma c^2, d^2 (g(theta) is left with ONE fp64 division), the shell
// bounds R -+ D and pi / (2 D) (no division inside the cutoff function), the b_ij branch thresholds, m == 3 as a flag.
struct TersL {
    double Rmax, Rmin, R, piD2, g0, g1, d2, h, lam3;       // three-body use of the entry (i, j, k)
    double A, lam1, B, lam2, beta, n, c1, c2, c3, c4;      // pair use of the entry (i, j, j); c1..c4: b_ij branch thresholds
    int m3, pad;
};
__device__ inline TersL t_derive(const TersP &p) {
    TersL l;
    l.Rmax = p.R + p.D; l.Rmin = p.R - p.D; l.R = p.R; l.piD2 = M_PI_2 / p.D;
    const double c2 = p.c * p.c, d2 = p.d * p.d;
    l.g0 = p.gamma * (1.0 + c2 / d2); l.g1 = p.gamma * c2; l.d2 = d2; l.h = p.h; l.lam3 = p.lam3;
    l.A = p.A; l.lam1 = p.lam1; l.B = p.B; l.lam2 = p.lam2; l.beta = p.beta; l.n = p.n;
    t_bij_limits(p.n, l.c1, l.c2);
    l.c3 = 1.0 / l.c2; l.c4 = 1.0 / l.c1;
    l.m3 = (int)p.m == 3; l.pad = 0;
    return l;
}
__device__ __forceinline__ void t_fc_both(double r, const TersL &p, double &fc, double &dfc) {
    if (r < p.Rmin) { fc = 1.0; dfc = 0.0; return; }
    if (r > p.Rmax) { fc = 0.0; dfc = 0.0; return; }
    double sn, cs;
    t_sincos_half_pi(p.piD2 * (r - p.R), sn, cs);
    fc = 0.5 * (1.0 - sn);
    dfc = -0.5 * p.piD2 * cs;
}
struct TersTri { double g, dg, ex, dex, fc, dfc; };
// The three-body fields of an entry as a register copy.  The (j, k) loops of the site tile read everything an iteration may need
// from LDS in ONE batch in front of its branches (tri_fields + pin): with the reads left behind the early-outs the compiler emitted
// five to six dependent LDS round trips per iteration (type -> pref -> vectors -> Rmax -> shape fields -> cutoff fields, ~130 cycles
// each at one or two waves per SIMD), more than the arithmetic of the term (profiles/r05/NOTES_tersoff.md section 5).
struct TersL3 { double Rmax, Rmin, R, piD2, g0, g1, d2, h, lam3; int m3; };
__device__ __forceinline__ void pin(double &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(int &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ TersL3 tri_fields(const TersL &e) {
    TersL3 p = {e.Rmax, e.Rmin, e.R, e.piD2, e.g0, e.g1, e.d2, e.h, e.lam3, e.m3};
    pin(p.Rmax); pin(p.Rmin); pin(p.R); pin(p.piD2); pin(p.g0); pin(p.g1); pin(p.d2); pin(p.h); pin(p.lam3); pin(p.m3);
    return p;
}
__device__ __forceinline__ void t_fc_both(double r, const TersL3 &p, double &fc, double &dfc) {
    if (r < p.Rmin) { fc = 1.0; dfc = 0.0; return; }
    if (r > p.Rmax) { fc = 0.0; dfc = 0.0; return; }
    double sn, cs;
    t_sincos_half_pi(p.piD2 * (r - p.R), sn, cs);
    fc = 0.5 * (1.0 - sn);
    dfc = -0.5 * p.piD2 * cs;
}
// three-body factors of (i, j, k) with entry p: false when k is outside the entry's cutoff
template <class ENTRY>
__device__ __forceinline__ bool t_tri(const ENTRY &p, double rj, double rk, double cs, TersTri &o) {
    if (rk > p.Rmax) return false;
    const double hc = p.h - cs, inv = 1.0 / (p.d2 + hc * hc);
    o.g = p.g0 - p.g1 * inv;
    o.dg = -2.0 * p.g1 * hc * (inv * inv);
    if (p.lam3 == 0.0) { o.ex = 1.0; o.dex = 0.0; }   // exp(0) = 1 and a zero derivative: the same values without exp()
    else {
        double arg = p.lam3 * (rj - rk), darg = p.lam3;
        if (p.m3) { darg = 3.0 * p.lam3 * arg * arg; arg = arg * arg * arg; }
        if (arg > 69.0776) { o.ex = 1.e30; o.dex = 0.0; }
        else if (arg < -69.0776) { o.ex = 0.0; o.dex = 0.0; }
        else { o.ex = exp(arg); o.dex = o.ex * darg; }
    }
    t_fc_both(rk, p, o.fc, o.dfc);
    return true;
}
// b_ij and its derivative from the derived entry: the branches of t_bij with the thresholds read instead of recomputed
__device__ __forceinline__ void t_bij(double zeta, const TersL &l, double &b, double &db) {
    const double tmp = l.beta * zeta, n = l.n;
    if (tmp > l.c1) { b = 1.0 / sqrt(tmp); db = l.beta * -0.5 * pow(tmp, -1.5); return; }
    if (tmp > l.c2) {
        b = (1.0 - pow(tmp, -n) / (2.0 * n)) / sqrt(tmp);
        db = l.beta * (-0.5 * pow(tmp, -1.5) * (1.0 - (1.0 + 1.0 / (2.0 * n)) * pow(tmp, -n)));
        return;
    }
    if (tmp < l.c4) { b = 1.0; db = 0.0; return; }
    if (tmp < l.c3) { b = 1.0 - pow(tmp, n) / (2.0 * n); db = -0.5 * l.beta * pow(tmp, n - 1.0); return; }
    if (n == 1.0) {
        const double s1 = 1.0 + tmp, rs = 1.0 / sqrt(s1);
        b = rs;
        db = -0.5 * (rs * rs * rs) * l.beta;   // -1/2 (1 + x)^(-3/2) beta  (= -1/2 b / (1 + x) tmp / zeta)
        return;
    }
    const double tn = pow(tmp, n);
    b = pow(1.0 + tn, -1.0 / (2.0 * n));
    db = -0.5 * (b / (1.0 + tn)) * tn / zeta;
}

// LDS of a 64-centre tile: neighborhood (unit vectors i -> n, distances, types; -1 = padding slot), pref_j, derived parameter entries
struct TersShared {
    double ux[TS_MAXD][TS_CENTRES], uy[TS_MAXD][TS_CENTRES], uz[TS_MAXD][TS_CENTRES];
    double r[TS_MAXD][TS_CENTRES], pref[TS_MAXD][TS_CENTRES];
    signed char tp[TS_MAXD][TS_CENTRES];
    TersL P[TS_MAXP];
};
// (no barrier: the first barrier of tersoff_site4_tile covers it)
__device__ __forceinline__ void tersoff_derive_params(TersShared &sh, int nt, const TersP *__restrict__ P) {
    for (int t = threadIdx.x; t < nt * nt * nt; t += TS_CENTRES * TS_LANES) sh.P[t] = t_derive(P[t]);
}
// One tile of TS_CENTRES centres, TS_LANES lanes each: thread tid serves centre i = (tile's first atom) + (tid >> 2) with lane
// q = tid & 3; `mine`: the centre exists and is evaluated.  Every thread of the 256-thread workgroup calls this (two barriers
// inside); sh.P must have been written (tersoff_derive_params) before.  row_start: indexed [i], [i + 1].
__device__ __forceinline__ void tersoff_site4_tile(TersShared &sh, int i, bool mine, int nt, const int *__restrict__ type,
                const int *__restrict__ atom_cfg, const double *__restrict__ cell,
                const double *__restrict__ wpos, const int *__restrict__ row_start,
                const float4 *__restrict__ edge, const int *__restrict__ edge_S,
                double *__restrict__ eps /*[slots]*/, double *__restrict__ gslot /*[slots][3]*/) {
    auto &s_ux = sh.ux; auto &s_uy = sh.uy; auto &s_uz = sh.uz; auto &s_r = sh.r; auto &s_pref = sh.pref; auto &s_tp = sh.tp; auto &s_P = sh.P;
    const int tid = threadIdx.x, cb = tid >> 2, q = tid & 3;
    int e0 = 0, deg = 0, ti = 0;
    TS_MARK_INIT
    if (mine) {
        e0 = row_start[i];
        deg = row_start[i + 1] - e0;
        mine = deg <= TS_MAXD;   // longer rows: tersoff_site_atom
        ti = type[i];
    }
    if (!mine) deg = 0;
    // ---- neighborhood -> LDS -----------------------------------------------------------------------------------
    if (deg > 0) {
        const double *C = cell + 9 * atom_cfg[i];
        for (int n = q; n < deg; n += TS_LANES) {
            const int j = __float_as_int(edge[e0 + n].w);
            signed char tp = -1;
            double u[3] = {0.0, 0.0, 0.0}, r = 0.0;
            if (j >= 0) {
                edge_vec(wpos, C, i, j, edge_S[e0 + n], u);
                r = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
                const double inv = 1.0 / r;
                u[0] *= inv; u[1] *= inv; u[2] *= inv;
                tp = (signed char)type[j];
            }
            s_ux[n][cb] = u[0]; s_uy[n][cb] = u[1]; s_uz[n][cb] = u[2];
            s_r[n][cb] = r; s_tp[n][cb] = tp;
        }
    }
    __syncthreads();
    TS_MARK(0)
    // ---- pass 1: the lane's slots as j ---------------------------------------------------------------------------
    // zeta_n over all k, b_ij, the pair energy -- and, in the same walk over k, the sums that make up the slot's OWN three-body
    // gradient d zeta_n / d r_n: it is linear in pref_n = 1/2 fc fA db/dzeta, which is only known once zeta_n is complete, so the
    // unscaled sums Sa = sum_k fc dg ex (v_k - cs u) and Sb = sum_k fc g dex are collected first and scaled afterwards.  (Until
    // round 5 pass 2 evaluated these factors a second time: three evaluations of the three-body factors per ordered pair (j, k),
    // now two -- the site kernel is ~55 % of an evaluation of the GaN workloads, profiles/r05/NOTES_tersoff.md.)
    // (rolled loops: the body holds inlined fp64 exp / sin / cos / pow; four unrolled copies of both passes were 140 KB of code,
    // more than the instruction cache two CUs share)
#pragma unroll 1
    for (int n = q; n < deg; n += TS_LANES) {
        const int tj = s_tp[n][cb];
        double e_pair = 0.0, pref = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
        if (tj >= 0) {
            const int eij = (ti * nt + tj) * nt + tj;
            const TersL &pij = s_P[eij];
            const double r = s_r[n][cb];
            if (r <= pij.Rmax) {
                const double ux = s_ux[n][cb], uy = s_uy[n][cb], uz = s_uz[n][cb];
                double zeta = 0.0, sax = 0.0, say = 0.0, saz = 0.0, sb = 0.0;
#pragma unroll 1
                for (int m = 0; m < deg; ++m) {
                    // every LDS operand of the (n, m) term in one batch, in front of the branches (a padding slot reads entry .. 0)
                    int tk = s_tp[m][cb];
                    double rk = s_r[m][cb];
                    double vx = s_ux[m][cb], vy = s_uy[m][cb], vz = s_uz[m][cb];
                    const TersL3 pk = tri_fields(s_P[(ti * nt + tj) * nt + max(tk, 0)]);
                    pin(tk); pin(rk); pin(vx); pin(vy); pin(vz);
                    if (m == n || tk < 0) continue;
                    const double cs = ux * vx + uy * vy + uz * vz;
                    TersTri t;
                    if (!t_tri(pk, r, rk, cs, t)) continue;
                    zeta += t.fc * t.g * t.ex;
                    const double a = t.fc * t.dg * t.ex;
                    sax += a * (vx - cs * ux);
                    say += a * (vy - cs * uy);
                    saz += a * (vz - cs * uz);
                    sb += t.fc * t.g * t.dex;
                }
                double fc, dfc;
                t_fc_both(r, pij, fc, dfc);
                const double fR = pij.A * exp(-pij.lam1 * r), fA = -pij.B * exp(-pij.lam2 * r);
                double bij, dbij;
                t_bij(zeta, pij, bij, dbij);
                e_pair = 0.5 * fc * (fR + bij * fA);
                const double dVdr = 0.5 * (dfc * (fR + bij * fA) + fc * (-pij.lam1 * fR - pij.lam2 * bij * fA));
                pref = 0.5 * fc * fA * dbij;
                // the slot as j, complete: dV/dr u + pref (Sa / r + Sb u)
                const double pa = pref / r, su = dVdr + pref * sb;
                gx = pa * sax + su * ux;
                gy = pa * say + su * uy;
                gz = pa * saz + su * uz;
            }
        }
        eps[e0 + n] = e_pair;
        gslot[3 * (e0 + n)] = gx; gslot[3 * (e0 + n) + 1] = gy; gslot[3 * (e0 + n) + 2] = gz;   // pass 2 (the same lane) adds the rest
        s_pref[n][cb] = pref;
    }
    __syncthreads();
    TS_MARK(1)
    // ---- pass 2: the lane's slots as k of every other j (d zeta_m / d r_n, weighted with pref_m) --------------------------------
#pragma unroll 1
    for (int n = q; n < deg; n += TS_LANES) {
        const int tn = s_tp[n][cb];
        if (tn < 0) continue;
        const double ux = s_ux[n][cb], uy = s_uy[n][cb], uz = s_uz[n][cb], r = s_r[n][cb], inv_r = 1.0 / r;
        double gx = 0.0, gy = 0.0, gz = 0.0, su = 0.0;   // su: coefficient of u collected over all terms
        bool any = false;
#pragma unroll 1
        for (int m = 0; m < deg; ++m) {
            int tm = s_tp[m][cb];
            double pf = s_pref[m][cb];
            double vx = s_ux[m][cb], vy = s_uy[m][cb], vz = s_uz[m][cb], rm = s_r[m][cb];
            const TersL3 pm = tri_fields(s_P[(ti * nt + max(tm, 0)) * nt + tn]);
            pin(tm); pin(pf); pin(vx); pin(vy); pin(vz); pin(rm);
            if (m == n || tm < 0) continue;
            if (pf == 0.0) continue;
            const double cs = ux * vx + uy * vy + uz * vz;
            TersTri t;
            if (!t_tri(pm, rm, r, cs, t)) continue;   // m as j, n as k
            const double a = pf * t.fc * t.dg * t.ex * inv_r;
            gx += a * (vx - cs * ux);
            gy += a * (vy - cs * uy);
            gz += a * (vz - cs * uz);
            su += pf * (t.dfc * t.g * t.ex - t.fc * t.g * t.dex);
            any = true;
        }
        if (any) {
            gslot[3 * (e0 + n)] += gx + su * ux;
            gslot[3 * (e0 + n) + 1] += gy + su * uy;
            gslot[3 * (e0 + n) + 2] += gz + su * uz;
        }
    }
    TS_MARK(2)
}

// row_start: indexed [c], [c + 1]
__device__ __forceinline__ void tersoff_gather_atom(int c, const int *__restrict__ row_start, const int *__restrict__ rev,
                                 const double *__restrict__ eps, const double *__restrict__ gslot, double *__restrict__ e_atom,
                                 double *__restrict__ forces) {
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

// energy of chain b: 256 threads (strided partial sums, binary tree in LDS); red: 256 doubles
__device__ __forceinline__ void tersoff_chain_energy(int b, double *red, const int *__restrict__ cfg_start, const double *__restrict__ e_atom,
                                                     double *__restrict__ energy) {
    const int tid = threadIdx.x;
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

}  // namespace vssr
#endif
