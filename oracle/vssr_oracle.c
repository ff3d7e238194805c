/*
 * vssr_oracle.c — CPU ORACLE (test infrastructure; see vssr_oracle.h for scope + provenance).
 * Not linked, loaded or called by the product path.
 */
#include "vssr_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

int orc_set_threads(int n) {
    if (n < 1) n = 1;
    omp_set_num_threads(n);
    return n;
}

/* ---- PaiNN in double and in float ------------------------------------------------------ */
#define REAL double
#define SUFFIX _f64
#define REXP exp
#define RSIN sin
#define RCOS cos
#define RSQRT sqrt
#define RPOW pow
#include "painn_impl.inc"
#undef REAL
#undef SUFFIX
#undef REXP
#undef RSIN
#undef RCOS
#undef RSQRT
#undef RPOW

#define REAL float
#define SUFFIX _f32
#define REXP expf
#define RSIN sinf
#define RCOS cosf
#define RSQRT sqrtf
#define RPOW powf
#include "painn_impl.inc"
#undef REAL
#undef SUFFIX
#undef REXP
#undef RSIN
#undef RCOS
#undef RSQRT
#undef RPOW

/* ---- cell helpers ----------------------------------------------------------------------- */
static void cross3(const double a[3], const double b[3], double c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* inverse cell (columns = reciprocal vectors / volume) and perpendicular heights */
static int cell_setup(const double cell[9], double inv[9], double height[3]) {
    const double *a = cell, *b = cell + 3, *c = cell + 6;
    double bc[3], ca[3], ab[3];
    cross3(b, c, bc); cross3(c, a, ca); cross3(a, b, ab);
    double vol = dot3(a, bc);
    if (fabs(vol) < 1e-12) return -1;
    for (int x = 0; x < 3; ++x) {
        inv[0 * 3 + x] = bc[x] / vol; /* frac_a = inv[0].r */
        inv[1 * 3 + x] = ca[x] / vol;
        inv[2 * 3 + x] = ab[x] / vol;
    }
    height[0] = fabs(vol) / sqrt(dot3(bc, bc));
    height[1] = fabs(vol) / sqrt(dot3(ca, ca));
    height[2] = fabs(vol) / sqrt(dot3(ab, ab));
    return 0;
}

/*
 * Neighbor multigraph (a pair may occur through several images).
 * nff builds its list at cutoff + skin and PaiNN trims it to d <= cutoff on every call
 * (Appendix A item 1); single-point parity needs only the trimmed set, built here directly.
 * Positions are wrapped into the cell along periodic axes first so that the image range
 * floor(cutoff/height)+1 is sufficient wherever the caller left the atoms.
 */
int64_t orc_neighbors(int32_t n, const double *pos, const double cell[9], const uint8_t pbc[3],
                      double cutoff, int64_t cap, int32_t *ei, int32_t *ej, int32_t *eS, double *er) {
    double inv[9], height[3];
    if (n < 0 || cutoff <= 0) return -1;
    int any_pbc = pbc[0] || pbc[1] || pbc[2];
    if (any_pbc && cell_setup(cell, inv, height) != 0) return -2;
    int nimg[3] = {0, 0, 0};
    int32_t *wrap = (int32_t *)calloc(3 * (size_t)(n > 0 ? n : 1), sizeof(int32_t));
    double *wp = (double *)malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
    for (int a = 0; a < 3; ++a)
        if (pbc[a]) nimg[a] = (int)floor(cutoff / height[a]) + 1;
    for (int i = 0; i < n; ++i) {
        for (int x = 0; x < 3; ++x) wp[3 * i + x] = pos[3 * i + x];
        for (int a = 0; a < 3; ++a) {
            if (!pbc[a]) continue;
            double f = dot3(inv + 3 * a, pos + 3 * i);
            int32_t wv = (int32_t)floor(f);
            wrap[3 * i + a] = wv;
            for (int x = 0; x < 3; ++x) wp[3 * i + x] -= wv * cell[3 * a + x];
        }
    }
    const double rc2 = cutoff * cutoff;
    int64_t cnt = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double base[3] = {wp[3 * j] - wp[3 * i], wp[3 * j + 1] - wp[3 * i + 1], wp[3 * j + 2] - wp[3 * i + 2]};
            /* true shift S = S' + wrap_i - wrap_j ; iterate S in lexicographic order */
            int off[3];
            for (int a = 0; a < 3; ++a) off[a] = wrap[3 * i + a] - wrap[3 * j + a];
            for (int s0 = -nimg[0] + off[0]; s0 <= nimg[0] + off[0]; ++s0)
                for (int s1 = -nimg[1] + off[1]; s1 <= nimg[1] + off[1]; ++s1)
                    for (int s2 = -nimg[2] + off[2]; s2 <= nimg[2] + off[2]; ++s2) {
                        if (i == j && s0 == 0 && s1 == 0 && s2 == 0) continue;
                        int p0 = s0 - off[0], p1 = s1 - off[1], p2 = s2 - off[2];
                        double r[3];
                        for (int x = 0; x < 3; ++x)
                            r[x] = base[x] + p0 * cell[x] + p1 * cell[3 + x] + p2 * cell[6 + x];
                        double d2 = dot3(r, r);
                        if (d2 > rc2 || d2 <= 0.0) continue;
                        if (cnt < cap) {
                            ei[cnt] = i; ej[cnt] = j;
                            if (eS) { eS[3 * cnt] = s0; eS[3 * cnt + 1] = s1; eS[3 * cnt + 2] = s2; }
                            if (er) { er[3 * cnt] = r[0]; er[3 * cnt + 1] = r[1]; er[3 * cnt + 2] = r[2]; }
                        }
                        ++cnt;
                    }
        }
    free(wrap); free(wp);
    return cnt;
}

static int build_edges(int32_t n, const double *pos, const double cell[9], const uint8_t pbc[3],
                       double cutoff, int64_t *E_out, int32_t **ei, int32_t **ej, int32_t **eS, double **er) {
    int64_t cap = (int64_t)n * 96 + 64;
    for (;;) {
        *ei = (int32_t *)malloc(sizeof(int32_t) * cap);
        *ej = (int32_t *)malloc(sizeof(int32_t) * cap);
        *eS = (int32_t *)malloc(sizeof(int32_t) * 3 * cap);
        *er = (double *)malloc(sizeof(double) * 3 * cap);
        int64_t E = orc_neighbors(n, pos, cell, pbc, cutoff, cap, *ei, *ej, *eS, *er);
        if (E < 0) { free(*ei); free(*ej); free(*eS); free(*er); return (int)E; }
        if (E <= cap) { *E_out = E; return 0; }
        free(*ei); free(*ej); free(*eS); free(*er);
        cap = E;
    }
}

int orc_painn_eval(int real_bits, const float *blob, int64_t blob_len, const orc_painn_hparams *hp,
                   int32_t n, const int32_t *Z, const double *pos, const double cell[9],
                   const uint8_t pbc[3], double *energy, double *grad, const orc_painn_dump *dump) {
    int32_t *ei, *ej, *eS; double *er; int64_t E;
    int rc = build_edges(n, pos, cell, pbc, (double)hp->cutoff, &E, &ei, &ej, &eS, &er);
    if (rc) return rc;
    if (real_bits == 32) {
        /* nff holds float32 positions (nxyz) and float32 offsets = S.cell: redo r_e in float */
        for (int64_t e = 0; e < E; ++e)
            for (int x = 0; x < 3; ++x) {
                float off = (float)(eS[3 * e] * cell[x] + eS[3 * e + 1] * cell[3 + x] + eS[3 * e + 2] * cell[6 + x]);
                float r = (float)pos[3 * ej[e] + x] - (float)pos[3 * ei[e] + x] + off;
                er[3 * e + x] = (double)r;
            }
        model_t_f32 M;
        rc = model_bind_f32(&M, blob, blob_len, hp);
        if (!rc) rc = painn_run_f32(&M, hp, n, Z, E, ei, ej, er, energy, grad, dump);
    } else {
        model_t_f64 M;
        rc = model_bind_f64(&M, blob, blob_len, hp);
        if (!rc) rc = painn_run_f64(&M, hp, n, Z, E, ei, ej, er, energy, grad, dump);
    }
    free(ei); free(ej); free(eS); free(er);
    return rc;
}

/* EnsembleNFF.calculate (nff/io/ase_calcs.py; reference call site calculators.py:484):
 * per-model energy/grad -> eV, + stoichiometric offset, mean and population std. */
int orc_ensemble_eval(int real_bits, int32_t n_models, const float *const *blobs, int64_t blob_len,
                      const orc_painn_hparams *hp, double model_units_per_ev,
                      const double *offset_per_z, double offset_const,
                      int32_t n, const int32_t *Z, const double *pos, const double cell[9],
                      const uint8_t pbc[3], double *e_mean, double *e_std,
                      double *f_mean, double *f_std, double *e_model) {
    if (n_models < 1) return -1;
    double *E = (double *)malloc(sizeof(double) * n_models);
    double *G = (double *)malloc(sizeof(double) * 3 * (size_t)n * n_models);
    double off = 0.0;
    if (offset_per_z) {
        for (int i = 0; i < n; ++i) off += offset_per_z[Z[i]];
        off += offset_const;
    }
    int rc = 0;
    for (int m = 0; m < n_models && !rc; ++m) {
        rc = orc_painn_eval(real_bits, blobs[m], blob_len, hp, n, Z, pos, cell, pbc, &E[m],
                            G + 3 * (size_t)n * m, NULL);
        E[m] = E[m] / model_units_per_ev + off;
        if (e_model) e_model[m] = E[m];
    }
    if (!rc) {
        double mu = 0;
        for (int m = 0; m < n_models; ++m) mu += E[m];
        mu /= n_models;
        double var = 0;
        for (int m = 0; m < n_models; ++m) var += (E[m] - mu) * (E[m] - mu);
        *e_mean = mu;
        if (e_std) *e_std = sqrt(var / n_models);
        for (size_t t = 0; t < 3 * (size_t)n; ++t) {
            double fm = 0;
            for (int m = 0; m < n_models; ++m) fm += -G[3 * (size_t)n * m + t] / model_units_per_ev;
            fm /= n_models;
            double fv = 0;
            for (int m = 0; m < n_models; ++m) {
                double f = -G[3 * (size_t)n * m + t] / model_units_per_ev;
                fv += (f - fm) * (f - fm);
            }
            if (f_mean) f_mean[t] = fm;
            if (f_std) f_std[t] = sqrt(fv / n_models);
        }
    }
    free(E); free(G);
    return rc;
}

/* ---- Tersoff (LAMMPS pair_tersoff.cpp semantics; SURVEY.md Appendix A, last paragraph) ---- */
typedef struct { double m, gamma, lam3, c, d, h, n, beta, lam2, B, R, D, lam1, A; } ters_p;

static double ters_fc(double r, const ters_p *p) {
    if (r < p->R - p->D) return 1.0;
    if (r > p->R + p->D) return 0.0;
    return 0.5 * (1.0 - sin(M_PI_2 * (r - p->R) / p->D));
}
static double ters_fc_d(double r, const ters_p *p) {
    if (r < p->R - p->D) return 0.0;
    if (r > p->R + p->D) return 0.0;
    return -(M_PI_4 / p->D) * cos(M_PI_2 * (r - p->R) / p->D);
}
static double ters_gijk(double cs, const ters_p *p) {
    double c2 = p->c * p->c, d2 = p->d * p->d, hc = p->h - cs;
    return p->gamma * (1.0 + c2 / d2 - c2 / (d2 + hc * hc));
}
static double ters_gijk_d(double cs, const ters_p *p) {
    double c2 = p->c * p->c, d2 = p->d * p->d, hc = p->h - cs;
    double den = d2 + hc * hc;
    return p->gamma * (-2.0 * c2 * hc) / (den * den);
}
static double ters_ex(double rij, double rik, const ters_p *p, double *dex_drij) {
    double arg = p->lam3 * (rij - rik), darg = p->lam3;
    if ((int)p->m == 3) { darg = 3.0 * p->lam3 * arg * arg; arg = arg * arg * arg; }
    double ex;
    if (arg > 69.0776) { ex = 1.e30; darg = 0; }
    else if (arg < -69.0776) { ex = 0.0; darg = 0; }
    else ex = exp(arg);
    *dex_drij = ex * darg; /* d/d rij ; d/d rik = -this */
    return ex;
}
static double ters_bij(double zeta, const ters_p *p, double *dbij) {
    double tmp = p->beta * zeta, n = p->n;
    double c1 = pow(2.0 * n * 1.0e-16, -1.0 / n), c2 = pow(2.0 * n * 1.0e-8, -1.0 / n);
    double c3 = 1.0 / c2, c4 = 1.0 / c1;
    if (tmp > c1) { *dbij = p->beta * -0.5 * pow(tmp, -1.5); return 1.0 / sqrt(tmp); }
    if (tmp > c2) {
        *dbij = p->beta * (-0.5 * pow(tmp, -1.5) * (1.0 - (1.0 + 1.0 / (2.0 * n)) * pow(tmp, -n)));
        return (1.0 - pow(tmp, -n) / (2.0 * n)) / sqrt(tmp);
    }
    if (tmp < c4) { *dbij = 0.0; return 1.0; }
    if (tmp < c3) { *dbij = -0.5 * p->beta * pow(tmp, n - 1.0); return 1.0 - pow(tmp, n) / (2.0 * n); }
    double tn = pow(tmp, n);
    *dbij = -0.5 * pow(1.0 + tn, -1.0 - (1.0 / (2.0 * n))) * tn / zeta;
    return pow(1.0 + tn, -1.0 / (2.0 * n));
}

int orc_tersoff_eval(int32_t nt, const double *params, int32_t n, const int32_t *type,
                     const double *pos, const double cell[9], const uint8_t pbc[3],
                     double *energy, double *e_atom, double *forces) {
    const ters_p *P = (const ters_p *)params;
    double cutmax = 0;
    for (int t = 0; t < nt * nt * nt; ++t)
        if (P[t].R + P[t].D > cutmax) cutmax = P[t].R + P[t].D;
    int32_t *ei, *ej, *eS; double *er; int64_t E;
    int rc = build_edges(n, pos, cell, pbc, cutmax, &E, &ei, &ej, &eS, &er);
    if (rc) return rc;
    /* CSR start per centre (edges are sorted by i) */
    int64_t *start = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    for (int64_t e = 0; e < E; ++e) start[ei[e] + 1]++;
    for (int i = 0; i < n; ++i) start[i + 1] += start[i];
    double *ea = (double *)calloc((size_t)n, sizeof(double));
    double *F = (double *)calloc(3 * (size_t)n, sizeof(double));
    double Etot = 0;
    for (int i = 0; i < n; ++i) {
        int ti = type[i];
        for (int64_t e = start[i]; e < start[i + 1]; ++e) {
            int j = ej[e], tj = type[j];
            const ters_p *pij = &P[(ti * nt + tj) * nt + tj];
            const double *rij = er + 3 * e;
            double r = sqrt(dot3(rij, rij));
            if (r > pij->R + pij->D) continue;
            double fc = ters_fc(r, pij), dfc = ters_fc_d(r, pij);
            double fR = pij->A * exp(-pij->lam1 * r), fA = -pij->B * exp(-pij->lam2 * r);
            /* zeta_ij */
            double zeta = 0;
            for (int64_t e2 = start[i]; e2 < start[i + 1]; ++e2) {
                if (e2 == e) continue;
                int k = ej[e2];
                const ters_p *pijk = &P[(ti * nt + tj) * nt + type[k]];
                const double *rik = er + 3 * e2;
                double r2 = sqrt(dot3(rik, rik));
                if (r2 > pijk->R + pijk->D) continue;
                double cs = dot3(rij, rik) / (r * r2), dex;
                zeta += ters_fc(r2, pijk) * ters_gijk(cs, pijk) * ters_ex(r, r2, pijk, &dex);
            }
            double dbij, bij = ters_bij(zeta, pij, &dbij);
            double vrep = 0.5 * fc * fR, vatt = 0.5 * fc * bij * fA;
            Etot += vrep + vatt;
            /* LAMMPS ev_tally: each (directed) term is split half/half between i and j */
            ea[i] += 0.5 * (vrep + vatt);
            ea[j] += 0.5 * (vrep + vatt);
            if (!forces) continue;
            /* d/d r_ij of the radial parts */
            double dV_dr = 0.5 * (dfc * (fR + bij * fA) + fc * (-pij->lam1 * fR - pij->lam2 * bij * fA));
            double pref = 0.5 * fc * fA * dbij; /* dV/dzeta */
            double gij[3] = {dV_dr * rij[0] / r, dV_dr * rij[1] / r, dV_dr * rij[2] / r}; /* dV/d rij_vec */
            for (int64_t e2 = start[i]; e2 < start[i + 1]; ++e2) {
                if (e2 == e) continue;
                int k = ej[e2];
                const ters_p *pijk = &P[(ti * nt + tj) * nt + type[k]];
                const double *rik = er + 3 * e2;
                double r2 = sqrt(dot3(rik, rik));
                if (r2 > pijk->R + pijk->D) continue;
                double cs = dot3(rij, rik) / (r * r2), dex;
                double fck = ters_fc(r2, pijk), dfck = ters_fc_d(r2, pijk);
                double g = ters_gijk(cs, pijk), dg = ters_gijk_d(cs, pijk);
                double ex = ters_ex(r, r2, pijk, &dex);
                double gik[3];
                for (int x = 0; x < 3; ++x) {
                    double dcs_drij = (rik[x] / r2 - cs * rij[x] / r) / r;
                    double dcs_drik = (rij[x] / r - cs * rik[x] / r2) / r2;
                    double dz_drij = fck * (dg * dcs_drij * ex + g * dex * rij[x] / r);
                    double dz_drik = dfck * rik[x] / r2 * g * ex + fck * (dg * dcs_drik * ex - g * dex * rik[x] / r2);
                    gij[x] += pref * dz_drij;
                    gik[x] = pref * dz_drik;
                }
                for (int x = 0; x < 3; ++x) { F[3 * k + x] -= gik[x]; F[3 * i + x] += gik[x]; }
            }
            for (int x = 0; x < 3; ++x) { F[3 * j + x] -= gij[x]; F[3 * i + x] += gij[x]; }
        }
    }
    *energy = Etot;
    if (e_atom) memcpy(e_atom, ea, sizeof(double) * (size_t)n);
    if (forces) memcpy(forces, F, sizeof(double) * 3 * (size_t)n);
    free(ea); free(F); free(start); free(ei); free(ej); free(eS); free(er);
    return 0;
}
