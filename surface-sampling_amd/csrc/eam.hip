// eam.hip — embedded-atom energy / per-atom energy / forces on gfx950 (fp64), batched over independent configurations.
//
// Replaces the `lmp` subprocess behind LAMMPSRunSurfCalc with `pair_style eam` + a funcfl file (reference
// mcmc/calculators/calculators.py:755-811, mcmc/calculators/lammpsrun.py:309-469; potential mcmc/potentials/Cu_u3.eam;
// BASELINE configs[0], tests/test_Cu.py).  Semantics follow LAMMPS pair_eam for one funcfl element:
//   E = sum_i F(rho_i) + 1/2 sum_{i != j} phi(r_ij),  rho_i = sum_j rho(r_ij),  phi(r) = z2r(r) / r,
//   z2r = 27.2 * 0.529 * Z(r)^2 (Hartree * Bohr -> eV * A), all three functions as LAMMPS' cubic splines over the file's
//   grids (coefficients built at vssr_eam_create, see build_spline), rho beyond the table extrapolated linearly,
//   pe/atom = F(rho_i) + 1/2 sum_j phi.
// One thread owns one centre and walks its CSR row (padded multigraph of nbr.hip, cutoff = the file's cutoff):
// pass 1 densities and F'(rho_i); pass 2 energies and the force sum_slots [(F'_i + F'_j) rho'(r) + phi'(r)] r_hat --
// every pair is seen from both ends, so there is no scatter and no atomics.
#include "vssr_internal.h"

namespace vssr {

// spline row m (1-based like LAMMPS): [0..2] derivative coefficients, [3..6] value coefficients
__device__ inline void eam_eval(const double *__restrict__ spl, int n, double x, double rd, bool clamp_lo, double &val,
                                double &der) {
    double p = x * rd + 1.0;
    int m = (int)p;
    m = clamp_lo ? max(1, min(m, n - 1)) : min(m, n - 1);
    p -= m;
    p = fmin(p, 1.0);
    const double *c = spl + 7 * (size_t)m;
    val = ((c[3] * p + c[4]) * p + c[5]) * p + c[6];
    der = (c[0] * p + c[1]) * p + c[2];
}

__device__ inline void eam_edge(const double *__restrict__ wpos, const double *C, int i, int j, int packedS, double r[3]) {
    int s0 = (packedS & 255) - 128, s1 = ((packedS >> 8) & 255) - 128, s2 = ((packedS >> 16) & 255) - 128;
    for (int x = 0; x < 3; ++x)
        r[x] = wpos[3 * j + x] - wpos[3 * i + x] + s0 * C[x] + s1 * C[3 + x] + s2 * C[6 + x];
}

__global__ void k_eam_density(int N, vssr_eam_grid g, const double *__restrict__ frho, const double *__restrict__ rhor,
                              const int *__restrict__ atom_cfg, const double *__restrict__ cell,
                              const double *__restrict__ wpos, const int *__restrict__ row_start,
                              const float4 *__restrict__ edge, const int *__restrict__ edge_S,
                              const int *__restrict__ counters, double *__restrict__ e_embed, double *__restrict__ fp,
                              ActiveView av) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || counters[2] || !av.atom(i)) return;
    const double *C = cell + 9 * atom_cfg[i];
    double rho = 0.0;
    for (int e = row_start[i]; e < row_start[i + 1]; ++e) {
        const int j = __float_as_int(edge[e].w);
        if (j < 0) continue;
        double r[3];
        eam_edge(wpos, C, i, j, edge_S[e], r);
        const double d = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        if (d >= g.cutoff) continue;
        double v, dv;
        eam_eval(rhor, g.nr, d, 1.0 / g.dr, false, v, dv);
        rho += v;
    }
    double F, dF;
    eam_eval(frho, g.nrho, rho, 1.0 / g.drho, true, F, dF);
    const double rhomax = (g.nrho - 1) * g.drho;
    if (rho > rhomax) F += dF * (rho - rhomax);   // linear continuation beyond the table (pair_eam.cpp)
    e_embed[i] = F;
    fp[i] = dF;
}

__global__ void k_eam_force(int N, vssr_eam_grid g, const double *__restrict__ rhor, const double *__restrict__ z2r,
                            const int *__restrict__ atom_cfg, const double *__restrict__ cell,
                            const double *__restrict__ wpos, const int *__restrict__ row_start,
                            const float4 *__restrict__ edge, const int *__restrict__ edge_S,
                            const int *__restrict__ counters, const double *__restrict__ e_embed,
                            const double *__restrict__ fp, double *__restrict__ e_atom, double *__restrict__ forces,
                            ActiveView av) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || counters[2] || !av.atom(i)) return;
    const double *C = cell + 9 * atom_cfg[i];
    const double fpi = fp[i];
    double ea = e_embed[i], f0 = 0.0, f1 = 0.0, f2 = 0.0;
    for (int e = row_start[i]; e < row_start[i + 1]; ++e) {
        const int j = __float_as_int(edge[e].w);
        if (j < 0) continue;
        double r[3];
        eam_edge(wpos, C, i, j, edge_S[e], r);
        const double d = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        if (d >= g.cutoff) continue;
        double rh, drh, z, dz;
        eam_eval(rhor, g.nr, d, 1.0 / g.dr, false, rh, drh);
        eam_eval(z2r, g.nr, d, 1.0 / g.dr, false, z, dz);
        const double recip = 1.0 / d;
        const double phi = z * recip;
        const double phip = dz * recip - phi * recip;
        const double psip = (fpi + fp[j]) * drh + phip;     // dE / d r of this pair, seen from centre i
        ea += 0.5 * phi;
        const double s = psip * recip;                      // force on i = + psip * r_hat (r points from i to j)
        f0 += s * r[0]; f1 += s * r[1]; f2 += s * r[2];
    }
    e_atom[i] = ea;
    forces[3 * i] = f0; forces[3 * i + 1] = f1; forces[3 * i + 2] = f2;
}

__global__ void __launch_bounds__(256)
k_eam_energy(const int *__restrict__ cfg_start, const double *__restrict__ e_atom, double *__restrict__ energy,
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

// LAMMPS PairEAM::interpolate(): rows 1..n, [6] = f_m, [5] = finite-difference slope, [4], [3] = cubic through (f, slope) of
// m and m + 1, [2..0] = the derivative's coefficients / delta.  Row 0 is unused.
void eam_build_spline(const double *f, int n, double delta, double *spl /*[n + 1][7]*/) {
    auto S = [&](int m, int k) -> double & { return spl[7 * (size_t)m + k]; };
    for (int k = 0; k < 7; ++k) S(0, k) = 0.0;
    for (int m = 1; m <= n; ++m) S(m, 6) = f[m - 1];
    S(1, 5) = S(2, 6) - S(1, 6);
    S(2, 5) = 0.5 * (S(3, 6) - S(1, 6));
    S(n - 1, 5) = 0.5 * (S(n, 6) - S(n - 2, 6));
    S(n, 5) = S(n, 6) - S(n - 1, 6);
    for (int m = 3; m <= n - 2; ++m) S(m, 5) = ((S(m - 2, 6) - S(m + 2, 6)) + 8.0 * (S(m + 1, 6) - S(m - 1, 6))) / 12.0;
    for (int m = 1; m <= n - 1; ++m) {
        S(m, 4) = 3.0 * (S(m + 1, 6) - S(m, 6)) - 2.0 * S(m, 5) - S(m + 1, 5);
        S(m, 3) = S(m, 5) + S(m + 1, 5) - 2.0 * (S(m + 1, 6) - S(m, 6));
    }
    S(n, 4) = 0.0;
    S(n, 3) = 0.0;
    for (int m = 1; m <= n; ++m) {
        S(m, 2) = S(m, 5) / delta;
        S(m, 1) = 2.0 * S(m, 4) / delta;
        S(m, 0) = 3.0 * S(m, 3) / delta;
    }
}

int eam_run(vssr_handle *h, uint32_t want) {
    (void)want;
    const int N = h->n_atoms;
    hipStream_t st = h->stream;
    int rc = build_neighbors(h, h->eam_grid.cutoff);
    if (rc) return rc;
    if (h->d_ters_e.ensure(sizeof(double) * h->n_cfg) || h->d_ters_ea.ensure(sizeof(double) * N) ||
        h->d_ters_f.ensure(sizeof(double) * 3 * N) || h->d_gbar.ensure(sizeof(double) * 2 * (size_t)N))
        return set_err(h, VSSR_E_NOMEM, "EAM buffers: out of device memory");
    double *e_embed = h->d_gbar.as<double>(), *fp = e_embed + N;
    const double *frho = h->ters_params.as<double>();
    const double *rhor = frho + 7 * (size_t)(h->eam_grid.nrho + 1);
    const double *z2r = rhor + 7 * (size_t)(h->eam_grid.nr + 1);
    h->prof.begin(KC_TERSOFF, st);
    dim3 blk(64), grd((N + 63) / 64);
    const ActiveView av{h->active_mask, h->d_atom_cfg.as<int>()};
    hipLaunchKernelGGL(k_eam_density, grd, blk, 0, st, N, h->eam_grid, frho, rhor, h->d_atom_cfg.as<int>(),
                       h->d_cell.as<double>(), h->d_wpos.as<double>(), h->d_row_start.as<int>(), h->d_edge.as<float4>(),
                       h->d_edge_S.as<int>(), h->d_counters.as<int>(), e_embed, fp, av);
    hipLaunchKernelGGL(k_eam_force, grd, blk, 0, st, N, h->eam_grid, rhor, z2r, h->d_atom_cfg.as<int>(),
                       h->d_cell.as<double>(), h->d_wpos.as<double>(), h->d_row_start.as<int>(), h->d_edge.as<float4>(),
                       h->d_edge_S.as<int>(), h->d_counters.as<int>(), e_embed, fp, h->d_ters_ea.as<double>(),
                       h->d_ters_f.as<double>(), av);
    hipLaunchKernelGGL(k_eam_energy, dim3(h->n_cfg), dim3(256), 0, st, h->d_cfg_start.as<int>(),
                       h->d_ters_ea.as<double>(), h->d_ters_e.as<double>(), h->active_mask);
    h->prof.end(st);
    VSSR_HIP(h, hipGetLastError());
    return VSSR_OK;
}

}  // namespace vssr
