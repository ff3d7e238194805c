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
// The site energy E_i depends only on the vectors r_ij from i to its neighbors, so a centre is self-contained:
// G_slot = dE_i/d r_ij is written for its own slots, and the force on atom c is gathered as
// sum_{slots of c} (G[slot] - G[rev[slot]]) — no atomics, deterministic.
//
// k_tersoff_site4 (rows of <= TS_MAXD slots, <= 4 species: every row of the GaN workloads): FOUR lanes per centre.  The
// neighborhood (unit vectors, distances, types) is written to LDS once, the parameter entries sit in LDS next to it, and
// every loop of the three-body sums runs from LDS -- the first form of this kernel (k_tersoff_site, one thread per centre,
// kept for longer rows / more species) chased edge -> type -> parameters -> position through global memory for every (j, k)
// and accumulated dE/dr_ik with global read-modify-writes: 882 us per evaluation of 4 096 x 48 atoms, all of it latency
// (profiles/r04/NOTES_tersoff.md).  A lane owns the slots n = q, q + 4, ... of its centre in both passes:
//   pass 1 (slot n as j): zeta_n over all k, b_ij, the pair energy, pref_n = 1/2 fc fA db/dzeta -> LDS, and the slot's own part of
//                          G_n = dV/dr_n rhat_n + pref_n sum_m dzeta_nm/dr_n  (sums collected in the zeta walk, scaled afterwards)
//   pass 2 (slot n as k): G_n += sum_m pref_m dzeta_mn/dr_n
// so every G is produced by one lane (stored by pass 1, completed by pass 2), without atomics.
#include "tersoff_dev.h"

namespace vssr {

__global__ void __launch_bounds__(64) k_tersoff_site(int N, int nt, const TersP *__restrict__ P, const int *__restrict__ type,
                               const int *__restrict__ atom_cfg, const double *__restrict__ cell,
                               const double *__restrict__ wpos, const int *__restrict__ row_start,
                               const float4 *__restrict__ edge, const int *__restrict__ edge_S,
                               const int *__restrict__ counters, double *__restrict__ eps /*[slots]*/,
                               double *__restrict__ gslot /*[slots][3]*/, ActiveView av, int longer_than) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || counters[2] || !av.atom(i)) return;
    tersoff_site_atom(i, nt, P, type, atom_cfg, cell, wpos, row_start, edge, edge_S, eps, gslot, longer_than);
}

__global__ void __launch_bounds__(TS_CENTRES * TS_LANES)
k_tersoff_site4(int N, int nt, const TersP *__restrict__ P, const int *__restrict__ type,
                const int *__restrict__ atom_cfg, const double *__restrict__ cell,
                const double *__restrict__ wpos, const int *__restrict__ row_start,
                const float4 *__restrict__ edge, const int *__restrict__ edge_S,
                const int *__restrict__ counters, double *__restrict__ eps /*[slots]*/,
                double *__restrict__ gslot /*[slots][3]*/, ActiveView av) {
    __shared__ TersShared sh;
    if (counters[2]) return;   // (uniform)
    const int i = blockIdx.x * TS_CENTRES + (threadIdx.x >> 2);
    tersoff_derive_params(sh, nt, P);
    tersoff_site4_tile(sh, i, i < N && av.atom(i), nt, type, atom_cfg, cell, wpos, row_start, edge, edge_S, eps, gslot);
}

__global__ void k_tersoff_gather(int N, const int *__restrict__ row_start, const int *__restrict__ rev,
                                 const int *__restrict__ counters, const double *__restrict__ eps,
                                 const double *__restrict__ gslot, double *__restrict__ e_atom,
                                 double *__restrict__ forces, ActiveView av) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N || counters[2] || !av.atom(c)) return;
    tersoff_gather_atom(c, row_start, rev, eps, gslot, e_atom, forces);
}

__global__ void __launch_bounds__(256)
k_tersoff_energy(const int *__restrict__ cfg_start, const double *__restrict__ e_atom, double *__restrict__ energy,
                 const unsigned char *__restrict__ active) {
    __shared__ double red[256];
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    tersoff_chain_energy(b, red, cfg_start, e_atom, energy);
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
    // rows of <= TS_MAXD slots: four lanes per centre from LDS; longer rows (and potentials of more than 4 species): one thread per
    // centre.
    const bool fast = h->n_types * h->n_types * h->n_types <= TS_MAXP;
    if (fast)
        hipLaunchKernelGGL(k_tersoff_site4, dim3((N + TS_CENTRES - 1) / TS_CENTRES), dim3(TS_CENTRES * TS_LANES), 0, st, N, h->n_types,
                           h->ters_params.as<TersP>(), h->d_Z.as<int>(), h->d_atom_cfg.as<int>(), h->d_cell.as<double>(),
                           h->d_wpos.as<double>(), h->d_row_start.as<int>(), h->d_edge.as<float4>(), h->d_edge_S.as<int>(),
                           h->d_counters.as<int>(), eps, gslot, av);
    hipLaunchKernelGGL(k_tersoff_site, grd, blk, 0, st, N, h->n_types, h->ters_params.as<TersP>(), h->d_Z.as<int>(),
                       h->d_atom_cfg.as<int>(), h->d_cell.as<double>(), h->d_wpos.as<double>(),
                       h->d_row_start.as<int>(), h->d_edge.as<float4>(), h->d_edge_S.as<int>(),
                       h->d_counters.as<int>(), eps, gslot, av, fast ? TS_MAXD : -1);
    hipLaunchKernelGGL(k_tersoff_gather, grd, blk, 0, st, N, h->d_row_start.as<int>(), h->d_rev.as<int>(),
                       h->d_counters.as<int>(), eps, gslot, h->d_ters_ea.as<double>(), h->d_ters_f.as<double>(), av);
    hipLaunchKernelGGL(k_tersoff_energy, dim3(h->n_cfg), dim3(256), 0, st, h->d_cfg_start.as<int>(),
                       h->d_ters_ea.as<double>(), h->d_ters_e.as<double>(), h->active_mask);
    h->prof.end(st);
    VSSR_HIP(h, hipGetLastError());
    return VSSR_OK;
}

}  // namespace vssr
