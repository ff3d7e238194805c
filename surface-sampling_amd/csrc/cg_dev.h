// cg_dev.h — device side of the LAMMPS-style conjugate-gradient minimiser (relax.hip), shared with the chain-resident minimiser
// (chain_min.hip): fp64 block reductions and the per-chain state machine.
#ifndef VSSR_CG_DEV_H
#define VSSR_CG_DEV_H
#include "vssr_internal.h"

namespace vssr {

// ---- shared block reductions (fp64) ---------------------------------------------------------------------------------------
// Wave-level butterflies (fixed order, every lane ends with the wave's result) + one exchange of the per-wave results through LDS:
// two barriers per reduction instead of ten (the optimizer kernels run ~15 .. 250 of them per step).  `red` holds >= 16 doubles.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline double block_sum(double v, double *red) {
    const int tid = threadIdx.x, nw = (blockDim.x + 63) >> 6;
    v = wave_sum_f64(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double r = red[0];
    for (int w = 1; w < nw; ++w) r += red[w];
    __syncthreads();
    return r;
}
__device__ inline double block_max(double v, double *red) {
    const int tid = threadIdx.x, nw = (blockDim.x + 63) >> 6;
    v = wave_max_f64(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double r = red[0];
    for (int w = 1; w < nw; ++w) r = fmax(r, red[w]);
    __syncthreads();
    return r;
}

// ---- LAMMPS min_style cg (Polak-Ribiere + quadratic line search) as a per-chain state machine ------------------------------
// One evaluation of the batch per driver iteration; every chain consumes it according to its own phase:
//   PH_START   forces at the current CG point are known: set up the line search (or the very first direction), move to the
//              first trial point x0 + alphamax h;
//   PH_TRIAL   energy / forces at x0 + alpha h: secant ("quadratic") projection -> PH_PROJ, or accept, or halve alpha;
//   PH_PROJ    energy at the projected x0 + alpha0 h: accept if it is below the start, else fall through to the backtrack test
//              of LAMMPS' loop (which, like LAMMPS, then compares the PROJECTED point's energy with the ideal decrease at alpha);
//   PH_RESET   the chain went back to x0 after a failed line search: this evaluation restores consistent results, then stop.
// Constants of min_linesearch.cpp: ALPHA_MAX 1, ALPHA_REDUCE 0.5, BACKTRACK_SLOPE 0.4, QUADRATIC_TOL 0.1, EMACH 1e-8,
// EPS_QUAD 1e-28; of min_cg.cpp: EPS_ENERGY 1e-8.  thermo normalisation off (forces / energies extensive).
enum { PH_START = 0, PH_TRIAL = 1, PH_PROJ = 2, PH_RESET = 3 };
struct CgState {
    int phase, started, niter, neval, reason, pending;   // reason != 0: finished; pending: reason to report after PH_RESET
    double gg, eprevious, eoriginal, alpha, alphaprev, alphamax, fhprev, engprev, fdothall, fh_trial;
};

// One step of chain b's state machine on the evaluation that has just been made (energy[b], forces of its atoms); every thread of
// the (256-thread) workgroup calls it; red: >= 16 doubles of LDS.  The lock-step driver (k_cg_step, relax.hip) and the chain-resident
// minimiser (chain_min.hip) run this same code.
__device__ __forceinline__ void cg_step_chain(int b, double *red, const int *__restrict__ cfg_start, const double *__restrict__ energy,
          const double *__restrict__ forces, const uint8_t *__restrict__ fixed, int max_iter, int max_eval, double etol,
          double ftol, double dmax, double *__restrict__ pos, double *__restrict__ x0all, double *__restrict__ hall,
          double *__restrict__ gall, CgState *__restrict__ st, unsigned char *__restrict__ active, int *__restrict__ n_active) {
    const int tid = threadIdx.x, nt = blockDim.x;
    CgState S = st[b];
    // every wave has its copy of the state before thread 0 can store the advanced one: several exits below (a back-tracked trial,
    // the reset paths, stop right behind PH_PROJ) store without a barrier in front of them, and a wave that loaded late would take
    // another branch than the rest of its workgroup and wait alone at a block_sum barrier (advisor r5)
    __syncthreads();
    if (S.reason) return;
    const int a0 = cfg_start[b], n = 3 * (cfg_start[b + 1] - a0);
    double *x = pos + 3 * (size_t)a0, *x0 = x0all + 3 * (size_t)a0, *h = hall + 3 * (size_t)a0, *g = gall + 3 * (size_t)a0;
    const double *fg = forces + 3 * (size_t)a0;
    const uint8_t *fx = fixed ? fixed + a0 : nullptr;
    auto F = [&](int k) -> double { return (fx && fx[k / 3]) ? 0.0 : fg[k]; };
    const double ecur = energy[b];
    // LAMMPS counts the evaluations of alpha_step() only (min.cpp zeroes neval after the setup evaluation) and tests
    // max_eval once per iteration, after a completed line search (min_cg.cpp iterate)
    if (S.started) S.neval += 1;
    S.started = 1;
    // every exit stores the state; `stop` also switches the chain off
    auto stop = [&](int reason) {
        if (tid == 0) { S.reason = reason; st[b] = S; active[b] = 0; }
    };
    auto keep_going = [&]() {
        if (tid == 0) { st[b] = S; atomicAdd(n_active, 1); }
    };
    auto move_to = [&](double alpha) {
        for (int k = tid; k < n; k += nt) x[k] = x0[k] + alpha * h[k];
    };
    auto reset_to_start = [&](int pending) {   // alpha_step(0.0, 0): back to x0, one more evaluation, then report `pending`
        move_to(0.0);
        S.phase = PH_RESET; S.pending = pending;
        keep_going();
    };
    // a new iteration of min_cg.cpp: line search along h from the current point (forces of this evaluation)
    auto start_linesearch = [&]() {
        S.niter += 1;
        S.eprevious = ecur;
        double c = 0.0;
        for (int k = tid; k < n; k += nt) c += F(k) * h[k];
        const double fh = block_sum(c, red);
        S.fdothall = fh;
        if (fh <= 0.0) { stop(5); return; }
        double hm = 0.0;
        for (int k = tid; k < n; k += nt) hm = fmax(hm, fabs(h[k]));
        hm = block_max(hm, red);
        if (hm == 0.0) { stop(6); return; }
        S.alphamax = fmin(1.0, dmax / hm);
        for (int k = tid; k < n; k += nt) x0[k] = x[k];
        __syncthreads();
        S.eoriginal = ecur; S.alpha = S.alphamax; S.alphaprev = 0.0; S.fhprev = fh; S.engprev = ecur;
        S.phase = PH_TRIAL;
        move_to(S.alpha);
        keep_going();
    };

    if (S.phase == PH_RESET) { stop(S.pending); return; }
    if (S.phase == PH_START) {   // min_cg.cpp setup: h = g = f, gg = f.f
        double a = 0.0;
        for (int k = tid; k < n; k += nt) { const double f = F(k); g[k] = f; h[k] = f; a += f * f; }
        S.gg = block_sum(a, red);
        if (max_iter <= 0) { stop(3); return; }
        start_linesearch();
        return;
    }
    bool line_done = false;
    double e_test = ecur, fh_test = S.fh_trial;
    if (S.phase == PH_TRIAL) {
        double c = 0.0;
        for (int k = tid; k < n; k += nt) c += F(k) * h[k];
        const double fh = block_sum(c, red);
        const double delfh = fh - S.fhprev;
        if (fabs(fh) < 1e-28 || fabs(delfh) < 1e-28) { reset_to_start(7); return; }
        const double relerr = fabs(1.0 - (0.5 * (S.alpha - S.alphaprev) * (fh + S.fhprev) + ecur) / S.engprev);
        const double alpha0 = S.alpha - (S.alpha - S.alphaprev) * fh / delfh;
        if (relerr <= 0.1 && alpha0 > 0.0 && alpha0 < S.alphamax) {   // secant projection: evaluate x0 + alpha0 h next
            S.fh_trial = fh;
            S.phase = PH_PROJ;
            move_to(alpha0);
            keep_going();
            return;
        }
        fh_test = fh;
    } else {   // PH_PROJ
        if (ecur - S.eoriginal < 1e-8) line_done = true;
    }
    if (!line_done) {   // backtracking test of the loop, on the energy of the last evaluated point
        const double de_ideal = -0.4 * S.alpha * S.fdothall;
        const double de = e_test - S.eoriginal;
        if (de <= de_ideal) line_done = true;
        else {
            S.fhprev = fh_test; S.engprev = e_test; S.alphaprev = S.alpha;
            S.alpha *= 0.5;
            if (S.alpha <= 0.0 || de_ideal >= -1e-8) { reset_to_start(8); return; }
            S.phase = PH_TRIAL;
            move_to(S.alpha);
            keep_going();
            return;
        }
    }
    // ---- line search succeeded: tolerances and the next direction (min_cg.cpp iterate) ----
    if (S.neval >= max_eval) { stop(4); return; }
    if (fabs(ecur - S.eprevious) < etol * 0.5 * (fabs(ecur) + fabs(S.eprevious) + 1e-8)) { stop(1); return; }
    double d0 = 0.0, d1 = 0.0;
    for (int k = tid; k < n; k += nt) { const double f = F(k); d0 += f * f; d1 += f * g[k]; }
    d0 = block_sum(d0, red);
    d1 = block_sum(d1, red);
    if (d0 < ftol * ftol) { stop(2); return; }
    double beta = fmax(0.0, (d0 - d1) / S.gg);
    if ((S.niter + 1) % max(n, 1) == 0) beta = 0.0;   // restart every ndof iterations
    S.gg = d0;
    double gh = 0.0;
    for (int k = tid; k < n; k += nt) { const double f = F(k); g[k] = f; h[k] = f + beta * h[k]; gh += f * h[k]; }
    gh = block_sum(gh, red);
    if (gh <= 0.0)
        for (int k = tid; k < n; k += nt) h[k] = g[k];
    __syncthreads();
    if (S.niter >= max_iter) { stop(3); return; }
    start_linesearch();
}


}  // namespace vssr
#endif
