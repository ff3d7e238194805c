// relax.hip — lock-step relaxation of all resident chains on the device: FIRE and BFGS.
//
// Counterpart of optimize_slab (reference mcmc/dynamics.py:83-170):
//   dyn = Optimizer(slab); dyn.run(steps=relax_steps, fmax=0.01)
// with Optimizer = ase.optimize.BFGS for the reference's SrTiO3 configuration (scripts/configs/sample_config_painn.json:26,
// dispatch mcmc/dynamics.py:119-127) and ase.optimize.FIRE as the default.  FixAtoms (reference mcmc/system.py:288-294)
// enters as a per-atom mask that zeroes the force.  The reference drives one structure from Python; here every chain
// carries its own optimizer state on the device and all chains step together: one energy+force evaluation of the batch
// per iteration, positions and forces never leave HBM.  A chain whose max |F_i| drops below fmax freezes (ASE's
// convergence test, evaluated before each step) and is marked inactive: the evaluation kernels skip inactive chains, so an
// iteration costs what its unconverged chains cost.
//
// FIRE (ase/optimize/fire.py; defaults dt=0.1, maxstep=0.2, dtmax=1.0, Nmin=5, finc=1.1, fdec=0.5, astart=0.1, fa=0.99)
// is restated per chain with (velocity, dt, a, Nsteps) state.
//
// BFGS (ase/optimize/bfgs.py; alpha = 70, maxstep = 0.2): H0 = alpha I; per step the rank-2 update
//   H <- H - df df^T / (dr.df) - (H dr)(H dr)^T / (dr.H dr)        (skipped when max|dr| < 1e-7)
// and the step dr = V |W|^-1 V^T f from the eigen-decomposition H = V W V^T, rescaled so that the longest atomic
// displacement is at most maxstep.  ASE holds the dense 3N x 3N matrix; here the SAME matrix is held in factored form.
// After k updates H = alpha I + Q B Q^T exactly, where the m <= 2k orthonormal columns of Q span the update vectors
// (df_i, H_i dr_i) and B is m x m: H dr costs O(N m), the eigen-decomposition is that of the m x m matrix alpha I + B
// (every direction outside span(Q) keeps the eigenvalue alpha), and
//   |H|^-1 f = (f - Q Q^T f) / alpha + Q Y |L|^-1 Y^T Q^T f ,   alpha I + B = Y L Y^T.
// No 3N x 3N storage and no large eigensolve for any number of free atoms; the m x m problem (m <= 2 relax_steps) is
// solved by cyclic Jacobi rotations in LDS, fp64.  Pinned against the reference's stored BFGS traces through
// tests/bfgs_oracle.py (dense restatement of ASE's algorithm) -- tests/test_bfgs.py.
#include "cg_dev.h"

namespace vssr {

// chain b has converged: freeze it and take it out of the evaluation
__device__ inline void mark_converged(int b, int *converged, unsigned char *active) {
    *converged = 1;
    active[b] = 0;
}

// ---- FIRE ---------------------------------------------------------------------------------------------------------------
struct FireState {   // per chain
    double dt, a;
    int nsteps_pos;  // steps since the last power <= 0 (ASE's Nsteps)
    int has_v;       // 0 until the first step (ASE: self.v is None)
    int steps;       // optimizer steps taken
    int converged;
};

__global__ void k_fire_init(int B, double dt0, double astart, FireState *__restrict__ st, unsigned char *__restrict__ active) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    st[b].dt = dt0; st[b].a = astart; st[b].nsteps_pos = 0; st[b].has_v = 0; st[b].steps = 0; st[b].converged = 0;
    active[b] = 1;
}

// One workgroup per chain: convergence test on the forces of the CURRENT positions, then one FIRE step.  Nothing moves
// when the neighbor capacity overflowed in the evaluation that produced `forces` (counters[2]): the host regrows the
// buffers, repeats the evaluation and launches this kernel again.
__global__ void __launch_bounds__(256)
k_fire_step(const int *__restrict__ cfg_start, const int *__restrict__ counters, const float *__restrict__ forces,
            const uint8_t *__restrict__ fixed, double fmax_tol, double maxstep, double dtmax, double finc, double fdec,
            double astart, double fa, int nmin, int max_steps, double *__restrict__ pos, double *__restrict__ vel,
            FireState *__restrict__ st, unsigned char *__restrict__ active, int *__restrict__ n_active) {
    __shared__ double red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (counters[2]) return;
    const int a0 = cfg_start[b], a1 = cfg_start[b + 1];
    FireState S = st[b];
    if (S.converged) return;
    // max_i |F_i| over unconstrained atoms
    double fm2 = 0.0;
    for (int i = a0 + tid; i < a1; i += blockDim.x) {
        if (fixed && fixed[i]) continue;
        double fx = forces[3 * i], fy = forces[3 * i + 1], fz = forces[3 * i + 2];
        fm2 = fmax(fm2, fx * fx + fy * fy + fz * fz);
    }
    fm2 = block_max(fm2, red);
    if (!(fm2 >= fmax_tol * fmax_tol)) {   // converged (a NaN force also stops the chain: nothing sane to follow)
        if (tid == 0) mark_converged(b, &st[b].converged, active);
        return;
    }
    if (S.steps >= max_steps) return;   // (iterations repeated after a capacity regrow never exceed relax_steps)
    double vf = 0.0, ff = 0.0, vv = 0.0;
    for (int i = a0 + tid; i < a1; i += blockDim.x) {
        const bool fx_ = fixed && fixed[i];
        for (int x = 0; x < 3; ++x) {
            double f = fx_ ? 0.0 : (double)forces[3 * i + x], v = S.has_v ? vel[3 * i + x] : 0.0;
            vf += f * v; ff += f * f; vv += v * v;
        }
    }
    vf = block_sum(vf, red);
    ff = block_sum(ff, red);
    vv = block_sum(vv, red);
    double mix_a = 0.0, mix_s = 0.0;
    bool zero_v = false;
    if (S.has_v) {
        if (vf > 0.0) {
            mix_a = S.a;
            mix_s = S.a * sqrt(vv / ff);   // v = (1 - a) v + a f |v| / |f|
            if (S.nsteps_pos > nmin) { S.dt = fmin(S.dt * finc, dtmax); S.a *= fa; }
            S.nsteps_pos += 1;
        } else {
            zero_v = true;
            S.a = astart; S.dt *= fdec; S.nsteps_pos = 0;
        }
    }
    // v <- mix(v, f) + dt f ; dr = dt v
    double dr2 = 0.0;
    for (int i = a0 + tid; i < a1; i += blockDim.x) {
        const bool fx_ = fixed && fixed[i];
        for (int x = 0; x < 3; ++x) {
            double f = fx_ ? 0.0 : (double)forces[3 * i + x];
            double v = (S.has_v && !zero_v) ? vel[3 * i + x] : 0.0;
            if (S.has_v && !zero_v) v = (1.0 - mix_a) * v + mix_s * f;
            v += S.dt * f;
            vel[3 * i + x] = v;
            dr2 += (S.dt * v) * (S.dt * v);
        }
    }
    dr2 = block_sum(dr2, red);
    const double normdr = sqrt(dr2);
    const double scale = normdr > maxstep ? maxstep / normdr : 1.0;
    for (int i = a0 + tid; i < a1; i += blockDim.x)
        for (int x = 0; x < 3; ++x) pos[3 * i + x] += scale * S.dt * vel[3 * i + x];
    if (tid == 0) {
        S.has_v = 1;
        S.steps += 1;
        st[b] = S;
        atomicAdd(n_active, 1);
    }
}

// ---- BFGS ---------------------------------------------------------------------------------------------------------------
struct BfgsState {   // per chain
    int m;           // columns of Q
    int has_prev;    // r0 / f0 hold the previous step's positions / forces
    int steps;
    int converged;
};

__global__ void k_bfgs_init(int B, BfgsState *__restrict__ st, unsigned char *__restrict__ active) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    st[b].m = 0; st[b].has_prev = 0; st[b].steps = 0; st[b].converged = 0;
    active[b] = 1;
}

// Eigen-decomposition of the symmetric m x m matrix A (LDS, row stride ld): cyclic Jacobi with the round-robin parallel
// ordering -- m / 2 disjoint rotations per round, m - 1 rounds per sweep.  On return diag(A) = eigenvalues and the columns
// of Y the eigenvectors (A_in = Y diag Y^T).  cs: 2 * (m / 2 + 1) doubles of scratch.  All threads of the block take part.
__device__ void jacobi_eigh(double *A, double *Y, int m, int ld, double *cs, double *red) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int t = tid; t < m * m; t += nt) Y[(t / m) * ld + t % m] = (t / m == t % m) ? 1.0 : 0.0;
    __syncthreads();
    if (m < 2) return;
    const int me = m + (m & 1);          // players of the tournament (a dummy player when m is odd)
    const int half = me / 2;
    for (int sweep = 0; sweep < 30; ++sweep) {
        // off-diagonal weight against the diagonal
        double off = 0.0, dg = 0.0;
        for (int t = tid; t < m * m; t += nt) {
            const int p = t / m, q = t % m;
            const double v = A[p * ld + q];
            if (p == q) dg += v * v; else off += v * v;
        }
        off = block_sum(off, red);
        dg = block_sum(dg, red);
        if (off <= 1e-30 * dg || off == 0.0) break;
        for (int r = 0; r < me - 1; ++r) {
            // rotation angles of this round's pairs
            if (tid < half) {
                int p = tid == 0 ? me - 1 : (r + tid) % (me - 1);
                int q = tid == 0 ? r : (r - tid + (me - 1)) % (me - 1);
                if (p > q) { const int t2 = p; p = q; q = t2; }
                double c = 1.0, s = 0.0;
                if (q < m) {
                    const double apq = A[p * ld + q];
                    if (apq != 0.0) {
                        const double tau = (A[q * ld + q] - A[p * ld + p]) / (2.0 * apq);
                        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t);
                        s = t * c;
                    }
                }
                cs[2 * tid] = c; cs[2 * tid + 1] = s;
            }
            __syncthreads();
            // columns p, q of A and of Y:  (x_p, x_q) <- (c x_p - s x_q, s x_p + c x_q) for every row
            for (int t = tid; t < half * m; t += nt) {
                const int i = t / m, k = t % m;
                int p = i == 0 ? me - 1 : (r + i) % (me - 1);
                int q = i == 0 ? r : (r - i + (me - 1)) % (me - 1);
                if (p > q) { const int t2 = p; p = q; q = t2; }
                if (q >= m) continue;
                const double c = cs[2 * i], s = cs[2 * i + 1];
                const double ap = A[k * ld + p], aq = A[k * ld + q];
                A[k * ld + p] = c * ap - s * aq; A[k * ld + q] = s * ap + c * aq;
                const double yp = Y[k * ld + p], yq = Y[k * ld + q];
                Y[k * ld + p] = c * yp - s * yq; Y[k * ld + q] = s * yp + c * yq;
            }
            __syncthreads();
            // rows p, q of A
            for (int t = tid; t < half * m; t += nt) {
                const int i = t / m, k = t % m;
                int p = i == 0 ? me - 1 : (r + i) % (me - 1);
                int q = i == 0 ? r : (r - i + (me - 1)) % (me - 1);
                if (p > q) { const int t2 = p; p = q; q = t2; }
                if (q >= m) continue;
                const double c = cs[2 * i], s = cs[2 * i + 1];
                const double ap = A[p * ld + k], aq = A[q * ld + k];
                A[p * ld + k] = c * ap - s * aq; A[q * ld + k] = s * ap + c * aq;
            }
            __syncthreads();
        }
    }
}

// One workgroup per chain.  Vectors are over ALL 3 N_b coordinates of the chain with the entries of held atoms zero (their
// force is zeroed and they never move), which is exactly ASE's block structure.  Q: [mmax][3 N_b] rows (basis vectors),
// Bm: [mmax][mmax] (row stride mmax).
__global__ void __launch_bounds__(256)
k_bfgs_step(const int *__restrict__ cfg_start, const int *__restrict__ counters, const float *__restrict__ forces,
            const uint8_t *__restrict__ fixed, double fmax_tol, double alpha, double maxstep, int mmax, int max_steps,
            double *__restrict__ pos, double *__restrict__ r0, double *__restrict__ f0, double *__restrict__ Qall,
            double *__restrict__ Ball, double *__restrict__ work /*[3 sum N] step vector*/, BfgsState *__restrict__ st,
            unsigned char *__restrict__ active, int *__restrict__ n_active) {
    extern __shared__ double lds[];
    __shared__ double red[256];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nt >> 6;
    if (counters[2]) return;
    const int a0 = cfg_start[b], a1 = cfg_start[b + 1];
    const int n = 3 * (a1 - a0);
    BfgsState S = st[b];
    if (S.converged) return;
    double *Q = Qall + (size_t)3 * a0 * mmax;          // rows of length n
    double *Bm = Ball + (size_t)b * mmax * mmax;
    double *x = pos + 3 * (size_t)a0, *xr0 = r0 + 3 * (size_t)a0, *xf0 = f0 + 3 * (size_t)a0, *w = work + 3 * (size_t)a0;
    const float *fg = forces + 3 * (size_t)a0;
    const uint8_t *fx = fixed ? fixed + a0 : nullptr;
    auto F = [&](int k) -> double { return (fx && fx[k / 3]) ? 0.0 : (double)fg[k]; };
    // LDS carve-up
    const int ld = mmax | 1;                             // odd row stride: no bank conflicts on column walks
    double *A = lds, *Y = A + (size_t)mmax * ld, *c1 = Y + (size_t)mmax * ld, *c2 = c1 + mmax, *cf = c2 + mmax,
           *tv = cf + mmax, *cs = tv + mmax;             // cs: mmax + 2

    // ---- convergence: max_i |F_i|^2 < fmax^2 (ase/optimize/optimize.py converged()) ----------------------------------------
    double fm2 = 0.0;
    for (int i = tid; i < a1 - a0; i += nt) {
        const double f0_ = F(3 * i), f1_ = F(3 * i + 1), f2_ = F(3 * i + 2);
        fm2 = fmax(fm2, f0_ * f0_ + f1_ * f1_ + f2_ * f2_);
    }
    fm2 = block_max(fm2, red);
    if (!(fm2 >= fmax_tol * fmax_tol)) {
        if (tid == 0) mark_converged(b, &st[b].converged, active);
        return;
    }
    if (S.steps >= max_steps) return;
    int m = S.m;
    // ---- update of H from the previous step (ase/optimize/bfgs.py update()) -------------------------------------------------
    if (S.has_prev) {
        double mx = 0.0;
        for (int k = tid; k < n; k += nt) mx = fmax(mx, fabs(x[k] - xr0[k]));
        mx = block_max(mx, red);
        if (mx >= 1e-7 && m + 2 <= mmax) {
            // a = dr.df ; c1 = Q^T dr
            double a = 0.0;
            for (int k = tid; k < n; k += nt) a += (x[k] - xr0[k]) * (F(k) - xf0[k]);
            a = block_sum(a, red);
            for (int j = wave; j < m; j += nwaves) {   // one wave per basis vector: no block barrier per dot product
                double d = 0.0;
                for (int k = lane; k < n; k += 64) d += Q[(size_t)j * n + k] * (x[k] - xr0[k]);
                d = wave_sum_f64(d);
                if (lane == 0) c1[j] = d;
            }
            __syncthreads();
            // tv = B c1 ; dg = alpha dr + Q tv -> w ; b = dr.dg
            for (int j = tid; j < m; j += nt) {
                double d = 0.0;
                for (int i = 0; i < m; ++i) d += Bm[(size_t)j * mmax + i] * c1[i];
                tv[j] = d;
            }
            __syncthreads();
            double bb = 0.0;
            for (int k = tid; k < n; k += nt) {
                const double dr = x[k] - xr0[k];
                double g = alpha * dr;
                for (int j = 0; j < m; ++j) g += Q[(size_t)j * n + k] * tv[j];
                w[k] = g;
                bb += dr * g;
            }
            bb = block_sum(bb, red);
            // append the two update vectors to the basis: coefficients c1 (df), c2 (dg) in the extended basis.
            // Two Gram-Schmidt passes ("twice is enough"); a vector already inside span(Q) adds no column.
            for (int which = 0; which < 2; ++which) {
                double *cc = which == 0 ? c1 : c2;
                double *qn = Q + (size_t)m * n;            // candidate column
                double nrm0 = 0.0;
                for (int k = tid; k < n; k += nt) {
                    const double v = which == 0 ? F(k) - xf0[k] : w[k];
                    qn[k] = v;
                    nrm0 += v * v;
                }
                nrm0 = block_sum(nrm0, red);
                for (int j = tid; j < mmax; j += nt) cc[j] = 0.0;
                __syncthreads();
                // classical Gram-Schmidt, twice ("twice is enough"): all m projections of a pass at once (a wave per basis vector),
                // then one sweep over the candidate -- 2 x 2 barriers instead of 2 m block reductions (it was the modified form,
                // sequential in j: ~170 of the kernel's ~250 reductions at m = 42)
                for (int pass = 0; pass < 2; ++pass) {
                    for (int j = wave; j < m; j += nwaves) {
                        double d = 0.0;
                        for (int k = lane; k < n; k += 64) d += Q[(size_t)j * n + k] * qn[k];
                        d = wave_sum_f64(d);
                        if (lane == 0) tv[j] = d;
                    }
                    __syncthreads();
                    for (int k = tid; k < n; k += nt) {
                        double v = qn[k];
                        for (int j = 0; j < m; ++j) v -= tv[j] * Q[(size_t)j * n + k];
                        qn[k] = v;
                    }
                    for (int j = tid; j < m; j += nt) cc[j] += tv[j];
                    __syncthreads();
                }
                double nrm = 0.0;
                for (int k = tid; k < n; k += nt) nrm += qn[k] * qn[k];
                nrm = block_sum(nrm, red);
                if (nrm > 1e-24 * nrm0 && nrm > 0.0) {
                    const double inv = 1.0 / sqrt(nrm);
                    for (int k = tid; k < n; k += nt) qn[k] *= inv;
                    if (tid == 0) cc[m] = sqrt(nrm);
                    // new row / column of B (zeros)
                    for (int j = tid; j <= m; j += nt) { Bm[(size_t)m * mmax + j] = 0.0; Bm[(size_t)j * mmax + m] = 0.0; }
                    ++m;
                }
                __syncthreads();
            }
            // B -= c1 c1^T / a + c2 c2^T / b
            for (int t = tid; t < m * m; t += nt) {
                const int i = t / m, j = t % m;
                Bm[(size_t)i * mmax + j] -= c1[i] * c1[j] / a + c2[i] * c2[j] / bb;
            }
            __syncthreads();
        }
    }
    // ---- step: dr = |H|^-1 f -------------------------------------------------------------------------------------------------
    for (int t = tid; t < m * m; t += nt) {
        const int i = t / m, j = t % m;
        A[i * ld + j] = Bm[(size_t)i * mmax + j] + (i == j ? alpha : 0.0);
    }
    __syncthreads();
    jacobi_eigh(A, Y, m, ld, cs, red);
    for (int j = wave; j < m; j += nwaves) {   // cf = Q^T f, a wave per basis vector
        double d = 0.0;
        for (int k = lane; k < n; k += 64) d += Q[(size_t)j * n + k] * F(k);
        d = wave_sum_f64(d);
        if (lane == 0) cf[j] = d;
    }
    __syncthreads();
    for (int j = tid; j < m; j += nt) {   // tv = |L|^-1 Y^T cf
        double d = 0.0;
        for (int i = 0; i < m; ++i) d += Y[i * ld + j] * cf[i];
        tv[j] = d / fabs(A[j * ld + j]);
    }
    __syncthreads();
    for (int i = tid; i < m; i += nt) {   // c1 = Y tv - cf / alpha   (coefficients of Q in the step)
        double d = 0.0;
        for (int j = 0; j < m; ++j) d += Y[i * ld + j] * tv[j];
        c1[i] = d - cf[i] / alpha;
    }
    __syncthreads();
    double mx2 = 0.0;
    for (int i = tid; i < a1 - a0; i += nt) {
        double s2 = 0.0;
        for (int xk = 0; xk < 3; ++xk) {
            const int k = 3 * i + xk;
            double d = F(k) / alpha;
            for (int j = 0; j < m; ++j) d += Q[(size_t)j * n + k] * c1[j];
            w[k] = d;
            s2 += d * d;
        }
        mx2 = fmax(mx2, s2);
    }
    mx2 = block_max(mx2, red);
    const double longest = sqrt(mx2);
    const double scale = longest >= maxstep ? maxstep / longest : 1.0;
    for (int k = tid; k < n; k += nt) {
        xr0[k] = x[k];
        xf0[k] = F(k);
        x[k] += scale * w[k];
    }
    if (tid == 0) {
        S.m = m;
        S.has_prev = 1;
        S.steps += 1;
        st[b] = S;
        atomicAdd(n_active, 1);
    }
}

__global__ void k_narrow_forces(int n3, const double *__restrict__ f64, float *__restrict__ f32) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) f32[i] = (float)f64[i];
}

template <class State>
__global__ void k_relax_report(int B, const State *__restrict__ st, int *__restrict__ steps, uint8_t *__restrict__ conv) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    steps[b] = st[b].steps;
    conv[b] = (uint8_t)st[b].converged;
}

// ---- trajectory observer ----------------------------------------------------------------------------------------------------
// Counterpart of TrajectoryObserver + dyn.attach(obs, interval=record_interval) (reference mcmc/dynamics.py:20-80,131-151): ASE
// calls the observer after the initial evaluation (nsteps = 0) and after every step whose count is a multiple of the
// interval, including the step after which the run stops.  Here: after the evaluation of an iteration and before its step
// kernel, every chain that is still running and whose own step counter is a multiple of the interval copies its positions,
// forces (FixAtoms applied, like atoms.get_forces()) and energy into record steps / interval.  The chain's counter is used,
// not the driver's iteration index: the two part after a neighbor-capacity regrow.
template <class State>
__global__ void __launch_bounds__(256)
k_traj_record(const int *__restrict__ cfg_start, const int *__restrict__ counters, const State *__restrict__ st,
              const unsigned char *__restrict__ active, int interval, int nrec, int B, int N, const uint8_t *__restrict__ fixed,
              const double *__restrict__ pos, const float *__restrict__ forces, const float *__restrict__ e32,
              const double *__restrict__ e64, double *__restrict__ ring_pos, float *__restrict__ ring_f,
              double *__restrict__ ring_e, int *__restrict__ ring_n) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (counters[2] || !active[b]) return;
    const int steps = st[b].steps;
    if (st[b].converged || steps % interval) return;
    const int r = steps / interval;
    if (r >= nrec) return;
    const int a0 = cfg_start[b], a1 = cfg_start[b + 1];
    for (int k = 3 * a0 + tid; k < 3 * a1; k += blockDim.x) {
        ring_pos[(size_t)r * 3 * N + k] = pos[k];
        ring_f[(size_t)r * 3 * N + k] = (fixed && fixed[k / 3]) ? 0.f : forces[k];
    }
    if (tid == 0) {
        ring_e[(size_t)r * B + b] = e64 ? e64[b] : (double)e32[b];
        if (ring_n[b] < r + 1) ring_n[b] = r + 1;
    }
}

// ---- host driver ----------------------------------------------------------------------------------------------------------
// Iteration i: evaluate energies / forces of the chains still active, then (device side) test convergence and step.  The host
// only enqueues; every POLL iterations it reads the number of chains that took a step and the neighbor-capacity flag.  If
// the capacity overflowed, the step kernels of the affected iterations did nothing (they test the flag): the buffers are
// regrown and the loop continues from the same positions.  Iterations enqueued after every chain has converged find no
// active chain and cost only their (empty) launches.
static size_t bfgs_lds_bytes(int mmax) { return sizeof(double) * ((size_t)2 * mmax * (mmax | 1) + 5 * (size_t)mmax + 8); }

int relax_run(vssr_handle *h, int method, const vssr_fire_params *fp, const vssr_bfgs_params *bp,
              const uint8_t *fixed_host, uint32_t want) {
    const int B = h->n_cfg, N = h->n_atoms;
    hipStream_t st = h->stream;
    const int max_steps = method == 1 ? bp->max_steps : fp->max_steps;
    const double fmax_tol = method == 1 ? bp->fmax : fp->fmax;
    int mmax = 0;
    if (h->d_fixed.ensure((size_t)N) || h->d_relax_steps.ensure(sizeof(int) * B) || h->d_relax_conv.ensure((size_t)B) ||
        h->d_active.ensure((size_t)B) || h->d_counters.ensure(sizeof(int) * 4))
        return set_err(h, VSSR_E_NOMEM, "relaxation state: out of device memory");
    if (method == 0) {
        if (h->d_vel.ensure(sizeof(double) * 3 * N) || h->d_fire.ensure(sizeof(FireState) * B))
            return set_err(h, VSSR_E_NOMEM, "FIRE state: out of device memory");
    } else {
        // Two basis vectors per Hessian update; the rotated matrices of the eigen-decomposition (2 x mmax^2 doubles) live in
        // LDS, which holds 46 updates.  A longer relaxation keeps stepping with the Hessian of its first 46 updates (the
        // step kernel skips the update when the basis is full): identical to ASE up to step 46, a frozen quasi-Newton
        // method beyond (the reference's configurations use 20 steps).
        mmax = 2 * max_steps + 2;
        while (bfgs_lds_bytes(mmax) > 160 * 1024 - 4096) mmax -= 2;   // (the kernel also holds 2 KB of static reduction scratch)
        if (h->d_fire.ensure(sizeof(BfgsState) * B) || h->d_vel.ensure(sizeof(double) * 3 * N * 3) ||
            h->d_bfgs_q.ensure(sizeof(double) * 3 * (size_t)N * mmax) || h->d_bfgs_b.ensure(sizeof(double) * (size_t)B * mmax * mmax))
            return set_err(h, VSSR_E_NOMEM, "BFGS state: out of device memory");
        VSSR_HIP(h, hipFuncSetAttribute((const void *)k_bfgs_step, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)bfgs_lds_bytes(mmax)));
    }
    const uint8_t *fixed = nullptr;
    if (fixed_host) {
        VSSR_HIP(h, hipMemcpyAsync(h->d_fixed.p, fixed_host, (size_t)N, hipMemcpyHostToDevice, st));
        fixed = h->d_fixed.as<uint8_t>();
    }
    unsigned char *active = h->d_active.as<unsigned char>();
    if (method == 0)
        hipLaunchKernelGGL(k_fire_init, dim3((B + 127) / 128), dim3(128), 0, st, B, (double)fp->dt, (double)fp->astart,
                           h->d_fire.as<FireState>(), active);
    else
        hipLaunchKernelGGL(k_bfgs_init, dim3((B + 127) / 128), dim3(128), 0, st, B, h->d_fire.as<BfgsState>(), active);
    // The evaluation kernels skip chains whose entry of `active` is 0 -- once the mask is handed to them.  While every chain is
    // still running the mask stays away (h->active_mask = nullptr): the masked forms of the kernels cost ~0.15 ms per evaluation
    // (tile / chain tests, the per-chain instead of the streaming gradient reduction), and with the reference's settings (fmax
    // 0.01 within 20 steps) no chain ever converges.  The host learns at a poll that a chain has stopped stepping and installs
    // the mask from then on; a chain that converged inside the current poll window is evaluated a few more times at its final
    // positions, which reproduces its results bit for bit.
    h->active_mask = nullptr;
    bool masked = false;
    int *n_active_d = h->d_counters.as<int>() + 3;   // counters[3] is free for this purpose
    const int rec_iv = h->traj_interval, nrec = rec_iv > 0 ? max_steps / rec_iv + 1 : 0;
    h->traj_records = 0;
    if (nrec) {
        if (h->d_traj_pos.ensure(sizeof(double) * 3 * (size_t)N * nrec) || h->d_traj_f.ensure(sizeof(float) * 3 * (size_t)N * nrec) ||
            h->d_traj_e.ensure(sizeof(double) * (size_t)B * nrec) || h->d_traj_n.ensure(sizeof(int) * B))
            return set_err(h, VSSR_E_NOMEM, "trajectory records: out of device memory");
        VSSR_HIP(h, hipMemsetAsync(h->d_traj_n.p, 0, sizeof(int) * B, st));
        h->traj_records = nrec; h->traj_B = B; h->traj_N = N;
    }
    const int POLL = 4;
    int rc = VSSR_OK;
    auto evaluate = [&]() {
        return h->kind == 2 ? tersoff_run(h, want | VSSR_WANT_FORCES)
                            : h->kind == 3 ? eam_run(h, want | VSSR_WANT_FORCES) : painn_run(h, want | VSSR_WANT_FORCES);
    };
    h->relax_lockstep = 0;
    h->relax_chain_evals = 0;
    for (int it = 0; it <= max_steps && !rc; ++it) {
        rc = evaluate();
        if (rc) break;
        ++h->relax_lockstep;
        h->relax_chain_evals += B;
        const bool last = it == max_steps;
        {   // (last iteration: every unconverged chain has taken relax_steps steps -- the kernel only tests convergence, like the
            // final check of ASE's run loop)
            VSSR_HIP(h, hipMemsetAsync(n_active_d, 0, sizeof(int), st));
            if (h->kind == 2 || h->kind == 3) {   // Tersoff / EAM forces are fp64 on the device: the optimizer state works on an fp32 copy
                if (h->d_forces.ensure(sizeof(float) * 3 * N)) { rc = set_err(h, VSSR_E_NOMEM, "force buffer"); break; }
                hipLaunchKernelGGL(k_narrow_forces, dim3((3 * N + 255) / 256), dim3(256), 0, st, 3 * N,
                                   h->d_ters_f.as<double>(), h->d_forces.as<float>());
            }
            const float *forces = h->d_forces.as<float>();
            if (nrec) {
                const bool f64 = h->kind == 2 || h->kind == 3;
                const float *e32 = f64 ? nullptr : h->d_energy.as<float>();
                const double *e64 = f64 ? h->d_ters_e.as<double>() : nullptr;
                if (method == 0)
                    hipLaunchKernelGGL(k_traj_record<FireState>, dim3(B), dim3(256), 0, st, h->d_cfg_start.as<int>(), h->d_counters.as<int>(),
                                       h->d_fire.as<FireState>(), active, rec_iv, nrec, B, N, fixed, h->d_pos.as<double>(), forces, e32, e64,
                                       h->d_traj_pos.as<double>(), h->d_traj_f.as<float>(), h->d_traj_e.as<double>(), h->d_traj_n.as<int>());
                else
                    hipLaunchKernelGGL(k_traj_record<BfgsState>, dim3(B), dim3(256), 0, st, h->d_cfg_start.as<int>(), h->d_counters.as<int>(),
                                       h->d_fire.as<BfgsState>(), active, rec_iv, nrec, B, N, fixed, h->d_pos.as<double>(), forces, e32, e64,
                                       h->d_traj_pos.as<double>(), h->d_traj_f.as<float>(), h->d_traj_e.as<double>(), h->d_traj_n.as<int>());
            }
            if (method == 0)
                hipLaunchKernelGGL(k_fire_step, dim3(B), dim3(256), 0, st, h->d_cfg_start.as<int>(), h->d_counters.as<int>(),
                                   forces, fixed, fmax_tol, (double)fp->maxstep, (double)fp->dtmax, (double)fp->finc,
                                   (double)fp->fdec, (double)fp->astart, (double)fp->fa, fp->nmin, max_steps, h->d_pos.as<double>(),
                                   h->d_vel.as<double>(), h->d_fire.as<FireState>(), active, n_active_d);
            else
                hipLaunchKernelGGL(k_bfgs_step, dim3(B), dim3(256), bfgs_lds_bytes(mmax), st, h->d_cfg_start.as<int>(),
                                   h->d_counters.as<int>(), forces, fixed, fmax_tol, (double)bp->alpha, (double)bp->maxstep, mmax, max_steps,
                                   h->d_pos.as<double>(), h->d_vel.as<double>(), h->d_vel.as<double>() + 3 * (size_t)N,
                                   h->d_bfgs_q.as<double>(), h->d_bfgs_b.as<double>(),
                                   h->d_vel.as<double>() + 6 * (size_t)N, h->d_fire.as<BfgsState>(), active, n_active_d);
            VSSR_HIP(h, hipMemcpyAsync(h->h_counters + 3, n_active_d, sizeof(int), hipMemcpyDeviceToHost, st));
        }
        if (last || (it + 1) % POLL == 0) {
            VSSR_HIP(h, hipStreamSynchronize(st));
            if (h->h_counters[2]) {   // neighbor capacity overflow in one of the enqueued evaluations: grow, redo
                if (h->h_counters[0] <= 0) { rc = set_err(h, VSSR_E_CAPACITY, "neighbor list exceeds 2^31 slots"); break; }
                h->slot_cap = (int64_t)h->h_counters[0] + (h->cap_tight ? 0 : (int64_t)h->h_counters[0] / 8) + 64;
                // the step kernels behind the overflowed evaluations did not move anything and did not count steps: go back
                // by one polling window (chains that did step in it are held to relax_steps by their own step counters)
                it -= POLL;
                if (it < -1) it = -1;
                if (++h->relax_regrows > (h->cap_tight ? 64 : 8)) { rc = set_err(h, VSSR_E_CAPACITY, "neighbor capacity could not be satisfied"); break; }
                continue;
            }
            if (!last && h->h_counters[3] == 0) break;   // no chain stepped in the last iteration: all converged, results final
            // (never at the final poll: the last evaluation above ran unmasked over every chain, the resident graph is complete)
            if (!last && !masked && h->h_counters[3] < B) { h->active_mask = active; masked = true; }
        }
    }
    h->active_mask = nullptr;
    if (rc) return rc;
    if (method == 0)
        hipLaunchKernelGGL(k_relax_report<FireState>, dim3((B + 127) / 128), dim3(128), 0, st, B, h->d_fire.as<FireState>(),
                           h->d_relax_steps.as<int>(), h->d_relax_conv.as<uint8_t>());
    else
        hipLaunchKernelGGL(k_relax_report<BfgsState>, dim3((B + 127) / 128), dim3(128), 0, st, B, h->d_fire.as<BfgsState>(),
                           h->d_relax_steps.as<int>(), h->d_relax_conv.as<uint8_t>());
    VSSR_HIP(h, hipGetLastError());
    h->ran = true;
    h->graph_partial = masked;   // (with the mask in place the last evaluation covered only the chains still running: see vssr_batch_stats)
    return VSSR_OK;
}

// ---- LAMMPS min_style cg: the state machine lives in cg_dev.h (cg_step_chain) --------------------------------------------
__global__ void k_cg_init(int B, CgState *__restrict__ st, unsigned char *__restrict__ active) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    CgState S = {};
    S.phase = PH_START;
    st[b] = S;
    active[b] = 1;
}

__global__ void __launch_bounds__(256)
k_cg_step(const int *__restrict__ cfg_start, const int *__restrict__ counters, const double *__restrict__ energy,
          const double *__restrict__ forces, const uint8_t *__restrict__ fixed, int max_iter, int max_eval, double etol,
          double ftol, double dmax, double *__restrict__ pos, double *__restrict__ x0all, double *__restrict__ hall,
          double *__restrict__ gall, CgState *__restrict__ st, unsigned char *__restrict__ active, int *__restrict__ n_active) {
    __shared__ double red[256];
    if (counters[2]) return;
    cg_step_chain(blockIdx.x, red, cfg_start, energy, forces, fixed, max_iter, max_eval, etol, ftol, dmax, pos, x0all, hall, gall, st, active,
                  n_active);
}

// ---- live-chain compaction of the resident batch (fp64 analytic potentials) ---------------------------------------------------
// The CG minimiser stops every chain by its own criteria; with the activity mask alone a finished chain still costs its share of
// every later launch (grids are sized for the whole batch, its workgroups leave at once).  At a poll with at most 3/4 of the
// resident chains still running the batch is PHYSICALLY compacted: the live chains' inputs (positions, types, cells) and optimizer
// state are gathered into a smaller resident batch, the finished chains' final positions / states are parked in full-size
// arrays, and every kernel of the path (neighbor build, potential, CG step) runs unchanged on the smaller batch -- a chain's
// results do not depend on its batch, so the trajectories are the same bit for bit (tests/test_cg.py).  The original batch is
// restored before the final static evaluation.
struct CmpView {   // device pointers of one layout of the per-chain / per-atom arrays
    int *cfg_start, *Z, *atom_cfg, *nimg;
    double *pos, *cell, *inv, *x0, *hh, *gg;
    uint8_t *pbc, *fixed;
    CgState *st;
};

__global__ void __launch_bounds__(1024)
k_cmp_plan(int B, const int *__restrict__ cfg_start, const unsigned char *__restrict__ active, const int *__restrict__ live,
           int *__restrict__ live_new, int *__restrict__ src, int *__restrict__ start_new, int *__restrict__ totals) {
    __shared__ int sc[1024], sa[1024];
    const int t = threadIdx.x, per = (B + 1023) / 1024, c0 = min(B, t * per), c1 = min(B, c0 + per);
    int nc = 0, na = 0;
    for (int c = c0; c < c1; ++c)
        if (active[c]) { nc += 1; na += cfg_start[c + 1] - cfg_start[c]; }
    sc[t] = nc; sa[t] = na;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {   // inclusive scans
        const int vc = t >= d ? sc[t - d] : 0, va = t >= d ? sa[t - d] : 0;
        __syncthreads();
        sc[t] += vc; sa[t] += va;
        __syncthreads();
    }
    int oc = sc[t] - nc, oa = sa[t] - na;
    for (int c = c0; c < c1; ++c)
        if (active[c]) {
            live_new[oc] = live ? live[c] : c;
            src[oc] = c;
            start_new[oc] = oa;
            oc += 1;
            oa += cfg_start[c + 1] - cfg_start[c];
        }
    if (t == 1023) { start_new[sc[t]] = sa[t]; totals[0] = sc[t]; totals[1] = sa[t]; }
}

// chains of the CURRENT batch whose results are final (active == nullptr: all of them): positions and optimizer state to their
// places in the ORIGINAL batch
__global__ void __launch_bounds__(128)
k_cmp_flush(const int *__restrict__ cfg_start, const unsigned char *__restrict__ active, const int *__restrict__ live,
            const int *__restrict__ start0, const double *__restrict__ pos, const CgState *__restrict__ st,
            double *__restrict__ final_pos, CgState *__restrict__ final_st) {
    const int c = blockIdx.x;
    if (active && active[c]) return;
    const int o = live ? live[c] : c, a0 = cfg_start[c], n = 3 * (cfg_start[c + 1] - a0);
    const size_t d0 = 3 * (size_t)start0[o], s0 = 3 * (size_t)a0;
    for (int k = threadIdx.x; k < n; k += blockDim.x) final_pos[d0 + k] = pos[s0 + k];
    if (threadIdx.x == 0) final_st[o] = st[c];
}

__global__ void __launch_bounds__(128)
k_cmp_gather(const int *__restrict__ src, const int *__restrict__ start_new, CmpView from, CmpView to, unsigned char *__restrict__ active_new) {
    const int nc = blockIdx.x, c = src[nc], a0 = from.cfg_start[c], na = from.cfg_start[c + 1] - a0, b0 = start_new[nc];
    for (int k = threadIdx.x; k < 3 * na; k += blockDim.x) {
        to.pos[3 * (size_t)b0 + k] = from.pos[3 * (size_t)a0 + k];
        to.x0[3 * (size_t)b0 + k] = from.x0[3 * (size_t)a0 + k];
        to.hh[3 * (size_t)b0 + k] = from.hh[3 * (size_t)a0 + k];
        to.gg[3 * (size_t)b0 + k] = from.gg[3 * (size_t)a0 + k];
    }
    for (int k = threadIdx.x; k < na; k += blockDim.x) {
        to.Z[b0 + k] = from.Z[a0 + k];
        to.atom_cfg[b0 + k] = nc;
        if (from.fixed) to.fixed[b0 + k] = from.fixed[a0 + k];
    }
    if (threadIdx.x < 9) { to.cell[9 * (size_t)nc + threadIdx.x] = from.cell[9 * (size_t)c + threadIdx.x]; to.inv[9 * (size_t)nc + threadIdx.x] = from.inv[9 * (size_t)c + threadIdx.x]; }
    if (threadIdx.x < 3) { to.nimg[3 * nc + threadIdx.x] = from.nimg[3 * c + threadIdx.x]; to.pbc[3 * nc + threadIdx.x] = from.pbc[3 * c + threadIdx.x]; }
    if (threadIdx.x == 0) { to.st[nc] = from.st[c]; active_new[nc] = 1; }
}

// the gathered arrays back over the resident ones, one launch (13 small device-to-device copies cost more than the evaluation of a
// small batch); live / start_new travel along
__global__ void __launch_bounds__(256)
k_cmp_copyback(int Bn, int Nn, CmpView to, CmpView from, int *__restrict__ live, const int *__restrict__ live_new,
               const int *__restrict__ start_new) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t k = t; k < 3 * (size_t)Nn; k += step) { to.pos[k] = from.pos[k]; to.x0[k] = from.x0[k]; to.hh[k] = from.hh[k]; to.gg[k] = from.gg[k]; }
    for (size_t k = t; k < (size_t)Nn; k += step) { to.Z[k] = from.Z[k]; to.atom_cfg[k] = from.atom_cfg[k]; if (from.fixed && to.fixed) to.fixed[k] = from.fixed[k]; }
    for (size_t k = t; k < 9 * (size_t)Bn; k += step) { to.cell[k] = from.cell[k]; to.inv[k] = from.inv[k]; }
    for (size_t k = t; k < 3 * (size_t)Bn; k += step) { to.nimg[k] = from.nimg[k]; to.pbc[k] = from.pbc[k]; }
    for (size_t k = t; k < (size_t)Bn; k += step) { to.st[k] = from.st[k]; live[k] = live_new[k]; }
    for (size_t k = t; k <= (size_t)Bn; k += step) to.cfg_start[k] = start_new[k];
}

__global__ void k_cg_report(int B, const CgState *__restrict__ st, int *__restrict__ out /*[B][3]*/) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    out[3 * b] = st[b].niter; out[3 * b + 1] = st[b].neval; out[3 * b + 2] = st[b].reason;
}

int relax_cg(vssr_handle *h, const vssr_cg_params *cp, const uint8_t *fixed_host, uint32_t want) {
    const int B = h->n_cfg, N = h->n_atoms;
    hipStream_t st = h->stream;
    if (h->kind != 2 && h->kind != 3)
        return set_err(h, VSSR_E_STATE, "conjugate gradients need an fp64 potential (Tersoff / EAM handle); use BFGS or FIRE");
    if (h->d_fixed.ensure((size_t)N) || h->d_relax_steps.ensure(sizeof(int) * 3 * B) || h->d_active.ensure((size_t)B) ||
        h->d_counters.ensure(sizeof(int) * 4) || h->d_fire.ensure(sizeof(CgState) * B) || h->d_vel.ensure(sizeof(double) * 9 * N))
        return set_err(h, VSSR_E_NOMEM, "CG state: out of device memory");
    const uint8_t *fixed = nullptr;
    if (fixed_host) {
        VSSR_HIP(h, hipMemcpyAsync(h->d_fixed.p, fixed_host, (size_t)N, hipMemcpyHostToDevice, st));
        fixed = h->d_fixed.as<uint8_t>();
    }
    unsigned char *active = h->d_active.as<unsigned char>();
    hipLaunchKernelGGL(k_cg_init, dim3((B + 127) / 128), dim3(128), 0, st, B, h->d_fire.as<CgState>(), active);
    h->active_mask = active;
    int *n_active_d = h->d_counters.as<int>() + 3;
    const int POLL = 8;
    int rc = VSSR_OK;
    // ---- live-chain compaction (see k_cmp_plan): B_cur / N_cur = the resident batch the kernels see ------------------------
    const char *cmp_env = getenv("VSSR_RELAX_COMPACT");   // (read per call: 0 switches the compaction off -- A/B runs, the equality test)
    const bool cmp_enabled = !cmp_env || atoi(cmp_env) != 0;
    const int cmp_min_atoms = cmp_env && atoi(cmp_env) > 1 ? atoi(cmp_env) : 65536;   // (VSSR_RELAX_COMPACT=n > 1: smallest resident batch, atoms)
    int B_cur = B, N_cur = N;
    bool compacted = false;
    double *const x0v = h->d_vel.as<double>(), *const hv = x0v + 3 * (size_t)N, *const gv = x0v + 6 * (size_t)N;
    // arena: [originals | parked finals | maps | gather targets]
    struct Arena {
        int *start0, *Z0, *cfg0, *nimg0, *live, *live_new, *src, *start_new, *totals;
        double *cell0, *inv0, *final_pos;
        uint8_t *pbc0, *fixed0;
        CgState *final_st;
        CmpView tmp;
    } A{};
    auto carve = [&]() -> int {
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
        const size_t Bz = (size_t)B, Nz = (size_t)N;
        const size_t o_start0 = take(4 * (Bz + 1)), o_Z0 = take(4 * Nz), o_cfg0 = take(4 * Nz), o_nimg0 = take(12 * Bz), o_live = take(4 * Bz),
                     o_live_new = take(4 * Bz), o_src = take(4 * Bz), o_start_new = take(4 * (Bz + 1)), o_tot = take(16),
                     o_cell0 = take(72 * Bz), o_inv0 = take(72 * Bz), o_fpos = take(24 * Nz), o_pbc0 = take(3 * Bz), o_fixed0 = take(Nz),
                     o_fst = take(sizeof(CgState) * Bz),
                     t_start = take(4 * (Bz + 1)), t_Z = take(4 * Nz), t_cfg = take(4 * Nz), t_nimg = take(12 * Bz), t_pos = take(24 * Nz),
                     t_cell = take(72 * Bz), t_inv = take(72 * Bz), t_x0 = take(24 * Nz), t_h = take(24 * Nz), t_g = take(24 * Nz),
                     t_pbc = take(3 * Bz), t_fixed = take(Nz), t_st = take(sizeof(CgState) * Bz);
        if (h->d_cmp.ensure(off)) return -1;
        char *p = h->d_cmp.as<char>();
        A.start0 = (int *)(p + o_start0); A.Z0 = (int *)(p + o_Z0); A.cfg0 = (int *)(p + o_cfg0); A.nimg0 = (int *)(p + o_nimg0);
        A.live = (int *)(p + o_live); A.live_new = (int *)(p + o_live_new); A.src = (int *)(p + o_src);
        A.start_new = (int *)(p + o_start_new); A.totals = (int *)(p + o_tot);
        A.cell0 = (double *)(p + o_cell0); A.inv0 = (double *)(p + o_inv0); A.final_pos = (double *)(p + o_fpos);
        A.pbc0 = (uint8_t *)(p + o_pbc0); A.fixed0 = (uint8_t *)(p + o_fixed0); A.final_st = (CgState *)(p + o_fst);
        A.tmp = CmpView{(int *)(p + t_start), (int *)(p + t_Z), (int *)(p + t_cfg), (int *)(p + t_nimg), (double *)(p + t_pos),
                        (double *)(p + t_cell), (double *)(p + t_inv), (double *)(p + t_x0), (double *)(p + t_h), (double *)(p + t_g),
                        (uint8_t *)(p + t_pbc), (uint8_t *)(p + t_fixed), (CgState *)(p + t_st)};
        return 0;
    };
    auto cur_view = [&]() {
        return CmpView{h->d_cfg_start.as<int>(), h->d_Z.as<int>(), h->d_atom_cfg.as<int>(), h->d_nimg.as<int>(), h->d_pos.as<double>(),
                       h->d_cell.as<double>(), h->d_invcell.as<double>(), x0v, hv, gv, h->d_pbc.as<uint8_t>(),
                       const_cast<uint8_t *>(fixed), h->d_fire.as<CgState>()};
    };
#define CMP_COPY(dst, src_, bytes) VSSR_HIP(h, hipMemcpyAsync((dst), (src_), (bytes), hipMemcpyDeviceToDevice, st))
    auto compact = [&]() -> int {
        const CmpView cur = cur_view();
        if (!compacted) {   // first time: keep the original batch
            if (carve()) return set_err(h, VSSR_E_NOMEM, "compaction arena: out of device memory");
            CMP_COPY(A.start0, cur.cfg_start, 4 * ((size_t)B + 1)); CMP_COPY(A.Z0, cur.Z, 4 * (size_t)N); CMP_COPY(A.cfg0, cur.atom_cfg, 4 * (size_t)N);
            CMP_COPY(A.nimg0, cur.nimg, 12 * (size_t)B); CMP_COPY(A.cell0, cur.cell, 72 * (size_t)B); CMP_COPY(A.inv0, cur.inv, 72 * (size_t)B);
            CMP_COPY(A.pbc0, cur.pbc, 3 * (size_t)B);
            if (fixed) CMP_COPY(A.fixed0, fixed, (size_t)N);
        }
        const int *live = compacted ? A.live : nullptr;
        hipLaunchKernelGGL(k_cmp_plan, dim3(1), dim3(1024), 0, st, B_cur, cur.cfg_start, active, live, A.live_new, A.src, A.start_new, A.totals);
        hipLaunchKernelGGL(k_cmp_flush, dim3(B_cur), dim3(128), 0, st, cur.cfg_start, active, live, A.start0, cur.pos, cur.st, A.final_pos, A.final_st);
        int tot[2] = {0, 0};
        VSSR_HIP(h, hipMemcpyAsync(tot, A.totals, sizeof(tot), hipMemcpyDeviceToHost, st));
        VSSR_HIP(h, hipStreamSynchronize(st));
        const int Bn = tot[0], Nn = tot[1];
        if (Bn <= 0 || Bn > B_cur || Nn <= 0 || Nn > N_cur) return set_err(h, VSSR_E_STATE, "live-chain compaction: inconsistent plan");
        hipLaunchKernelGGL(k_cmp_gather, dim3(Bn), dim3(128), 0, st, A.src, A.start_new, cur, A.tmp, active);
        // (the gather reads the resident arrays and writes the arena; one more launch copies the arena over the resident arrays)
        {
            CmpView from = A.tmp;
            if (!fixed) from.fixed = nullptr;
            const int blocks = (int)std::min<size_t>(1024, (3 * (size_t)Nn + 255) / 256);
            hipLaunchKernelGGL(k_cmp_copyback, dim3(blocks), dim3(256), 0, st, Bn, Nn, cur, from, A.live, A.live_new, A.start_new);
        }
        compacted = true;
        B_cur = Bn; N_cur = Nn;
        h->n_cfg = Bn; h->n_atoms = Nn;
        ++h->relax_compactions;
        return VSSR_OK;
    };
    auto restore = [&]() -> int {   // park what is still resident, then bring the original batch back (positions = the final ones)
        if (!compacted) return VSSR_OK;
        const CmpView cur = cur_view();
        hipLaunchKernelGGL(k_cmp_flush, dim3(B_cur), dim3(128), 0, st, cur.cfg_start, (const unsigned char *)nullptr, A.live, A.start0, cur.pos, cur.st,
                           A.final_pos, A.final_st);
        CMP_COPY(cur.cfg_start, A.start0, 4 * ((size_t)B + 1)); CMP_COPY(cur.Z, A.Z0, 4 * (size_t)N); CMP_COPY(cur.atom_cfg, A.cfg0, 4 * (size_t)N);
        CMP_COPY(cur.nimg, A.nimg0, 12 * (size_t)B); CMP_COPY(cur.cell, A.cell0, 72 * (size_t)B); CMP_COPY(cur.inv, A.inv0, 72 * (size_t)B);
        CMP_COPY(cur.pbc, A.pbc0, 3 * (size_t)B); CMP_COPY(cur.pos, A.final_pos, 24 * (size_t)N); CMP_COPY(cur.st, A.final_st, sizeof(CgState) * (size_t)B);
        if (fixed) CMP_COPY(cur.fixed, A.fixed0, (size_t)N);
        compacted = false;
        B_cur = B; N_cur = N;
        h->n_cfg = B; h->n_atoms = N;
        return VSSR_OK;
    };
#undef CMP_COPY
    h->relax_compactions = 0;
    // every launch is one evaluation; max_eval is tested between line searches, and a line search ends after at most ~60
    // halvings of alpha (fp64), so the launch budget is max_eval plus one line search plus setup / reset evaluations
    const long long max_launch = (long long)cp->max_eval + 72;
    auto regrow = [&]() -> int {   // capacity overflow seen at a poll: grow; the step kernels behind it did nothing
        if (h->h_counters[0] <= 0) return set_err(h, VSSR_E_CAPACITY, "neighbor list exceeds 2^31 slots");
        h->slot_cap = (int64_t)h->h_counters[0] + (h->cap_tight ? 0 : (int64_t)h->h_counters[0] / 8) + 64;
        if (++h->relax_regrows > 64) return set_err(h, VSSR_E_CAPACITY, "neighbor capacity could not be satisfied");
        return VSSR_OK;
    };
    long long it = 0;
    bool finished = false;
    h->relax_lockstep = 0;
    h->relax_chain_evals = 0;
    while (!rc && !finished) {
        for (; it < max_launch && !rc; ++it) {
            rc = h->kind == 2 ? tersoff_run(h, want | VSSR_WANT_FORCES) : eam_run(h, want | VSSR_WANT_FORCES);
            if (rc) break;
            ++h->relax_lockstep;
            h->relax_chain_evals += B_cur;
            // the count of chains still running is read at the polls only: it is cleared and copied back in those iterations (the
            // step kernels in between add to a value nobody looks at) -- two dispatches less per evaluation, ~15 of them at 48 atoms
            const bool poll_it = (it + 1) % POLL == 0;
            if (poll_it) VSSR_HIP(h, hipMemsetAsync(n_active_d, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_cg_step, dim3(B_cur), dim3(256), 0, st, h->d_cfg_start.as<int>(), h->d_counters.as<int>(),
                               h->d_ters_e.as<double>(), h->d_ters_f.as<double>(), fixed, cp->max_iter, cp->max_eval, cp->etol,
                               cp->ftol, cp->dmax, h->d_pos.as<double>(), h->d_vel.as<double>(), h->d_vel.as<double>() + 3 * (size_t)N,
                               h->d_vel.as<double>() + 6 * (size_t)N, h->d_fire.as<CgState>(), active, n_active_d);
            if (poll_it) {
                VSSR_HIP(h, hipMemcpyAsync(h->h_counters + 3, n_active_d, sizeof(int), hipMemcpyDeviceToHost, st));
                VSSR_HIP(h, hipStreamSynchronize(st));
                if (h->h_counters[2]) {
                    rc = regrow();
                    it -= POLL;   // the launches of this window are given back (those behind the overflow did nothing; a chain
                                  // that did step is bounded by its own iteration / evaluation counters)
                    continue;
                }
                if (h->h_counters[3] == 0) { finished = true; break; }   // every chain has finished
                // at most 3/4 of the resident chains are still running: continue on a compacted batch.  Only where the kernels are
                // throughput-bound: below ~one round of workgroups (256 CUs x 3 x 64 centres = 49 k atoms) a launch costs the same
                // whatever the live share, and the compaction (four launches + a host read) would only add to it (measured, 256
                // chains x 48 atoms: -3 %; profiles/r05/NOTES_tersoff.md)
                if (cmp_enabled && N_cur >= cmp_min_atoms && (long long)h->h_counters[3] * 4 <= (long long)B_cur * 3) rc = compact();
            }
        }
        if (rc || finished) break;
        // the budget ran out between two polls: look at the last window as well
        VSSR_HIP(h, hipStreamSynchronize(st));
        if (h->h_counters[2]) {
            rc = regrow();
            it -= POLL;
            continue;
        }
        finished = true;
    }
    h->active_mask = nullptr;
    if (rc) {   // (a compacted batch is not handed back half-way: the caller uploads again)
        if (compacted) { h->n_cfg = B; h->n_atoms = N; h->batch_valid = false; }
        return rc;
    }
    rc = restore();
    if (rc) return rc;
    // results of the final positions for every chain (finished chains were switched off at different times)
    rc = h->kind == 2 ? tersoff_run(h, want | VSSR_WANT_FORCES) : eam_run(h, want | VSSR_WANT_FORCES);
    if (rc) return rc;
    ++h->relax_lockstep;
    h->relax_chain_evals += B;
    hipLaunchKernelGGL(k_cg_report, dim3((B + 127) / 128), dim3(128), 0, st, B, h->d_fire.as<CgState>(), h->d_relax_steps.as<int>());
    VSSR_HIP(h, hipGetLastError());
    h->ran = true;
    return VSSR_OK;
}

}  // namespace vssr
