// relax.hip — lock-step FIRE relaxation of all resident chains on the device.
//
// Counterpart of optimize_slab (reference mcmc/dynamics.py:83-170) with optimizer="FIRE" (its default):
//   dyn = FIRE(slab); dyn.run(steps=relax_steps, fmax=0.01)
// ASE's FIRE (ase/optimize/fire.py; defaults dt=0.1, maxstep=0.2, dtmax=1.0, Nmin=5, finc=1.1, fdec=0.5,
// astart=0.1, fa=0.99) restated per chain; FixAtoms (reference mcmc/system.py:288-294) enters as a per-atom mask
// that zeroes the force.  The reference drives one structure from Python; here every chain carries its own
// (velocity, dt, a, Nsteps) state on the device and all chains step together: one energy+force evaluation of the
// whole batch per iteration, positions and forces never leave HBM.  A chain whose max |F_i| drops below fmax
// freezes (ASE's convergence test, evaluated before each step).
#include "vssr_internal.h"

namespace vssr {

struct FireState {   // per chain
    double dt, a;
    int nsteps_pos;  // steps since the last power <= 0 (ASE's Nsteps)
    int has_v;       // 0 until the first step (ASE: self.v is None)
    int steps;       // optimizer steps taken
    int converged;
};

__global__ void k_fire_init(int B, double dt0, double astart, FireState *__restrict__ st) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    st[b].dt = dt0; st[b].a = astart; st[b].nsteps_pos = 0; st[b].has_v = 0; st[b].steps = 0; st[b].converged = 0;
}

__device__ inline double block_sum(double v, double *red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    double r = red[0];
    __syncthreads();
    return r;
}
__device__ inline double block_max(double v, double *red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    double r = red[0];
    __syncthreads();
    return r;
}

// One workgroup per chain: convergence test on the forces of the CURRENT positions, then one FIRE step.
__global__ void __launch_bounds__(256)
k_fire_step(const int *__restrict__ cfg_start, const float *__restrict__ forces, const uint8_t *__restrict__ fixed,
            double fmax_tol, double maxstep, double dtmax, double finc, double fdec, double astart, double fa, int nmin,
            double *__restrict__ pos, double *__restrict__ vel, FireState *__restrict__ st, int *__restrict__ n_active) {
    __shared__ double red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
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
        if (tid == 0) st[b].converged = 1;
        return;
    }
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

__global__ void k_narrow_forces(int n3, const double *__restrict__ f64, float *__restrict__ f32) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) f32[i] = (float)f64[i];
}

__global__ void k_fire_report(int B, const FireState *__restrict__ st, int *__restrict__ steps, uint8_t *__restrict__ conv) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    steps[b] = st[b].steps;
    conv[b] = (uint8_t)st[b].converged;
}

int relax_fire(vssr_handle *h, const vssr_fire_params *fp, const uint8_t *fixed_host, uint32_t want) {
    const int B = h->n_cfg, N = h->n_atoms;
    hipStream_t st = h->stream;
    if (h->d_vel.ensure(sizeof(double) * 3 * N) || h->d_fire.ensure(sizeof(FireState) * B) ||
        h->d_fixed.ensure((size_t)N) || h->d_relax_steps.ensure(sizeof(int) * B) || h->d_relax_conv.ensure((size_t)B) ||
        h->d_counters.ensure(sizeof(int) * 4))
        return set_err(h, VSSR_E_NOMEM, "relaxation state: out of device memory");
    const uint8_t *fixed = nullptr;
    if (fixed_host) {
        VSSR_HIP(h, hipMemcpyAsync(h->d_fixed.p, fixed_host, (size_t)N, hipMemcpyHostToDevice, st));
        fixed = h->d_fixed.as<uint8_t>();
    }
    hipLaunchKernelGGL(k_fire_init, dim3((B + 127) / 128), dim3(128), 0, st, B, (double)fp->dt, (double)fp->astart,
                       h->d_fire.as<FireState>());
    int *n_active_d = h->d_counters.as<int>() + 3;   // counters[3] is free for this purpose
    for (int it = 0; it <= fp->max_steps; ++it) {
        int rc = h->kind == 2 ? tersoff_run(h, want | VSSR_WANT_FORCES) : painn_run(h, want | VSSR_WANT_FORCES);
        if (rc) return rc;
        if (it == fp->max_steps) break;
        VSSR_HIP(h, hipMemsetAsync(n_active_d, 0, sizeof(int), st));
        if (h->kind == 2) {   // Tersoff forces are fp64 on the device: the optimizer state works on an fp32 copy
            if (h->d_forces.ensure(sizeof(float) * 3 * N)) return set_err(h, VSSR_E_NOMEM, "force buffer");
            hipLaunchKernelGGL(k_narrow_forces, dim3((3 * N + 255) / 256), dim3(256), 0, st, 3 * N,
                               h->d_ters_f.as<double>(), h->d_forces.as<float>());
        }
        const float *forces = h->d_forces.as<float>();
        hipLaunchKernelGGL(k_fire_step, dim3(B), dim3(256), 0, st, h->d_cfg_start.as<int>(), forces, fixed,
                           (double)fp->fmax, (double)fp->maxstep, (double)fp->dtmax, (double)fp->finc, (double)fp->fdec,
                           (double)fp->astart, (double)fp->fa, fp->nmin, h->d_pos.as<double>(), h->d_vel.as<double>(),
                           h->d_fire.as<FireState>(), n_active_d);
        VSSR_HIP(h, hipMemcpyAsync(h->h_counters + 3, n_active_d, sizeof(int), hipMemcpyDeviceToHost, st));
        VSSR_HIP(h, hipStreamSynchronize(st));
        if (h->h_counters[2])   // neighbor capacity overflow: grow and redo this iteration's evaluation
            return set_err(h, VSSR_E_CAPACITY, "neighbor capacity exceeded during relaxation (re-upload and retry)");
        if (h->h_counters[3] == 0) break;   // every chain converged: positions did not move, results are final
    }
    hipLaunchKernelGGL(k_fire_report, dim3((B + 127) / 128), dim3(128), 0, st, B, h->d_fire.as<FireState>(),
                       h->d_relax_steps.as<int>(), h->d_relax_conv.as<uint8_t>());
    VSSR_HIP(h, hipGetLastError());
    h->ran = true;
    return VSSR_OK;
}

}  // namespace vssr
