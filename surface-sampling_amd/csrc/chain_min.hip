// chain_min.hip — the chain-resident minimiser: ONE workgroup runs the whole LAMMPS-style CG relaxation of ONE chain.
//
// Replaces, for small chains on the fp64 Tersoff potential, the lock-step driver of relax.hip (reference: `optimizer: "LAMMPS"`,
// LAMMMPSCalc.run_lammps_opt, mcmc/calculators/calculators.py:600-619 -> `min_style cg` / `minimize 1e-5 1e-5 {relax_steps} 10000`,
// tutorials/data/GaN_0001/GaN_0001_lammps_opt_template.txt; one relaxation per MC proposal, mcmc/system.py:450-470).
//
// Why: the GaN chains of BASELINE configs[1] have 48 atoms and stop after 21 .. 159 evaluations each (median 55).  In lock step
// every evaluation is ~12 dependent launches over the whole batch, each ~10 us of dispatch + drain whatever its size, and the
// batch runs until its slowest chain is done: 131 .. 161 lock-step evaluations per proposal, 2.2 .. 2.7 x the chain-evaluations
// the chains need (profiles/r05/NOTES_tersoff.md).  A 48-atom chain is a workgroup-sized problem: here a 256-thread workgroup
// owns a chain from its first evaluation to its stop criterion -- wrap, neighbor rows, Tersoff site terms, force gather, energy,
// CG state machine -- with barriers instead of launches between the phases, and leaves when ITS chain is done; the hardware
// workgroup scheduler hands the CU to the next chain.  No lock step, no host polls, one launch per relaxation.
//
// Same bits: every phase is the device function the lock-step kernels run (nbr_dev.h, tersoff_dev.h, cg_dev.h) with the same
// lane-to-work mapping (16 lanes per centre in the neighbor search, 4 lanes per centre in the site kernel, 256-thread
// reductions), so rows, energies, forces, iteration / evaluation counts and stop reasons equal the lock-step driver's bit for
// bit (tests/test_cg.py::test_chain_resident_minimiser_equals_the_lock_step_driver).  Only the slot numbering differs: a chain
// owns the fixed slot range [cfg_start[b], cfg_start[b + 1]) x cap_per_atom instead of a place in a batch-wide scan.
#include <algorithm>
#include "cg_dev.h"
#include "nbr_dev.h"
#ifdef CM_PHASE_TIMING   // sub-phases of the site tile: slots 10 .. 12 of g_cm_phase
#include <hip/hip_runtime.h>
extern __device__ unsigned long long g_cm_phase[16];
#define TS_MARK_INIT unsigned long long ts_t = wall_clock64();
#define TS_MARK(k) { __syncthreads(); const unsigned long long ts_n = wall_clock64(); if (threadIdx.x == 0 && blockIdx.x == 0) g_cm_phase[10 + (k)] += ts_n - ts_t; ts_t = wall_clock64(); }
#endif
#include "tersoff_dev.h"

#include <vector>

#ifdef CM_PHASE_TIMING   // debug build only (tools/gpu_cm_phase.py): 100 MHz wall-clock ticks per phase, workgroup 0
__device__ unsigned long long g_cm_phase[16];
#define CMP_INIT unsigned long long cmp_t = wall_clock64();
#define CMP(k) { __syncthreads(); const unsigned long long cmp_n = wall_clock64(); if (threadIdx.x == 0 && blockIdx.x == 0) g_cm_phase[k] += cmp_n - cmp_t; cmp_t = wall_clock64(); }
extern "C" int vssr_debug_cm_phases(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cm_phase), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_cm_phase), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define CMP_INIT
#define CMP(k)
#endif

namespace vssr {

struct ChainMinArgs {
    // resident batch
    int n_types, fast;
    const TersP *P;
    const int *type, *atom_cfg, *cfg_start, *nimg;
    const double *cell, *invcell;
    const uint8_t *pbc, *fixed;
    double *pos;
    // graph scratch (chain b: rows row_start[b + i], slots [cfg_start[b] * cap_per_atom, ...))
    double *wpos;
    int *wrap, *deg, *row_start, *edge_S, *rev;
    float4 *edge;
    unsigned long long *hits;
    int hits_stride, cap_per_atom;
    double rc2;
    // potential results
    double *eps, *gslot, *e_atom, *forces, *energy;
    // CG
    int max_iter, max_eval;
    double etol, ftol, dmax;
    double *x0, *hh, *gg;
    CgState *st;
    unsigned char *active;
    int *flags;   // [0]: a chain ran out of slot capacity (host regrows, relaunches); [1]: sink of the state machine's live counter
    int *n_evals; // [B] evaluations this workgroup made for its chain (work counters)
    long long max_launch;
};

constexpr int CM_THREADS = 256, CM_MAX_ATOMS = 256, CM_LPC = 16;

__global__ void __launch_bounds__(CM_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_cg_chain(const ChainMinArgs *__restrict__ Ap) {
    const ChainMinArgs &A = *Ap;   // (arguments in memory: ~45 pointers and scalars live across every phase would otherwise sit in -- and spill from -- scalar registers)
    __shared__ TersShared sh;
    __shared__ double red[256];
    __shared__ int scan[CM_THREADS];
    __shared__ int s_over;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int a0 = A.cfg_start[b], n = A.cfg_start[b + 1] - a0, a1 = a0 + n;
    const long long slot0 = (long long)a0 * A.cap_per_atom, slot1 = slot0 + (long long)n * A.cap_per_atom;
    int *rs = A.row_start + b;   // rs[i], rs[i + 1] for the global atom index i of this chain: n + 1 entries of its own
    if (A.fast) tersoff_derive_params(sh, A.n_types, A.P);
    if (tid == 0) s_over = 0;
    __syncthreads();
    int evals = 0;

    // one energy + force evaluation of the chain at its current positions; false: slot capacity exceeded (nothing was stepped)
    auto evaluate = [&]() -> bool {
        CMP_INIT
        for (int i = a0 + tid; i < a1; i += CM_THREADS) wrap_atom(i, A.pos, A.atom_cfg, A.cell, A.invcell, A.pbc, A.wpos, A.wrap);
        __syncthreads();
        CMP(0)
        for (int base = a0; base < a1; base += CM_THREADS / CM_LPC) {   // 16 lanes per centre, 16 centres per pass
            const int i = base + tid / CM_LPC;
            if (i < a1)
                nbr_row<false, CM_LPC>(i, A.wpos, A.atom_cfg, A.cfg_start, A.cell, A.invcell, A.nimg, A.rc2, A.deg, rs, A.edge, A.edge_S, slot1,
                                       A.hits, A.hits_stride, nullptr);
        }
        __syncthreads();
        CMP(1)
        {   // rows of the chain: exclusive scan of the padded degrees (n <= 256: one atom per thread)
            const int pd = tid < n ? max((A.deg[a0 + tid] + 3) & ~3, 8) : 0;
            scan[tid] = pd;
            __syncthreads();
            for (int d = 1; d < CM_THREADS; d <<= 1) {
                const int v = tid >= d ? scan[tid - d] : 0;
                __syncthreads();
                scan[tid] += v;
                __syncthreads();
            }
            if (tid < n) rs[a0 + tid] = (int)(slot0 + scan[tid] - pd);
            if (tid == CM_THREADS - 1) {
                rs[a1] = (int)(slot0 + scan[tid]);
                if (slot0 + scan[tid] > slot1) s_over = 1;
            }
            __syncthreads();
            if (s_over) return false;
        }
        CMP(2)
        for (int base = a0; base < a1; base += CM_THREADS / CM_LPC) {
            const int i = base + tid / CM_LPC;
            if (i < a1)
                nbr_row<true, CM_LPC>(i, A.wpos, A.atom_cfg, A.cfg_start, A.cell, A.invcell, A.nimg, A.rc2, A.deg, rs, A.edge, A.edge_S, slot1,
                                      A.hits, A.hits_stride, nullptr);
        }
        __syncthreads();
        CMP(3)
        for (int base = a0; base < a1; base += CM_THREADS / CM_LPC) {
            const int i = base + tid / CM_LPC;
            if (i < a1) rev_row<CM_LPC>(i, rs, A.edge, A.edge_S, A.rev);
        }
        __syncthreads();
        CMP(4)
        if (A.fast) {
            for (int t0 = a0; t0 < a1; t0 += TS_CENTRES) {
                const int i = t0 + (tid >> 2);
                tersoff_site4_tile(sh, i, i < a1, A.n_types, A.type, A.atom_cfg, A.cell, A.wpos, rs, A.edge, A.edge_S, A.eps, A.gslot);
                __syncthreads();
            }
        }
        CMP(5)
        for (int i = a0 + tid; i < a1; i += CM_THREADS)
            tersoff_site_atom(i, A.n_types, A.P, A.type, A.atom_cfg, A.cell, A.wpos, rs, A.edge, A.edge_S, A.eps, A.gslot, A.fast ? TS_MAXD : -1);
        __syncthreads();
        CMP(6)
        for (int i = a0 + tid; i < a1; i += CM_THREADS) tersoff_gather_atom(i, rs, A.rev, A.eps, A.gslot, A.e_atom, A.forces);
        __syncthreads();
        CMP(7)
        tersoff_chain_energy(b, red, A.cfg_start, A.e_atom, A.energy);
        __syncthreads();
        CMP(8)
        evals += 1;
        return true;
    };

    // Every stop of the state machine happens right behind an evaluation of the positions the chain is left at (cg_dev.h: no branch
    // moves atoms and stops), so the results of the LAST evaluation are the static results of the final geometry -- what the
    // lock-step driver obtains with one more batch-wide evaluation after its loop.
    for (long long it = 0;; ++it) {
        if (!evaluate()) {   // (state untouched: the host enlarges the slot pools and launches again)
            if (tid == 0) { atomicOr(A.flags, 1); A.n_evals[b] += evals; }
            return;
        }
        if (it >= A.max_launch) break;   // launch budget of the lock-step driver exhausted (max_eval + 72 evaluations)
        {
            CMP_INIT
            cg_step_chain(b, red, A.cfg_start, A.energy, A.forces, A.fixed, A.max_iter, A.max_eval, A.etol, A.ftol, A.dmax, A.pos, A.x0, A.hh, A.gg,
                          A.st, A.active, A.flags + 1);
            __syncthreads();
            CMP(9)
        }
        if (A.st[b].reason) break;   // (uniform: written by thread 0 in front of the barrier)
    }
    if (tid == 0) A.n_evals[b] += evals;
}

__global__ void k_cm_init(int B, CgState *__restrict__ st, unsigned char *__restrict__ active, int *__restrict__ n_evals) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    CgState S = {};
    S.phase = PH_START;
    st[b] = S;
    active[b] = 1;
    n_evals[b] = 0;
}
__global__ void k_cm_report(int B, const CgState *__restrict__ st, int *__restrict__ out /*[B][3]*/) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    out[3 * b] = st[b].niter; out[3 * b + 1] = st[b].neval; out[3 * b + 2] = st[b].reason;
}

// Which driver: the chain-resident kernel wins while the batch is small enough for launch latency and lock-step waste to dominate
// (same-box A/B, GaN 48-atom chains, MC proposals/s, profiles/r05/NOTES_tersoff.md: 256 chains 20.9 k vs 15.9 k, 1 024: 45.0 k vs
// 36.9 k, 4 096: 51.5 k vs 52 .. 56 k, 16 384: 63 k vs 70 .. 75 k): two 4-wave workgroups per CU cannot hide the fp64 latency chains
// of the site terms as well as the batch-wide kernels do at 12 waves per CU once every CU has work for many rounds.
// VSSR_CG_FUSED: 0 = always the lock-step driver, 1 = the chain-resident kernel whenever it applies, unset = by batch size.
bool chain_min_supported(const vssr_handle *h) {
    if (h->kind != 2 || h->max_cfg_atoms > CM_MAX_ATOMS) return false;
    const char *e = getenv("VSSR_CG_FUSED");   // (read per call)
    if (e) return atoi(e) != 0;
    // the automatic choice covers the regime that was measured: chains of <= 64 atoms (one site tile per workgroup).  A 200-atom
    // chain runs its 64-centre site tiles one after another inside one workgroup; that regime has no A/B, so it takes the lock-step
    // kernels unless VSSR_CG_FUSED=1 asks for the chain-resident one (advisor r5)
    return h->n_cfg <= 3072 && h->max_cfg_atoms <= 64;
}

// Same contract as relax_cg (relax.hip): afterwards the batch holds the minimised positions, d_ters_e / _ea / _f the static results of
// those geometries, d_relax_steps [B][3] = (iterations, evaluations, stop reason) per chain.
int chain_min_cg(vssr_handle *h, const vssr_cg_params *cp, const uint8_t *fixed_host, uint32_t want) {
    (void)want;
    const int B = h->n_cfg, N = h->n_atoms;
    hipStream_t st = h->stream;
    if (h->d_fixed.ensure((size_t)N) || h->d_relax_steps.ensure(sizeof(int) * 3 * (size_t)B) || h->d_active.ensure((size_t)B) ||
        h->d_counters.ensure(sizeof(int) * 4) || h->d_fire.ensure(sizeof(CgState) * (size_t)B) || h->d_vel.ensure(sizeof(double) * 9 * (size_t)N) ||
        h->d_cm.ensure(sizeof(int) * ((size_t)B + 8) + 1024) || h->d_wpos.ensure(sizeof(double) * 3 * (size_t)N) ||
        h->d_wrap.ensure(sizeof(int) * 3 * (size_t)N) || h->d_deg.ensure(sizeof(int) * (size_t)N) ||
        h->d_row_start.ensure(sizeof(int) * ((size_t)N + B + 1)) || h->d_ters_e.ensure(sizeof(double) * (size_t)B) ||
        h->d_ters_ea.ensure(sizeof(double) * (size_t)N) || h->d_ters_f.ensure(sizeof(double) * 3 * (size_t)N))
        return set_err(h, VSSR_E_NOMEM, "chain-resident minimiser: out of device memory");
    const uint8_t *fixed = nullptr;
    if (fixed_host) {
        VSSR_HIP(h, hipMemcpyAsync(h->d_fixed.p, fixed_host, (size_t)N, hipMemcpyHostToDevice, st));
        fixed = h->d_fixed.as<uint8_t>();
    }
    static_assert(sizeof(ChainMinArgs) <= 1024, "argument block");
    ChainMinArgs *d_args = reinterpret_cast<ChainMinArgs *>(h->d_cm.as<char>());   // [arguments (1 KB) | flags [8] | evaluations [B]]
    int *flags = reinterpret_cast<int *>(h->d_cm.as<char>() + 1024), *n_evals = flags + 8;
    hipLaunchKernelGGL(k_cm_init, dim3((B + 127) / 128), dim3(128), 0, st, B, h->d_fire.as<CgState>(), h->d_active.as<unsigned char>(), n_evals);
    h->relax_lockstep = 0;
    h->relax_chain_evals = 0;
    h->relax_compactions = 0;
    const double rc = h->ters_cutmax;
    // slots per atom of the per-chain pools: the handle's capacity, or what an earlier chain-resident relaxation had to grow to.  The
    // grown value stays with THIS driver (cm_cap_per_atom): the batch-wide runs size their buffers from cap_per_atom and repair an
    // overflow exactly, they must not inherit up to 64x from a pool that doubles (advisor r5)
    int cap = std::max(h->cap_per_atom, h->cm_cap_per_atom);
    for (int attempt = 0;; ++attempt) {
        // slot pools: every chain owns n_atoms x cap slots
        const long long slots = (long long)N * cap + 64;
        if (slots > 2147483000LL) return set_err(h, VSSR_E_CAPACITY, "neighbor list exceeds 2^31 slots");
        if (h->slot_cap < slots) h->slot_cap = slots;
        if (h->d_edge.ensure(sizeof(float4) * h->slot_cap) || h->d_edge_S.ensure(sizeof(int) * h->slot_cap) ||
            h->d_rev.ensure(sizeof(int) * h->slot_cap) || h->d_gbar.ensure(sizeof(double) * 4 * (size_t)h->slot_cap))
            return set_err(h, VSSR_E_NOMEM, "neighbor buffers: out of device memory");
        unsigned long long *hits_buf = nullptr;
        const int hits_stride = (h->max_cfg_atoms + 63) & ~63;
        if (h->max_images <= 64 && (size_t)N * hits_stride * 8 <= ((size_t)1 << 30) && !h->d_hits.ensure((size_t)N * hits_stride * 8))
            hits_buf = h->d_hits.as<unsigned long long>();
        VSSR_HIP(h, hipMemsetAsync(flags, 0, sizeof(int) * 8, st));
        ChainMinArgs A{};
        A.n_types = h->n_types;
        A.fast = (h->n_types * h->n_types * h->n_types <= TS_MAXP) ? 1 : 0;
        A.P = h->ters_params.as<TersP>();
        A.type = h->d_Z.as<int>(); A.atom_cfg = h->d_atom_cfg.as<int>(); A.cfg_start = h->d_cfg_start.as<int>(); A.nimg = h->d_nimg.as<int>();
        A.cell = h->d_cell.as<double>(); A.invcell = h->d_invcell.as<double>();
        A.pbc = h->d_pbc.as<uint8_t>(); A.fixed = fixed;
        A.pos = h->d_pos.as<double>();
        A.wpos = h->d_wpos.as<double>(); A.wrap = h->d_wrap.as<int>(); A.deg = h->d_deg.as<int>(); A.row_start = h->d_row_start.as<int>();
        A.edge_S = h->d_edge_S.as<int>(); A.rev = h->d_rev.as<int>(); A.edge = h->d_edge.as<float4>();
        A.hits = hits_buf; A.hits_stride = hits_stride; A.cap_per_atom = cap; A.rc2 = rc * rc;
        A.eps = h->d_gbar.as<double>(); A.gslot = A.eps + h->slot_cap;
        A.e_atom = h->d_ters_ea.as<double>(); A.forces = h->d_ters_f.as<double>(); A.energy = h->d_ters_e.as<double>();
        A.max_iter = cp->max_iter; A.max_eval = cp->max_eval; A.etol = cp->etol; A.ftol = cp->ftol; A.dmax = cp->dmax;
        A.x0 = h->d_vel.as<double>(); A.hh = A.x0 + 3 * (size_t)N; A.gg = A.x0 + 6 * (size_t)N;
        A.st = h->d_fire.as<CgState>(); A.active = h->d_active.as<unsigned char>(); A.flags = flags; A.n_evals = n_evals;
        A.max_launch = (long long)cp->max_eval + 72;   // the lock-step driver's launch budget (relax.hip), per chain here
        h->prof.begin(KC_TERSOFF, st);
        VSSR_HIP(h, hipMemcpyAsync(d_args, &A, sizeof(A), hipMemcpyHostToDevice, st));   // (pageable source: copied before the call returns)
        hipLaunchKernelGGL(k_cg_chain, dim3(B), dim3(CM_THREADS), 0, st, d_args);
        h->prof.end(st);
        VSSR_HIP(h, hipGetLastError());
        ++h->relax_lockstep;
        int over = 0;
        VSSR_HIP(h, hipMemcpyAsync(&over, flags, sizeof(int), hipMemcpyDeviceToHost, st));
        VSSR_HIP(h, hipStreamSynchronize(st));
        if (!over) break;
        // a chain needed more slots per atom than its pool holds: chains that were stopped kept their state (nothing is stepped on
        // an overflowed evaluation) and continue in the next launch with larger pools
        if (attempt >= 6) return set_err(h, VSSR_E_CAPACITY, "neighbor capacity could not be satisfied");
        cap *= 2;
        ++h->relax_regrows;
    }
    h->cm_cap_per_atom = cap;
    hipLaunchKernelGGL(k_cm_report, dim3((B + 127) / 128), dim3(128), 0, st, B, h->d_fire.as<CgState>(), h->d_relax_steps.as<int>());
    std::vector<int> ne(B);
    VSSR_HIP(h, hipMemcpyAsync(ne.data(), n_evals, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost, st));
    VSSR_HIP(h, hipStreamSynchronize(st));
    long long tot = 0;
    for (int b = 0; b < B; ++b) tot += ne[b];
    h->relax_chain_evals = tot;
    h->active_mask = nullptr;
    h->h_counters[2] = 0;      // (no batch-wide neighbor build ran: nothing for vssr_synchronize to repair)
    h->ran = true;
    h->graph_partial = true;   // (the rows are numbered per chain: the batch-wide introspection calls want one plain run first)
    return VSSR_OK;
}

}  // namespace vssr
