/*
 * vssr_eval.h — C ABI of the MI355X (gfx950) energy/force evaluation backend for VSSR-MC.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference
 * has no FFI today: its ASE calculators call into nff/torch and LAMMPS.  Each entry point below
 * names the reference interface it replaces (paths relative to the reference repo):
 *
 *   vssr_create / vssr_destroy      <- EnsembleNFFSurface.__init__ + load_model x3
 *                                      (mcmc/calculators/calculators.py:366-377,
 *                                       scripts/sample_surface.py:164-175)
 *   vssr_eval / vssr_eval_batch     <- EnsembleNFFSurface.calculate -> EnsembleNFF.calculate
 *                                      (mcmc/calculators/calculators.py:468-489, :484) including
 *                                      AtomsBatch.update_nbr_list (mcmc/dynamics.py:129,
 *                                      mcmc/utils/misc.py:34-42)
 *   vssr_batch_upload / _run / _download / _set_positions
 *                                   <- the same call split into its H2D / compute / D2H parts so
 *                                      a relaxation loop (mcmc/dynamics.py:133-143) can keep the
 *                                      batch resident in HBM between force calls
 *   vssr_tersoff_create / vssr_tersoff_eval_batch
 *                                   <- LAMMMPSCalc.run_lammps_calc / run_lammps_energy with
 *                                      pair_style tersoff (mcmc/calculators/calculators.py:507-640)
 *
 * Conventions
 *   - All arrays are caller-allocated and borrowed only for the duration of the call.
 *   - Positions are double [N][3] (Angstrom), cell is double[9] with rows = lattice vectors,
 *     pbc is uint8[3].  Results are float32 (fp32 state and fp32-level arithmetic, like the reference: matrix products
 *     run as exact-split fp16 pieces with fp32 accumulation).
 *   - A batch is a list of independent configurations (Markov chains), concatenated:
 *     n_atoms[B], then Z / pos / forces concatenated in chain order.
 *   - Status codes: 0 ok, <0 error (see VSSR_E_*); vssr_last_error() gives the message.
 *     Non-finite energies are returned, not raised (the +-1000 clamp is the caller's job,
 *     mcmc/dynamics.py:159-168).
 *   - A handle is not re-entrant; distinct handles are independent (one per GPU / stream) and may be driven from
 *     different host threads at the same time (mc.ConcurrentChains does).
 *     Calls are synchronous unless stated.
 *
 * Weight blob layout (float32, little endian), F=feat_dim, R=n_rbf, H=readout_hidden:
 *   embed [n_embed][F]
 *   for l in 0..num_conv-1:
 *     msg.W1 [F][F], msg.b1 [F], msg.W2 [3F][F], msg.b2 [3F], msg.Wd [3F][R], msg.bd [3F],
 *     upd.U [F][F], upd.V [F][F], upd.W3 [F][2F], upd.b3 [F], upd.W4 [3F][F], upd.b4 [3F]
 *   readout.W5 [H][F], readout.b5 [H], readout.w6 [H], readout.b6 [1]
 * (torch Linear layout W[out][in]; nff state-dict keys in surface-sampling_amd/checkpoint.py).
 */
#ifndef VSSR_EVAL_H
#define VSSR_EVAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSSR_ABI_VERSION 1

enum {
    VSSR_OK = 0,
    VSSR_E_BADARG = -1,
    VSSR_E_DEVICE = -2,   /* HIP runtime error */
    VSSR_E_CAPACITY = -3, /* neighbor capacity exceeded even after regrow / LDS limits */
    VSSR_E_NOMEM = -4,
    VSSR_E_STATE = -5     /* call sequence error (e.g. run before upload) */
};

/* `want` bitmask */
enum {
    VSSR_WANT_ENERGY = 1u,
    VSSR_WANT_FORCES = 2u,
    VSSR_WANT_STD = 4u,       /* ensemble std of energy / forces */
    VSSR_WANT_PER_MODEL = 8u, /* per-model energies */
    VSSR_WANT_PER_ATOM = 16u  /* per-atom energies (Tersoff: pe/atom) */
};

typedef struct vssr_handle vssr_handle;

typedef struct {
    uint32_t struct_size; /* sizeof(vssr_painn_config), for forward compatibility */
    int32_t device;       /* HIP device ordinal */
    /* ensemble */
    int32_t n_models;
    const float *const *weights; /* n_models blobs in the layout above */
    uint64_t weights_len;        /* floats per blob */
    /* PaiNN hyper-parameters (params.json of the reference checkpoints) */
    int32_t feat_dim;       /* 128 (the only compiled value) */
    int32_t n_rbf;          /* 20 */
    int32_t num_conv;       /* 3 */
    int32_t n_embed;        /* rows of the embedding table, 100 */
    int32_t readout_hidden; /* 64 */
    float cutoff;           /* 5.0 A */
    int32_t excl_vol;       /* 1 -> add sum_e (excl_sigma/d_e)^excl_power */
    int32_t excl_power;     /* 12 */
    float excl_sigma;       /* 1.5 A */
    /* EnsembleNFF unit handling: E_eV = E_model / model_units_per_ev + offset */
    double model_units_per_ev;  /* 23.0605 for kcal/mol models */
    const double *offset_per_z; /* [n_embed] eV per atom of species Z, or NULL */
    double offset_const;        /* eV per structure (applied only when offset_per_z != NULL) */
} vssr_painn_config;

typedef struct {
    float *energy;        /* [B]            ensemble mean, eV (incl. offset)            */
    float *energy_std;    /* [B]            population std over models (WANT_STD)       */
    float *forces;        /* [sum N][3]     -mean gradient, eV/A (WANT_FORCES)          */
    float *forces_std;    /* [sum N][3]     (WANT_FORCES|WANT_STD)                      */
    float *energy_models; /* [B][n_models]  (WANT_PER_MODEL)                            */
    float *energy_atoms;  /* [sum N]        per-atom energies where defined (PER_ATOM)  */
} vssr_out;

/* ---- PaiNN ensemble ------------------------------------------------------------------- */
int vssr_abi_version(void);
/* Environment variables -- test / measurement hooks, none is needed in production (eleven; the knobs of experiments that were
 * measured and dropped are gone with their code: profiles/EXPERIMENTS.md).  Read once by vssr_create:
 *   VSSR_EDGE_IMPL=gather    every chain takes the gather neighbor kernels (reference path of the parity tests)
 *   VSSR_L0_FACTORISE=0      layer 0 runs the generic kernels instead of the species factorisation
 *   VSSR_EDGE_FS16_MAX=n, VSSR_EDGE_FS8_MAX=n   largest chain (atoms) served by the single-pass 16- / 8-feature-slice kernels
 *                            (tests: lower = force the path of larger chains onto small structures)
 *   VSSR_EDGE_FWD_2PASS=0|8|16, VSSR_EDGE_BWD_MPASS=0|1|2, VSSR_EDGE_SUB_CHUNK=n
 *                            large chains (forward: > 405 atoms, reverse: > 557): 16-feature slices in several passes over sub-ranges of
 *                            the chain's neighbors (the default) instead of the narrower single-pass kernels (FWD_2PASS=0 /
 *                            BWD_MPASS=0); 8: the forward multi-pass form on 8-feature slices; BWD_MPASS=2 + SUB_CHUNK=n (tests):
 *                            every chain takes the multi-pass forms, ranges of n atoms
 *   VSSR_UPD_SAVE=1          update blocks store their forward intermediates for the reverse pass (measured: no gain; kept because
 *                            the parity tests use it as an independent second path through the reverse update kernel)
 *   VSSR_DEBUG_KEEP=1        materialise buffers that only vssr_debug_read consumes (the last block's vector output)
 * Read by every vssr_batch_relax_cg call:
 *   VSSR_CG_FUSED=0|1        0: always the lock-step driver (one batch-wide evaluation per launch sequence); 1: the chain-resident
 *                            minimiser (one workgroup relaxes one chain from start to stop, csrc/chain_min.hip) whenever it applies
 *                            (Tersoff handles, chains of <= 256 atoms); unset: chain-resident for batches of <= 3 072 chains of <= 64
 *                            atoms.  Same results bit for bit either way
 *   VSSR_RELAX_COMPACT=0     no live-chain compaction of the resident batch (default on for resident batches of >= 65 536 atoms: once
 *                            at most 3/4 of the chains are still minimising, the batch continues as a smaller one; same trajectories
 *                            bit for bit); n > 1: compact batches of >= n atoms (tests: 2 = always) */
int vssr_create(const vssr_painn_config *cfg, vssr_handle **out);
void vssr_destroy(vssr_handle *h);
const char *vssr_last_error(const vssr_handle *h); /* h may be NULL: last create() error of the calling thread */

/* One configuration (what one ASE calculate() call is). */
int vssr_eval(vssr_handle *h, int32_t n_atoms, const int32_t *Z, const double *pos,
              const double cell[9], const uint8_t pbc[3], uint32_t want, vssr_out *out);

/* B independent configurations in one lock-step evaluation (upload + run + download). */
int vssr_eval_batch(vssr_handle *h, int32_t n_cfg, const int32_t *n_atoms, const int32_t *Z,
                    const double *pos, const double *cell /*[B][9]*/, const uint8_t *pbc /*[B][3]*/,
                    uint32_t want, vssr_out *out);

/* The same, split: keep the batch resident in HBM across a relaxation loop. */
int vssr_batch_upload(vssr_handle *h, int32_t n_cfg, const int32_t *n_atoms, const int32_t *Z,
                      const double *pos, const double *cell, const uint8_t *pbc);
int vssr_batch_set_positions(vssr_handle *h, const double *pos /*[sum N][3]*/);
int vssr_batch_run(vssr_handle *h, uint32_t want); /* asynchronous on the handle's stream */
int vssr_batch_download(vssr_handle *h, uint32_t want, vssr_out *out); /* synchronises */
int vssr_synchronize(vssr_handle *h);

/* ---- lock-step relaxation of the resident batch (reference optimize_slab, mcmc/dynamics.py:83-170) ---- */
typedef struct {
    int32_t max_steps; /* relax_steps (reference default 20) */
    float fmax;        /* convergence: max_i |F_i| < fmax (reference: 0.01 eV/A) */
    /* ASE FIRE parameters (ase/optimize/fire.py defaults: 0.1, 0.2, 1.0, 1.1, 0.5, 0.1, 0.99, 5) */
    float dt, maxstep, dtmax, finc, fdec, astart, fa;
    int32_t nmin;
} vssr_fire_params;
/* FIRE-relaxes every chain of the resident batch in lock step (forces and positions stay in HBM).
 * fixed: [sum N] 1 = atom held by FixAtoms (force zeroed), NULL = all free.  Afterwards the batch holds the
 * relaxed positions and the results of the last evaluation: fetch them with vssr_batch_download;
 * pos_out [sum N][3], n_steps [B], converged [B] may each be NULL. */
int vssr_batch_relax_fire(vssr_handle *h, const vssr_fire_params *params, const uint8_t *fixed, uint32_t want,
                          double *pos_out, int32_t *n_steps, uint8_t *converged);

/* ASE BFGS (ase/optimize/bfgs.py: alpha = 70 eV/A^2, maxstep = 0.2 A) -- the optimizer of the reference's SrTiO3
 * configuration (scripts/configs/sample_config_painn.json:26 "optimizer": "BFGS", dispatched at mcmc/dynamics.py:119-141).
 * Same contract as vssr_batch_relax_fire.  The per-chain Hessian is kept in factored form H = alpha I + Q B Q^T; its
 * on-chip workspace holds 46 updates: identical to ASE up to step 46, later steps keep that Hessian (csrc/relax.hip). */
typedef struct {
    int32_t max_steps; /* relax_steps (reference: 20) */
    float fmax;        /* 0.01 eV/A */
    float alpha;       /* initial Hessian, 70 eV/A^2 */
    float maxstep;     /* longest atomic displacement per step, 0.2 A */
} vssr_bfgs_params;
int vssr_batch_relax_bfgs(vssr_handle *h, const vssr_bfgs_params *params, const uint8_t *fixed, uint32_t want,
                          double *pos_out, int32_t *n_steps, uint8_t *converged);

/* Trajectory recording during vssr_batch_relax_fire / _bfgs -- the reference's TrajectoryObserver attached with
 * `dyn.attach(obs, interval=record_interval)` (mcmc/dynamics.py:20-80,131-151; optimize_slab(save_traj=True, record_interval=5)).
 * record_interval > 0 arms it for the following relaxations of this handle (0 = off, the default): every chain records its
 * positions, forces (FixAtoms applied) and energy after 0, k, 2k, ... optimizer steps, like ASE calls the observer.
 * vssr_batch_traj_read: max_records (may be NULL) = relax_steps / k + 1 of the last relaxation; with buffers of
 * cap_records >= max_records: n_records [B] = valid records of every chain, pos [R][sum N][3], forces [R][sum N][3],
 * energy [R][B] (record-major; entries of a chain beyond its n_records are undefined).  Any pointer may be NULL. */
int vssr_batch_traj_configure(vssr_handle *h, int32_t record_interval);
int vssr_batch_traj_read(vssr_handle *h, int32_t cap_records, int32_t *n_records, double *pos, float *forces, double *energy,
                         int32_t *max_records);

/* LAMMPS `min_style cg` + `minimize etol ftol maxiter maxeval` for the analytic (fp64) potentials -- the reference relaxes
 * GaN with it ("optimizer": "LAMMPS": mcmc/dynamics.py:107-116 -> LAMMMPSCalc.run_lammps_opt, mcmc/calculators/calculators.py:600-619,
 * template tutorials/data/GaN_0001/GaN_0001_lammps_opt_template.txt: `fix 2 bulk setforce 0 0 0`, `min_style cg`,
 * `minimize 1e-5 1e-5 {relax_steps} 10000`).  Polak-Ribiere conjugate gradients with LAMMPS' quadratic line search
 * (dmax 0.1 A per coordinate and line search), restated from LAMMPS min_cg.cpp / min_linesearch.cpp; Tersoff and EAM
 * handles only (fp64 energies drive the line search).  stop_reason [B] (may be NULL): 1 energy tolerance, 2 force tolerance,
 * 3 maxiter, 4 maxeval, 5 search direction not downhill, 6 zero force, 7 zero quadratic step, 8 zero alpha. */
typedef struct {
    int32_t max_iter;  /* relax_steps (reference GaN: 100) */
    int32_t max_eval;  /* 10000 */
    double etol, ftol; /* 1e-5, 1e-5 */
    double dmax;       /* 0.1 */
} vssr_cg_params;
/* Two drivers, same results bit for bit (csrc/chain_min.hip, csrc/relax.hip; VSSR_CG_FUSED / VSSR_RELAX_COMPACT above): Tersoff batches
 * of <= 3 072 chains of <= 256 atoms are minimised by ONE workgroup per chain from the first evaluation to the stop criterion (no
 * lock step: the GaN chains of the reference stop after 21 .. 159 evaluations each); everything else in lock step, with the resident
 * batch compacted to the chains still minimising once it is large enough for that to pay.  The automatic choice takes the
 * chain-resident driver for batches of <= 3 072 chains of <= 64 atoms (the measured regime); VSSR_CG_FUSED=1 / 0 forces one.
 * State of the handle afterwards: positions, energies, per-atom energies and forces are those of the minimised geometries with
 * either driver.  The resident neighbor GRAPH differs: the lock-step driver leaves the batch-wide graph of the final evaluation
 * (vssr_batch_stats / vssr_batch_neighbors work at once), the chain-resident driver numbers its rows per chain and leaves no
 * batch-wide graph -- those two calls return VSSR_E_STATE until one vssr_batch_run has been made. */
int vssr_batch_relax_cg(vssr_handle *h, const vssr_cg_params *params, const uint8_t *fixed, uint32_t want,
                        double *pos_out, int32_t *n_iter, int32_t *n_eval, int32_t *stop_reason);

/* ---- introspection used by tests and bench (no effect on results) ---------------------- */
/* Per-kernel timing with HIP events on the handle's own stream.  enable=1 starts recording;
 * vssr_profile_read synchronises and returns, for each kernel class, the number of launches and
 * the summed duration since the last reset.  names[i] points to a static string. */
int vssr_profile_enable(vssr_handle *h, int enable);
int vssr_profile_reset(vssr_handle *h);
int vssr_profile_read(vssr_handle *h, int32_t cap, const char **names, int64_t *launches,
                      double *total_ms, int32_t *n_out);
/* Workload counters of the resident batch after a run: atoms, directed edges (unpadded),
 * padded edge slots.  After vssr_batch_relax_fire / _bfgs the resident graph and activations cover only the chains that
 * were still running in the last iteration: vssr_batch_stats, vssr_batch_neighbors and vssr_debug_read then return
 * VSSR_E_STATE until the batch has been run once more (vssr_batch_run); results (download) are complete at all times. */
int vssr_batch_stats(vssr_handle *h, int64_t *n_atoms, int64_t *n_edges, int64_t *n_slots);
/* Neighbor multigraph of the resident batch (after a run): for every directed edge its centre i,
 * neighbor j (global atom indices) and image shift S.  Returns the edge count through n_edges;
 * arrays may be NULL to query the size. */
int vssr_batch_neighbors(vssr_handle *h, int64_t cap, int32_t *ei, int32_t *ej, int32_t *eS,
                         float *er, int64_t *n_edges);
/* Device addresses of the per-chain results of the resident batch (energy [B], energy_std [B], fp32, valid after a
 * synchronised run): lets the multi-GPU result gather (RCCL all_gather of per-chain scalars) read
 * them in place instead of through the host.  The pointers stay valid until the next vssr_batch_upload. */
int vssr_batch_device_results(vssr_handle *h, const float **energy, const float **energy_std);
/* The same values as doubles (the device holds the ensemble mean / spread in fp64 before narrowing them to the float32 result
 * word): what the result gather of sharding.py moves between GPUs. */
int vssr_batch_device_results_f64(vssr_handle *h, const double **energy, const double **energy_std);
/* Energies of the LAST evaluation of the resident batch without the float32 output word (synchronises).  The reference's
 * results["energy"] is a float32 tensor (EnsembleNFF.calculate, mcmc/calculators/calculators.py:484) and vssr_out keeps that
 * type; but the per-chain sum over atoms, the unit conversion, the stoichiometric offset and the mean / population spread over
 * the models are formed in fp64 on the device, and at |E| ~ 2 ... 9 keV the spacing of float32 (1.2e-4 ... 4.9e-4 eV) is as
 * large as the whole arithmetic error of the evaluation.  energy [B], energy_std [B], energy_models [B][n_models]; any pointer
 * may be NULL.  Tersoff / EAM handles: energy = energy_models = the fp64 energy, energy_std = 0.  The Metropolis test of the
 * batched MC loop (mc.py) and relax_batch's returned energy take these values. */
int vssr_batch_energy_f64(vssr_handle *h, double *energy, double *energy_std, double *energy_models);
/* Latent-space embedding: the per-atom scalar features after the last update block, [sum N][feat_dim] fp32 per model --
 * what nff's Painn returns as results["embedding"] with requires_embedding=True and the reference's clustering /
 * uncertainty helpers read (get_embeddings_single, mcmc/calculators/calculators.py:67-93; scripts/clustering.py:239).
 * model >= 0: that ensemble member; model = -1: all members, model-major [M][sum N][feat_dim].  Valid after a run of the
 * resident batch; dst may be NULL to query the size through n_out. */
int vssr_batch_embedding(vssr_handle *h, int32_t model, float *dst, int64_t cap, int64_t *n_out);
/* Range guard of the PaiNN path.  The dense contractions run as exact 2-way fp16 splits: an
 * activation or adjoint beyond +-65504 cannot be represented and is clamped -- the results stay finite but are no longer
 * the model's (seen with weights scaled far outside the trained regime).  The kernels track the largest magnitude they
 * split; flags [B] (may be NULL) receives 1 for every chain whose LAST evaluation clamped a value or produced a non-finite
 * energy, n_flagged (may be NULL) their count.  The reference has no counterpart (fp32 torch arithmetic overflows at 3e38);
 * callers treat a flagged chain like the out-of-bounds energies of mcmc/dynamics.py:159-168.  Tersoff / EAM: always 0. */
int vssr_batch_saturated(vssr_handle *h, uint8_t *flags, int32_t *n_flagged);
/* Virial stress of every chain of the resident batch from its LAST evaluation (which must have produced forces): the
 * "stress" property that nff's EnsembleNFF / the reference's EnsembleNFFSurface list in implemented_properties
 * (mcmc/calculators/calculators.py:369) and ASE's Atoms.get_stress() asks a calculator for.  Nothing is re-evaluated: the
 * reverse pass leaves dE/d r for every directed edge on the device, and sigma_ab = (1/V) sum_edges (dE/d r_a) r_b.
 * stress, stress_std (may be NULL): [B][6] fp64, Voigt order xx yy zz yz xz xy, eV / A^3, ASE's sign convention; ensemble
 * mean and population standard deviation over the models.  V = |det cell| (also for slabs with a vacuum axis, as ASE).
 * PaiNN handles only (VSSR_E_STATE otherwise, after an energies-only run, or after a relaxation that left a partial graph). */
int vssr_batch_stress(vssr_handle *h, double *stress, double *stress_std);
/* The handle's HIP device ordinal, its stream (hipStream_t: every kernel of the handle is enqueued there) and the device
 * address of the neighbor-capacity overflow flag of the last run (int32, non-zero = the run's results are void and
 * vssr_synchronize will repeat it with grown buffers; NULL before the first run).  For consumers that order their own device
 * work behind an evaluation with events instead of a host synchronisation (the multi-GPU result gather, sharding.py). */
int vssr_device_context(vssr_handle *h, int32_t *device, void **stream, const int32_t **overflow_flag);
/* Test hook for the capacity-regrow paths: initial neighbor capacity in slots per atom (<= 0: unchanged), tight != 0:
 * regrow to the exact need only (every later growth of the edge count overflows again), tight < 0: unchanged;
 * n_regrows (may be NULL) receives the number of regrows of the last relaxation. */
int vssr_debug_capacity(vssr_handle *h, int32_t slots_per_atom, int32_t tight, int32_t *n_regrows);
/* Work counters of the LAST relaxation of this handle (vssr_batch_relax_fire / _bfgs / _cg): lockstep_evaluations = evaluations of
 * the batch the driver launched (the final static evaluation included); chain_evaluations = chain-evaluations those launches
 * dispatched (a launch over all B chains counts B even when converged chains leave their kernels at once).  Together with the
 * per-chain counts the relaxation returns (n_steps / n_eval) they give the lock-step waste: dispatched / needed. */
int vssr_batch_relax_counts(vssr_handle *h, int64_t *lockstep_evaluations, int64_t *chain_evaluations);
/* Copy a named device intermediate of model m (fp32) to host; for parity debugging.
 * Names: "phi<l>", "s_msg<l>", "v_msg<l>", "s_upd<l>", "v_upd<l>", "sbar_msg<l>", "vbar_msg<l>",
 * "e_atom".  Layouts: s [N][F], v [N][3][F], phi [N][3F]. */
int vssr_debug_read(vssr_handle *h, const char *name, int32_t model, float *dst, int64_t cap,
                    int64_t *n_out);

/* ---- EAM (Cu(100) toy config, BASELINE configs[0]) ------------------------------------------------------ */
/* One-element funcfl tables of LAMMPS `pair_style eam` (reference: LAMMPSRunSurfCalc + mcmc/potentials/Cu_u3.eam,
 * mcmc/calculators/calculators.py:755-811, tests/test_Cu.py:41): frho[nrho] embedding energy F(rho) in eV on the grid
 * rho = k drho; zr[nr] effective charge Z(r) and rhor[nr] density rho(r) on r = k dr; pair term
 * phi(r) = 27.2 * 0.529 * Z(r)^2 / r.  Evaluate with vssr_tersoff_eval_batch / vssr_eam_eval_batch (all types 0). */
typedef struct {
    int32_t nrho, nr;
    double drho, dr, cutoff;
} vssr_eam_grid;
int vssr_eam_create(int32_t device, const vssr_eam_grid *grid, const double *frho, const double *zr, const double *rhor,
                    vssr_handle **out);
/* same signature and meaning as vssr_tersoff_eval_batch (fp64 energies / per-atom energies / forces) */
int vssr_eam_eval_batch(vssr_handle *h, int32_t n_cfg, const int32_t *n_atoms, const int32_t *type, const double *pos,
                        const double *cell, const uint8_t *pbc, uint32_t want, vssr_out *out, double *energy_f64,
                        double *energy_atoms_f64, double *forces_f64);

/* ---- Tersoff (GaN config) ---------------------------------------------------------------- */
/* params: n_types^3 entries ordered [i][j][k], 14 doubles each, LAMMPS column order
 * (m gamma lambda3 c d costheta0 n beta lambda2 B R D lambda1 A). */
int vssr_tersoff_create(int32_t device, int32_t n_types, const double *params, vssr_handle **out);
/* The same from the text of a LAMMPS tersoff potential file and the species in LAMMPS type order -- what the reference
 * passes to LAMMPS as `pair_coeff * * <file> Ga N` (mcmc/calculators/calculators.py:559-568).
 * Entries `e1 e2 e3 m gamma lambda3 c d costheta0 n beta lambda2 B R D lambda1 A` may span lines, `#` starts a comment;
 * every triplet of the given species must be present. */
int vssr_tersoff_create_from_text(int32_t device, const char *param_text, int32_t n_species, const char *const *species,
                                  vssr_handle **out);
/* type[i] in [0,n_types).  Fills out->energy (total, eV), out->energy_atoms (WANT_PER_ATOM),
 * out->forces (WANT_FORCES).  fp64 arithmetic on the device; results narrowed to fp32 in
 * vssr_out, and returned exactly through the optional double arrays. */
int vssr_tersoff_eval_batch(vssr_handle *h, int32_t n_cfg, const int32_t *n_atoms,
                            const int32_t *type, const double *pos, const double *cell,
                            const uint8_t *pbc, uint32_t want, vssr_out *out,
                            double *energy_f64 /*[B] or NULL*/, double *energy_atoms_f64,
                            double *forces_f64);

#ifdef __cplusplus
}
#endif
#endif /* VSSR_EVAL_H */
