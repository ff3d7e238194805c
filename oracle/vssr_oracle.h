/*
 * vssr_oracle.h — CPU ORACLE for the VSSR-MC energy-evaluation hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (surface-sampling_amd/csrc -> libvssr_eval.so) never links, loads or calls it.
 *
 * It restates, in plain C, the algorithm the reference reaches through its un-vendored
 * dependencies:
 *   - nff @ surface-sampling-0.3.0 (reference pyproject.toml:17): AtomsBatch neighbor list
 *     (nff/io/ase.py, call sites mcmc/utils/misc.py:34-42, mcmc/dynamics.py:129), PaiNN
 *     forward (nff/nn/models/painn.py, nff/nn/modules/painn.py, nff/nn/layers.py),
 *     EnsembleNFF.calculate (nff/io/ase_calcs.py; call site
 *     mcmc/calculators/calculators.py:484), unit constants nff/utils/constants.py.
 *   - LAMMPS pair_style tersoff (reference environment.yml:6; call site
 *     mcmc/calculators/calculators.py:507-598 with mcmc/potentials/GaN.tersoff).
 * Parity pinning: the known answers stored in the reference's notebooks
 * (tutorials/SrTiO3_001.ipynb:241, tests/test_SrTiO3_terms.ipynb:201,208,212,257,
 * tutorials/GaN_0001.ipynb:228) — checked by tests/test_oracle_kat.py.
 */
#ifndef VSSR_ORACLE_H
#define VSSR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t feat_dim;       /* F   = 128 */
    int32_t n_rbf;          /* R   = 20  */
    int32_t num_conv;       /* L   = 3   */
    int32_t n_embed;        /* rows of the embedding table (100) */
    int32_t readout_hidden; /* H   = 64  */
    int32_t excl_vol;       /* 1: add sum_e (sigma/d)^power per atom */
    int32_t excl_power;     /* 12 */
    float cutoff;           /* 5.0 Angstrom */
    float excl_sigma;       /* 1.5 Angstrom */
} orc_painn_hparams;

/* Optional per-layer intermediates (all double, caller-allocated, any pointer may be NULL).
 * Layouts: s [N][F]; v [N][3][F] (Cartesian-major, the layout the HIP path uses);
 * phi [N][3F]. Index l = layer. */
typedef struct {
    double *phi[8];      /* message MLP output entering layer l              */
    double *s_msg[8];    /* s after message block l                          */
    double *v_msg[8];    /* v after message block l                          */
    double *s_upd[8];    /* s after update block l                           */
    double *v_upd[8];    /* v after update block l                           */
    double *sbar_msg[8]; /* dE/d(s after message block l)                    */
    double *vbar_msg[8]; /* dE/d(v after message block l)                    */
    double *sbar_in[8];  /* dE/d(s entering layer l)                         */
    double *vbar_in[8];  /* dE/d(v entering layer l)                         */
    double *e_atom;      /* [N] per-atom energies (kcal/mol), incl. excl vol */
    double *edge_gbar;   /* [E][3] dE/d r_e for every directed edge          */
} orc_painn_dump;

/* Neighbor multigraph: all directed (i, j, S) with 0 < |x_j + S.cell - x_i| <= cutoff,
 * sorted by (i, j, S lexicographic).  cell rows are lattice vectors.  Returns the number of
 * edges (which may exceed cap: then only the first cap are written), or <0 on bad input. */
int64_t orc_neighbors(int32_t n, const double *pos, const double cell[9], const uint8_t pbc[3],
                      double cutoff, int64_t cap, int32_t *ei, int32_t *ej, int32_t *eS /*[cap][3]*/,
                      double *er /*[cap][3], may be NULL*/);

/* One PaiNN model, energy in the model's units (kcal/mol), gradient dE/dx [N][3].
 * real_bits = 64: all arithmetic in double; 32: arithmetic in float (positions are rounded to
 * float first, like nff's float32 nxyz), results widened to double. */
int orc_painn_eval(int real_bits, const float *blob, int64_t blob_len, const orc_painn_hparams *hp,
                   int32_t n, const int32_t *Z, const double *pos, const double cell[9],
                   const uint8_t pbc[3], double *energy, double *grad /*may be NULL*/,
                   const orc_painn_dump *dump /*may be NULL*/);

/* Ensemble of n_models blobs: mean/std (ddof=0) of energy and forces in eV, eV/Angstrom.
 * energy = mean_m(E_m)/model_to_ev_div + offset_ev, forces = -mean_m(grad_m)/model_to_ev_div.
 * offset_per_z [n_embed] (eV per atom of species Z) + offset_const are added when non-NULL. */
int orc_ensemble_eval(int real_bits, int32_t n_models, const float *const *blobs, int64_t blob_len,
                      const orc_painn_hparams *hp, double model_units_per_ev,
                      const double *offset_per_z, double offset_const,
                      int32_t n, const int32_t *Z, const double *pos, const double cell[9],
                      const uint8_t pbc[3], double *e_mean, double *e_std,
                      double *f_mean /*[N][3]*/, double *f_std /*[N][3]*/, double *e_model /*[M]*/);

/* Tersoff (LAMMPS pair_style tersoff semantics, units metal).  params: n_types^3 entries in
 * order [i][j][k], each 14 doubles: m gamma lambda3 c d costheta0 n beta lambda2 B R D lambda1 A.
 * type[i] in [0, n_types).  Outputs: total energy (eV), per-atom energies, forces (may be NULL). */
int orc_tersoff_eval(int32_t n_types, const double *params, int32_t n, const int32_t *type,
                     const double *pos, const double cell[9], const uint8_t pbc[3],
                     double *energy, double *e_atom, double *forces);

/* Number of OpenMP threads the oracle uses from now on (returns the value set). */
int orc_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
