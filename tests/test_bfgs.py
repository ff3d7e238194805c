"""The BFGS restatement (tests/bfgs_oracle.py) driven by the CPU oracle against the relaxation traces the reference stores
(tests/golden/bfgs_traces.json <- tests/test_SrTiO3_terms.ipynb:201-227, tutorials/SrTiO3_001.ipynb:241-245), and the device
BFGS (vssr_batch_relax_bfgs) against the same traces."""

import json
import os

import numpy as np
import pytest

from bfgs_oracle import bfgs_relax
from conftest import GOLDEN, top_layer

# Tolerances.  The reference prints fp32 numbers of an fp32 model; the fp32 oracle reproduces the printed energies to
# 1.5e-4 eV (offset-dominated, as for the single-point KATs) and fmax to 2e-5 eV/A over the first points of every trace.
# A relaxation amplifies rounding differences (different summation orders -> slightly different steps), so late points of
# long traces get a looser bound: the 14-point trace ends within 4e-4 eV / 4e-3 eV/A.
E_TOL = 2e-4
F_TOL_EARLY, F_TOL_LATE = 3e-5, 5e-3
E_TOL_LATE = 5e-4


def _traces():
    with open(os.path.join(GOLDEN, "bfgs_traces.json")) as fh:
        return json.load(fh)


def _free_fixed(golden, case):
    s = golden.structure(case["structure"])
    free = top_layer(s) if case["free_atoms"] == "top_layer" else np.array(case["free_atoms"])
    return s, free, np.setdiff1d(np.arange(len(s)), free)


def check_trace(trace, ref, name):
    """Same length (the optimizer stops at the same step) and point-wise agreement."""
    assert len(trace) == len(ref), (name, len(trace), len(ref))
    for k, ((e, f), (er, fr)) in enumerate(zip(trace, ref)):
        late = k >= 6
        assert abs(e - er) <= (E_TOL_LATE if late else E_TOL), (name, k, e, er)
        assert abs(f - fr) <= (F_TOL_LATE if late else F_TOL_EARLY), (name, k, f, fr)


def test_bfgs_restatement_reproduces_reference_traces(golden, oracle_mod):
    T = _traces()
    table, const = golden.offset_table()
    n_points = 0
    for case in T["cases"]:
        s, free, fixed = _free_fixed(golden, case)

        def fn(pos):
            r = oracle_mod.ensemble(golden.blobs, s.numbers, pos, s.cell, s.pbc, 32, table, const)
            return r["energy"], r["forces"]

        pos, trace, steps, conv = bfgs_relax(fn, s.positions, fixed=fixed, max_steps=T["relax_steps"], fmax=T["fmax"])
        check_trace(trace, case["trace"], case["structure"])
        assert conv and steps == len(case["trace"]) - 1
        assert np.array_equal(pos[fixed], s.positions[fixed])        # FixAtoms
        n_points += len(trace)
    assert n_points == 28


def test_bfgs_restatement_on_a_quadratic():
    """Exact arithmetic check of the update: on a quadratic bowl BFGS with exact steps converges and H approaches the
    true Hessian along the visited directions."""
    rng = np.random.default_rng(0)
    A = rng.normal(size=(6, 6))
    K = A @ A.T + 6 * np.eye(6)
    x0 = rng.normal(size=(2, 3)) * 0.05

    def fn(pos):
        x = pos.reshape(-1)
        return 0.5 * x @ K @ x, -(K @ x).reshape(-1, 3)

    pos, trace, steps, conv = bfgs_relax(fn, x0, max_steps=60, fmax=1e-6)
    assert conv and steps < 40 and np.abs(pos).max() < 1e-6
    e = [t[0] for t in trace]
    assert all(b <= a + 1e-15 for a, b in zip(e, e[1:]))
