"""numpy restatement of ASE's BFGS (ase/optimize/bfgs.py, the optimizer of the reference's SrTiO3 configuration:
scripts/configs/sample_config_painn.json:26, dispatched at mcmc/dynamics.py:119-141) for tests: the checker of the
device relaxation.  TEST INFRASTRUCTURE.

Parity status: PINNED.  Driven by the CPU oracle (fp32 mode, like nff) it reproduces the (energy, fmax) traces stored in
the reference's notebooks (tests/test_SrTiO3_terms.ipynb:201-227: 6 + 3 + 14 points; tutorials/SrTiO3_001.ipynb:241-245:
5 points), committed as tests/golden/bfgs_traces.json -- see tests/test_bfgs.py for the tolerances.

ASE's algorithm, restated: H0 = alpha * I (alpha = 70 eV/A^2), maxstep = 0.2 A.  Every step: BFGS update of H from
(dr, df) of the previous step (skipped when max|dr| < 1e-7), eigen-decomposition H = V diag(w) V^T, step
dr = V (V^T f / |w|), rescaled so that the longest atomic displacement is at most maxstep.  Convergence: max_i |F_i| < fmax,
tested before every step; at most `max_steps` steps.  FixAtoms zeroes the forces of the held atoms and keeps their
positions; H stays block diagonal, so restricting it to the free atoms is exact."""

import numpy as np


def bfgs_relax(force_fn, pos, fixed=None, max_steps=20, fmax=0.01, alpha=70.0, maxstep=0.2):
    """force_fn(pos) -> (energy, forces[N,3]).  Returns (pos, trace [(energy, fmax)], n_steps, converged)."""
    pos = np.array(pos, dtype=np.float64)
    n = len(pos)
    free = np.ones(n, bool)
    if fixed is not None and len(fixed):
        free[np.asarray(fixed, dtype=np.int64)] = False
    idx = np.where(free)[0]
    H = None
    r0 = f0 = None
    trace = []
    steps, converged = 0, False
    for it in range(max_steps + 1):
        e, f = force_fn(pos)
        f = np.asarray(f, dtype=np.float64)[idx]
        fm = float(np.sqrt((f ** 2).sum(axis=1).max())) if len(idx) else 0.0
        trace.append((float(e), fm))
        if fm < fmax:
            converged = True
            break
        if it == max_steps:
            break
        r = pos[idx].reshape(-1)
        fv = f.reshape(-1)
        if H is None:
            H = np.eye(3 * len(idx)) * alpha
        else:
            dr = r - r0
            if np.abs(dr).max() >= 1e-7:
                df = fv - f0
                a = np.dot(dr, df)
                dg = np.dot(H, dr)
                b = np.dot(dr, dg)
                H = H - np.outer(df, df) / a - np.outer(dg, dg) / b
        omega, V = np.linalg.eigh(H)
        step = np.dot(V, np.dot(fv, V) / np.fabs(omega)).reshape(-1, 3)
        longest = np.sqrt((step ** 2).sum(axis=1)).max()
        if longest >= maxstep:
            step *= maxstep / longest
        r0, f0 = r.copy(), fv.copy()
        pos[idx] += step
        steps += 1
    return pos, trace, steps, converged
