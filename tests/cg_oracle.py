"""numpy restatement of LAMMPS ``min_style cg`` with its default quadratic line search (LAMMPS min_cg.cpp,
min_linesearch.cpp ``linemin_quadratic``), the minimiser behind the reference's ``optimizer: "LAMMPS"``
(``mcmc/calculators/calculators.py:600-619``; template ``tutorials/data/GaN_0001/GaN_0001_lammps_opt_template.txt``:
``fix 2 bulk setforce 0 0 0``, ``min_style cg``, ``minimize 1e-5 1e-5 {steps} 10000``).  TEST INFRASTRUCTURE: the checker of
vssr_batch_relax_cg.

Parity status: UNPINNED.  LAMMPS is not installable here and the reference stores no relaxed GaN energy (the tutorial prints
-144.059 for the pristine slab, which the static energy matches to the printed digits), so the algorithm is restated from the
LAMMPS sources from memory: Polak-Ribiere direction with restart every ndof iterations and when not downhill; line search
with alphamax = min(1, dmax / max|h|), secant projection when the quadratic model fits (QUADRATIC_TOL 0.1), backtracking
(ALPHA_REDUCE 0.5, BACKTRACK_SLOPE 0.4), EMACH 1e-8, EPS_QUAD 1e-28; stop on relative energy change (etol), |f|^2 < ftol^2,
maxiter, maxeval, or a failed line search.  It is pinned by invariants only (monotone energy, agreement of the minimum with
BFGS / FIRE) and serves as a same-algorithm cross-check of the device code."""

import numpy as np

REASONS = {1: "energy tolerance", 2: "force tolerance", 3: "max iterations", 4: "max force evaluations",
           5: "search direction is not downhill", 6: "forces are zero", 7: "linesearch: zero quadratic step",
           8: "linesearch alpha is zero"}


def cg_minimize(force_fn, pos, fixed=None, max_iter=100, max_eval=10000, etol=1e-5, ftol=1e-5, dmax=0.1):
    """force_fn(pos) -> (energy, forces [N,3]).  Returns (pos, energy, n_iter, n_eval, stop_reason, energies of the
    accepted points)."""
    x = np.array(pos, dtype=np.float64).reshape(-1)
    free = np.ones(len(x) // 3, bool)
    if fixed is not None and len(fixed):
        free[np.asarray(fixed, dtype=np.int64)] = False
    mask = np.repeat(free, 3)
    neval = 0

    def ef(xx):
        nonlocal neval
        neval += 1
        e, f = force_fn(xx.reshape(-1, 3))
        return float(e), np.where(mask, np.asarray(f, float).reshape(-1), 0.0)

    ecur, f = ef(x)
    neval = 0   # LAMMPS zeroes the counter after the setup evaluation: only alpha_step() evaluations count
    g, h = f.copy(), f.copy()
    gg = f @ f
    ndof = len(x)
    trace = [ecur]
    niter = 0
    for _ in range(max_iter):
        niter += 1
        eprev = ecur
        # ---- linemin_quadratic ----
        fdothall = f @ h
        if fdothall <= 0.0:
            return x.reshape(-1, 3), ecur, niter, neval, 5, trace
        hmax = np.abs(h).max()
        if hmax == 0.0:
            return x.reshape(-1, 3), ecur, niter, neval, 6, trace
        alphamax = min(1.0, dmax / hmax)
        x0, eorig = x.copy(), ecur
        alpha, alphaprev, fhprev, engprev = alphamax, 0.0, fdothall, eorig
        fail = 0
        while True:
            x = x0 + alpha * h
            ecur, f = ef(x)
            fh = f @ h
            delfh = fh - fhprev
            if abs(fh) < 1e-28 or abs(delfh) < 1e-28:
                fail = 7
                break
            relerr = abs(1.0 - (0.5 * (alpha - alphaprev) * (fh + fhprev) + ecur) / engprev)
            alpha0 = alpha - (alpha - alphaprev) * fh / delfh
            if relerr <= 0.1 and 0.0 < alpha0 < alphamax:
                x = x0 + alpha0 * h
                ecur, f = ef(x)
                if ecur - eorig < 1e-8:
                    break
            de_ideal = -0.4 * alpha * fdothall
            de = ecur - eorig
            if de <= de_ideal:
                break
            fhprev, engprev, alphaprev = fh, ecur, alpha
            alpha *= 0.5
            if alpha <= 0.0 or de_ideal >= -1e-8:
                fail = 8
                break
        if fail:
            x = x0
            ecur, f = ef(x)
            return x.reshape(-1, 3), ecur, niter, neval, fail, trace
        trace.append(ecur)
        if neval >= max_eval:   # tested once per iteration, after the line search (min_cg.cpp iterate)
            return x.reshape(-1, 3), ecur, niter, neval, 4, trace
        if abs(ecur - eprev) < etol * 0.5 * (abs(ecur) + abs(eprev) + 1e-8):
            return x.reshape(-1, 3), ecur, niter, neval, 1, trace
        d0, d1 = f @ f, f @ g
        if d0 < ftol * ftol:
            return x.reshape(-1, 3), ecur, niter, neval, 2, trace
        beta = max(0.0, (d0 - d1) / gg)
        if (niter + 1) % ndof == 0:
            beta = 0.0
        gg = d0
        g = f.copy()
        h = g + beta * h
        if g @ h <= 0.0:
            h = g.copy()
    return x.reshape(-1, 3), ecur, niter, neval, 3, trace
