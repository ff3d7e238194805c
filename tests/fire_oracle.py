"""numpy restatement of ASE's FIRE (ase/optimize/fire.py) for tests: the checker of the device relaxation.
TEST INFRASTRUCTURE.  Parity status: the optimizer trajectory of the reference is not pinned by any stored output
(notebook BFGS traces are trajectory-level only), so this restatement is pinned by its
invariants (energy decrease, final fmax) and is used as a same-algorithm cross-check of the device code."""

import numpy as np


def fire_relax(force_fn, pos, fixed=None, max_steps=20, fmax=0.01, dt=0.1, maxstep=0.2, dtmax=1.0, nmin=5, finc=1.1,
               fdec=0.5, astart=0.1, fa=0.99):
    """force_fn(pos) -> (energy, forces[N,3]).  Returns (pos, energies list, n_steps, converged)."""
    pos = np.array(pos, dtype=np.float64)
    mask = np.ones(len(pos), bool)
    if fixed is not None:
        mask[np.asarray(fixed, dtype=np.int64)] = False
    v = None
    a, nsteps_pos, steps = astart, 0, 0
    energies = []
    converged = False
    for it in range(max_steps + 1):
        e, f = force_fn(pos)
        f = np.where(mask[:, None], f, 0.0)
        energies.append(e)
        if np.linalg.norm(f, axis=1).max() < fmax:
            converged = True
            break
        if it == max_steps:
            break
        if v is None:
            v = np.zeros_like(pos)
        else:
            vf = np.vdot(f, v)
            if vf > 0.0:
                v = (1.0 - a) * v + a * f / np.sqrt(np.vdot(f, f)) * np.sqrt(np.vdot(v, v))
                if nsteps_pos > nmin:
                    dt = min(dt * finc, dtmax)
                    a *= fa
                nsteps_pos += 1
            else:
                v[:] = 0.0
                a = astart
                dt *= fdec
                nsteps_pos = 0
        v = v + dt * f
        dr = dt * v
        normdr = np.sqrt(np.vdot(dr, dr))
        if normdr > maxstep:
            dr = maxstep * dr / normdr
        pos = pos + dr
        steps += 1
    return pos, energies, steps, converged
