"""numpy restatement of LAMMPS ``pair_style eam`` for ONE funcfl element (energy, per-atom energies, forces).

TEST INFRASTRUCTURE ONLY (the checker of csrc/eam.hip).  The algorithm lives in the LAMMPS binary the reference drives
through ``LAMMPSRunSurfCalc`` (``mcmc/calculators/calculators.py:755-811``; conda package, unpinned, ``environment.yml:6``);
it is restated from LAMMPS ``pair_eam.cpp`` (``file2array`` for a single funcfl file, ``array2spline`` / ``interpolate``,
``compute``):  E = sum_i F(rho_i) + 1/2 sum_{i != j} phi(r_ij), rho_i = sum_j rho(r_ij), phi = z2r / r,
z2r = 27.2 * 0.529 * Z(r)^2; cubic splines with LAMMPS' finite-difference slopes, rho beyond the table continued linearly.
Parity status: PINNED by the reference's numbers -- minimum energy -25.2893 of tests/test_Cu.py:19 (one Cu adatom on a bridge
site of the 8-atom Cu(100) slab) and the energies printed in tutorials/example.ipynb (-24.740 = two adatoms); bulk fcc Cu
gives the potential's cohesive energy -3.54 eV (tests/test_eam.py).
"""

from __future__ import annotations

import itertools

import numpy as np


def build_spline(f, delta):
    """LAMMPS ``PairEAM::interpolate``: rows 1..n of 7 coefficients (row 0 unused)."""
    f = np.asarray(f, float)
    n = len(f)
    s = np.zeros((n + 1, 7))
    s[1:, 6] = f
    s[1, 5] = s[2, 6] - s[1, 6]
    s[2, 5] = 0.5 * (s[3, 6] - s[1, 6])
    s[n - 1, 5] = 0.5 * (s[n, 6] - s[n - 2, 6])
    s[n, 5] = s[n, 6] - s[n - 1, 6]
    m = np.arange(3, n - 1)
    s[m, 5] = ((s[m - 2, 6] - s[m + 2, 6]) + 8.0 * (s[m + 1, 6] - s[m - 1, 6])) / 12.0
    m = np.arange(1, n)
    s[m, 4] = 3.0 * (s[m + 1, 6] - s[m, 6]) - 2.0 * s[m, 5] - s[m + 1, 5]
    s[m, 3] = s[m, 5] + s[m + 1, 5] - 2.0 * (s[m + 1, 6] - s[m, 6])
    s[1:, 2] = s[1:, 5] / delta
    s[1:, 1] = 2.0 * s[1:, 4] / delta
    s[1:, 0] = 3.0 * s[1:, 3] / delta
    return s


def spline_eval(s, x, delta, n, clamp_lo=False):
    """Value and derivative at ``x`` (array): p = x / delta + 1, m = int(p) clamped to [1 or -, n - 1], p = min(p - m, 1)."""
    p = np.asarray(x, float) / delta + 1.0
    m = p.astype(np.int64)
    m = np.clip(m, 1, n - 1) if clamp_lo else np.minimum(m, n - 1)
    p = np.minimum(p - m, 1.0)
    c = s[m]
    return ((c[..., 3] * p + c[..., 4]) * p + c[..., 5]) * p + c[..., 6], (c[..., 0] * p + c[..., 1]) * p + c[..., 2]


def eam(funcfl, pos, cell, pbc):
    """``funcfl``: object with nrho, drho, nr, dr, cutoff, frho, zr, rhor.  Returns (E, e_atom [N], forces [N, 3])."""
    pos = np.asarray(pos, float).reshape(-1, 3)
    cell = np.asarray(cell, float).reshape(3, 3)
    n = len(pos)
    F = build_spline(funcfl.frho, funcfl.drho)
    R = build_spline(funcfl.rhor, funcfl.dr)
    Z2 = build_spline(27.2 * 0.529 * np.asarray(funcfl.zr, float) ** 2, funcfl.dr)
    # periodic images that can reach the cutoff
    vol = abs(np.linalg.det(cell))
    reps = []
    for k in range(3):
        if pbc[k]:
            cr = np.cross(cell[(k + 1) % 3], cell[(k + 2) % 3])
            reps.append(range(-int(np.ceil(funcfl.cutoff * np.linalg.norm(cr) / vol)), int(np.ceil(funcfl.cutoff * np.linalg.norm(cr) / vol)) + 1))
        else:
            reps.append(range(0, 1))
    ii, jj, rr = [], [], []
    for S in itertools.product(*reps):
        shift = np.dot(S, cell)
        d = pos[None, :, :] + shift - pos[:, None, :]           # d[i, j] = x_j + S - x_i
        dist = np.sqrt((d ** 2).sum(axis=2))
        mask = dist < funcfl.cutoff
        if S == (0, 0, 0):
            mask &= ~np.eye(n, dtype=bool)
        i, j = np.nonzero(mask)
        ii.append(i); jj.append(j); rr.append(d[i, j])
    ii, jj, rr = np.concatenate(ii), np.concatenate(jj), np.concatenate(rr)
    dist = np.sqrt((rr ** 2).sum(axis=1))
    rho_e, drho_e = spline_eval(R, dist, funcfl.dr, funcfl.nr)
    z_e, dz_e = spline_eval(Z2, dist, funcfl.dr, funcfl.nr)
    rho = np.zeros(n)
    np.add.at(rho, ii, rho_e)
    Fi, fp = spline_eval(F, rho, funcfl.drho, funcfl.nrho, clamp_lo=True)
    rhomax = (funcfl.nrho - 1) * funcfl.drho
    Fi = Fi + np.where(rho > rhomax, fp * (rho - rhomax), 0.0)
    phi = z_e / dist
    phip = dz_e / dist - phi / dist
    e_atom = Fi.copy()
    np.add.at(e_atom, ii, 0.5 * phi)
    psip = (fp[ii] + fp[jj]) * drho_e + phip
    forces = np.zeros((n, 3))
    np.add.at(forces, ii, (psip / dist)[:, None] * rr)
    return float(e_atom.sum()), e_atom, forces
