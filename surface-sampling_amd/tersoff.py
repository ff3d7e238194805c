"""LAMMPS ``pair_style tersoff`` potential-file parsing (host side, data format only).

The reference hands ``mcmc/potentials/GaN.tersoff`` to LAMMPS through
``pair_coeff * * <file> Ga N`` (``mcmc/calculators/calculators.py:559-568``, template
``tutorials/data/GaN_0001/GaN_0001_lammps_energy_template.txt``).  Entries are
``e1 e2 e3  m gamma lambda3 c d costheta0 n beta lambda2 B R D lambda1 A`` and may span lines;
e1 = centre atom i, e2 = bonded atom j, e3 = third atom k.
"""

from __future__ import annotations

import numpy as np

N_TERSOFF_FIELDS = 14
FIELD_NAMES = ("m", "gamma", "lambda3", "c", "d", "costheta0", "n", "beta", "lambda2", "B", "R", "D",
               "lambda1", "A")


def parse_tersoff(text: str, species: list[str]) -> np.ndarray:
    """Return params[nt, nt, nt, 14] (float64) for the given species order (LAMMPS type order)."""
    tokens: list[str] = []
    for raw in text.splitlines():
        line = raw.split("#", 1)[0].strip()
        if line:
            tokens += line.split()
    per = 3 + N_TERSOFF_FIELDS
    if len(tokens) % per:
        raise ValueError("tersoff file: token count is not a multiple of 17")
    nt = len(species)
    idx = {s: t for t, s in enumerate(species)}
    params = np.full((nt, nt, nt, N_TERSOFF_FIELDS), np.nan)
    for o in range(0, len(tokens), per):
        e1, e2, e3 = tokens[o:o + 3]
        if e1 in idx and e2 in idx and e3 in idx:
            vals = [float(x) for x in tokens[o + 3:o + per]]
            if vals[0] not in (1.0, 3.0):
                raise ValueError("tersoff: m must be 1 or 3")
            params[idx[e1], idx[e2], idx[e3]] = vals
    if np.isnan(params).any():
        raise ValueError("tersoff file lacks entries for some species triplets")
    return params


def max_cutoff(params: np.ndarray) -> float:
    return float((params[..., 10] + params[..., 11]).max())
