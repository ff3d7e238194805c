"""LAMMPS ``pair_style eam`` potential files in *funcfl* format (host side, data format only).

The reference hands ``mcmc/potentials/Cu_u3.eam`` to LAMMPS through ``LAMMPSRunSurfCalc.set(pair_style="eam",
pair_coeff=["* * Cu_u3.eam"])`` (``tests/test_Cu.py:41,65-70``, ``tutorials/example.ipynb`` cell 3).  Layout of a funcfl
file: line 1 comment; line 2 ``Z mass lattice-constant lattice-type``; line 3 ``Nrho drho Nr dr cutoff``; then ``Nrho``
values of the embedding energy F(rho) [eV], ``Nr`` values of the effective charge Z(r) [sqrt(Hartree Bohr)] and ``Nr``
values of the electron density rho(r), free format.
"""

from __future__ import annotations

import dataclasses

import numpy as np


@dataclasses.dataclass
class Funcfl:
    atomic_number: int
    mass: float
    lattice_constant: float
    lattice: str
    nrho: int
    drho: float
    nr: int
    dr: float
    cutoff: float
    frho: np.ndarray
    zr: np.ndarray
    rhor: np.ndarray
    comment: str = ""


def parse_funcfl(text: str) -> Funcfl:
    lines = text.splitlines()
    if len(lines) < 4:
        raise ValueError("funcfl file: too short")
    head = lines[1].split()
    grid = lines[2].split()
    if len(head) < 3 or len(grid) < 5:
        raise ValueError("funcfl file: malformed header")
    nrho, drho, nr, dr, cutoff = int(grid[0]), float(grid[1]), int(grid[2]), float(grid[3]), float(grid[4])
    vals = np.array(" ".join(lines[3:]).split(), dtype=np.float64)
    if vals.size < nrho + 2 * nr:
        raise ValueError(f"funcfl file: {vals.size} table values, need {nrho + 2 * nr}")
    if nrho < 5 or nr < 5 or not (drho > 0 and dr > 0 and cutoff > 0):
        raise ValueError("funcfl file: bad grid")
    return Funcfl(int(float(head[0])), float(head[1]), float(head[2]), head[3] if len(head) > 3 else "",
                  nrho, drho, nr, dr, cutoff, vals[:nrho].copy(), vals[nrho:nrho + nr].copy(),
                  vals[nrho + nr:nrho + 2 * nr].copy(), lines[0].strip())


def read_funcfl(path) -> Funcfl:
    with open(path) as fh:
        return parse_funcfl(fh.read())
