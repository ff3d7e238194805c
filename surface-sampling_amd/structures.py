"""Plain-array slab structures: the minimal stand-in for ``ase.Atoms`` on the hot path.

The calculators accept anything exposing ``get_positions()``, ``get_atomic_numbers()``,
``get_cell()`` and ``get_pbc()`` (``ase.Atoms`` does; ASE itself is optional).  This
module also reads the reference's structure files without ase/catkit
(SURVEY.md Appendix B): ASE-written P1 CIFs (``tests/data/SrTiO3_001/*.cif``) and the
pickled ``catkit.gratoms.Gratoms`` slabs (``tutorials/data/*/*.pkl``), and builds the
synthetic benchmark inputs of SURVEY.md §8(d).
"""

from __future__ import annotations

import io
import pickle
from dataclasses import dataclass, field

import numpy as np

SYMBOLS = (
    "X H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn "
    "Ga Ge As Se Br Kr Rb Sr Y Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr "
    "Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi Po At Rn Fr Ra "
    "Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm"
).split()
ATOMIC_NUMBERS = {s: z for z, s in enumerate(SYMBOLS)}


@dataclass
class Structure:
    """Positions [N,3] (Å, float64), atomic numbers [N], cell [3,3] (rows = lattice vectors), pbc [3]."""

    numbers: np.ndarray
    positions: np.ndarray
    cell: np.ndarray
    pbc: np.ndarray = field(default_factory=lambda: np.array([True, True, True]))
    constraints_fixed: np.ndarray | None = None  # indices held by FixAtoms, if any
    info: dict = field(default_factory=dict)

    def __post_init__(self):
        self.numbers = np.ascontiguousarray(self.numbers, dtype=np.int32)
        self.positions = np.ascontiguousarray(self.positions, dtype=np.float64).reshape(-1, 3)
        self.cell = np.ascontiguousarray(self.cell, dtype=np.float64).reshape(3, 3)
        self.pbc = np.ascontiguousarray(self.pbc, dtype=bool).reshape(3)
        if len(self.numbers) != len(self.positions):
            raise ValueError("numbers and positions differ in length")
        self.results: dict = {}
        self.calc = None

    # --- the subset of the ase.Atoms surface the calculators use -------------------------
    def __len__(self):
        return len(self.numbers)

    def get_positions(self):
        return self.positions.copy()

    def set_positions(self, pos):
        self.positions = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)

    def get_atomic_numbers(self):
        return self.numbers.copy()

    def get_chemical_symbols(self):
        return [SYMBOLS[z] for z in self.numbers]

    def get_cell(self):
        return self.cell.copy()

    def get_pbc(self):
        return self.pbc.copy()

    def get_potential_energy(self):
        return self.calc.get_potential_energy(self)

    def get_forces(self):
        return self.calc.get_forces(self)

    def copy(self):
        s = Structure(self.numbers.copy(), self.positions.copy(), self.cell.copy(), self.pbc.copy(),
                      None if self.constraints_fixed is None else self.constraints_fixed.copy(),
                      dict(self.info))
        return s

    def formula_counts(self) -> dict:
        out: dict[str, int] = {}
        for z in self.numbers:
            out[SYMBOLS[z]] = out.get(SYMBOLS[z], 0) + 1
        return out

    def repeat(self, reps) -> "Structure":
        """Tile the cell (like ``ase.Atoms.repeat``): image-major ordering, i0 slowest."""
        reps = tuple(int(r) for r in reps)
        pos, num = [], []
        for i0 in range(reps[0]):
            for i1 in range(reps[1]):
                for i2 in range(reps[2]):
                    shift = i0 * self.cell[0] + i1 * self.cell[1] + i2 * self.cell[2]
                    pos.append(self.positions + shift)
                    num.append(self.numbers)
        cell = self.cell * np.array(reps)[:, None]
        return Structure(np.concatenate(num), np.concatenate(pos), cell, self.pbc.copy())


def as_arrays(atoms):
    """(numbers int32 [N], positions f64 [N,3], cell f64 [3,3], pbc uint8 [3]) from Atoms-like."""
    numbers = np.ascontiguousarray(atoms.get_atomic_numbers(), dtype=np.int32)
    positions = np.ascontiguousarray(atoms.get_positions(), dtype=np.float64).reshape(-1, 3)
    cell = np.ascontiguousarray(np.asarray(atoms.get_cell()), dtype=np.float64).reshape(3, 3)
    pbc = np.ascontiguousarray(np.asarray(atoms.get_pbc()), dtype=np.uint8).reshape(3)
    return numbers, positions, cell, pbc


# --------------------------------------------------------------------------------------
# CIF (ASE-written, P1, orthorhombic or general angles)
# --------------------------------------------------------------------------------------
def read_cif(path: str) -> Structure:
    a = b = c = None
    alpha = beta = gamma = 90.0
    cols: list[str] = []
    rows: list[list[str]] = []
    in_loop = False
    in_atom_loop = False
    with open(path) as fh:
        for raw in fh:
            line = raw.strip()
            if not line or line.startswith("#"):
                if in_atom_loop and rows:
                    in_atom_loop = False
                continue
            if line.startswith("_cell_length_a"):
                a = float(line.split()[1])
            elif line.startswith("_cell_length_b"):
                b = float(line.split()[1])
            elif line.startswith("_cell_length_c"):
                c = float(line.split()[1])
            elif line.startswith("_cell_angle_alpha"):
                alpha = float(line.split()[1])
            elif line.startswith("_cell_angle_beta"):
                beta = float(line.split()[1])
            elif line.startswith("_cell_angle_gamma"):
                gamma = float(line.split()[1])
            elif line == "loop_":
                in_loop, in_atom_loop, cols = True, False, []
            elif in_loop and line.startswith("_"):
                cols.append(line.split()[0])
                in_atom_loop = any(cn.startswith("_atom_site_") for cn in cols)
            elif in_loop and in_atom_loop and "_atom_site_fract_x" in cols:
                parts = line.split()
                if len(parts) >= len(cols):
                    rows.append(parts)
            else:
                in_loop = False
    if a is None or not rows:
        raise ValueError(f"{path}: no cell or atom loop found")
    ca, cb, cg = (np.cos(np.deg2rad(x)) for x in (alpha, beta, gamma))
    sg = np.sin(np.deg2rad(gamma))
    if alpha == beta == gamma == 90.0:
        cell = np.diag([a, b, c])
    else:
        cx = c * cb
        cy = c * (ca - cb * cg) / sg
        cz = np.sqrt(max(c * c - cx * cx - cy * cy, 0.0))
        cell = np.array([[a, 0, 0], [b * cg, b * sg, 0], [cx, cy, cz]])
    isym = cols.index("_atom_site_type_symbol")
    ix, iy, iz = (cols.index(f"_atom_site_fract_{k}") for k in "xyz")
    numbers = [ATOMIC_NUMBERS[r[isym]] for r in rows]
    frac = np.array([[float(r[ix]), float(r[iy]), float(r[iz])] for r in rows])
    return Structure(np.array(numbers), frac @ cell, cell, np.array([True, True, True]))


# --------------------------------------------------------------------------------------
# catkit Gratoms / ase Atoms pickles, decoded with inert stubs (never executed)
# --------------------------------------------------------------------------------------
class _PStub:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self._state = state


class _SlabUnpickler(pickle.Unpickler):
    _NUMPY_OK = {
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy", "ndarray"), ("numpy", "dtype"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    }

    def find_class(self, module, name):
        if (module, name) in self._NUMPY_OK:
            import numpy._core.multiarray as ma  # numpy 2 home of _reconstruct / scalar

            if name in ("_reconstruct", "scalar"):
                return getattr(ma, name)
            return getattr(np, name)
        if module == "collections" and name == "OrderedDict":
            from collections import OrderedDict

            return OrderedDict
        if module in ("builtins", "__builtin__") and name in ("dict", "list", "set", "tuple", "frozenset"):
            return {"dict": dict, "list": list, "set": set, "tuple": tuple, "frozenset": frozenset}[name]
        if module.split(".")[0] in ("ase", "catkit", "networkx", "nff", "torch", "pymatgen", "mcmc"):
            return type(name, (_PStub,), {"__module__": module})
        raise pickle.UnpicklingError(f"global {module}.{name} is not allowed in a slab pickle")


def read_slab_pickle(path: str) -> Structure:
    with open(path, "rb") as fh:
        obj = _SlabUnpickler(io.BytesIO(fh.read())).load()
    d = vars(obj)
    arrays = d["arrays"]
    cellobj = d.get("_cellobj", d.get("_cell"))
    cell = np.asarray(getattr(cellobj, "array", cellobj), dtype=np.float64)
    pbc = np.asarray(d.get("_pbc", [True, True, True]), dtype=bool)
    fixed = None
    for con in d.get("_constraints", []) or []:
        idx = getattr(con, "index", None)
        if idx is not None:
            fixed = np.asarray(idx, dtype=np.int64)
    return Structure(np.asarray(arrays["numbers"]), np.asarray(arrays["positions"]), cell, pbc, fixed)


# --------------------------------------------------------------------------------------
# Synthetic benchmark inputs; deterministic in (base, chain index)
# --------------------------------------------------------------------------------------
def synth_chain(base: Structure, chain: int, species=(38, 22, 8), grid=(8, 8),
                jitter_xy=0.3, min_dist=1.5, sigma=0.05) -> Structure:
    """Chain ``c`` of the throughput workload: ``base`` plus ``8 + (c mod 25)`` adsorbates.

    Adsorbates are drawn with ``default_rng(1234 + c)`` uniformly from ``species`` and placed at
    z_top + 1.5 Å on a jittered in-plane grid, rejecting any pair closer than ``min_dist``
    (mirrors ``planar_distance`` / ``filter_distances``, reference ``mcmc/utils/misc.py:118-135``);
    then every atom is displaced by N(0, sigma) to emulate a mid-relaxation state.
    """
    rng = np.random.default_rng(1234 + int(chain))
    k = 8 + (int(chain) % 25)
    ztop = base.positions[:, 2].max()
    a, b = base.cell[0], base.cell[1]
    sites = [(i, j) for i in range(grid[0]) for j in range(grid[1])]
    order = rng.permutation(len(sites))
    ads_pos: list[np.ndarray] = []
    ads_num: list[int] = []
    for s in order:
        if len(ads_pos) == k:
            break
        i, j = sites[s]
        p = (i + 0.5) / grid[0] * a + (j + 0.5) / grid[1] * b
        p = p + np.array([rng.uniform(-jitter_xy, jitter_xy), rng.uniform(-jitter_xy, jitter_xy), 0.0])
        p[2] = ztop + 1.5
        ok = True
        for q in ads_pos:
            dv = p - q
            # in-plane minimum image
            f = np.linalg.solve(base.cell.T, dv)
            f[:2] -= np.round(f[:2])
            if np.linalg.norm(f @ base.cell) < min_dist:
                ok = False
                break
        if ok:
            ads_pos.append(p)
            ads_num.append(int(species[rng.integers(len(species))]))
    numbers = np.concatenate([base.numbers, np.array(ads_num, dtype=np.int32)])
    positions = np.concatenate([base.positions, np.array(ads_pos).reshape(-1, 3)])
    positions = positions + rng.normal(0.0, sigma, size=positions.shape)
    return Structure(numbers, positions, base.cell.copy(), base.pbc.copy())


# ---- LAMMPS data files (atom_style atomic) ----------------------------------------------------------------------------------
def _lammps_prism(cell):
    """LAMMPS' restricted triclinic box of a general cell: a along x, b in the xy plane (ASE ``Prism``): returns
    ``(lx, ly, lz, xy, xz, yz)`` and the matrix that rotates Cartesian vectors into that frame."""
    cell = np.asarray(cell, float).reshape(3, 3)
    a, b, c = cell
    lx = np.linalg.norm(a)
    ah = a / lx
    xy = float(np.dot(b, ah))
    ly = float(np.sqrt(max(np.dot(b, b) - xy * xy, 0.0)))
    xz = float(np.dot(c, ah))
    yz = float((np.dot(b, c) - xy * xz) / ly)
    lz = float(np.sqrt(max(np.dot(c, c) - xz * xz - yz * yz, 0.0)))
    new = np.array([[lx, 0.0, 0.0], [xy, ly, 0.0], [xz, yz, lz]])
    rot = np.linalg.solve(cell, new)          # x_lammps = x_cart @ rot   (fractional coordinates are preserved)
    return (float(lx), ly, lz, xy, xz, yz), rot


def write_lammps_data(path, atoms, specorder=None, name="lammps.data") -> list:
    """Write ``atoms`` as a LAMMPS data file, ``atom_style atomic`` -- what the reference's LAMMPS calculators produce
    with ``slab.write(lammps_data_file, format="lammps-data", atom_style="atomic")`` before every LAMMPS call (reference
    ``mcmc/calculators/calculators.py:548``; ``mcmc/calculators/lammpsrun.py:356-366``).  Atom types are numbered by
    ``specorder`` (symbols; default: the species present, sorted alphabetically like ASE).  Returns the species order.
    ``path`` may be a file name or an open text file."""
    Z, pos, cell, _ = as_arrays(atoms)
    symbols = [SYMBOLS[int(z)] for z in Z]
    species = list(specorder) if specorder is not None else sorted(set(symbols))
    missing = set(symbols) - set(species)
    if missing:
        raise ValueError(f"specorder lacks {sorted(missing)}")
    (lx, ly, lz, xy, xz, yz), rot = _lammps_prism(cell)
    x = np.asarray(pos, float) @ rot
    out = [f"{name} (written by surface_sampling_amd, ASE lammps-data layout)", "", f"{len(Z)} atoms",
           f"{len(species)} atom types", "", f"0.0 {lx:23.17g}  xlo xhi", f"0.0 {ly:23.17g}  ylo yhi",
           f"0.0 {lz:23.17g}  zlo zhi"]
    if max(abs(xy), abs(xz), abs(yz)) > 1e-12:
        out.append(f"{xy:23.17g} {xz:23.17g} {yz:23.17g}  xy xz yz")
    out += ["", "Atoms # atomic", ""]
    for i, (s, p) in enumerate(zip(symbols, x)):
        out.append(f"{i + 1:>6} {species.index(s) + 1:>3} {p[0]:23.17g} {p[1]:23.17g} {p[2]:23.17g}")
    text = "\n".join(out) + "\n"
    if hasattr(path, "write"):
        path.write(text)
    else:
        with open(path, "w") as fh:
            fh.write(text)
    return species


def read_lammps_data(path, species, pbc=(True, True, True)) -> Structure:
    """Read an ``atom_style atomic`` data file (the reference reads LAMMPS' ``write_data`` output back after a
    minimisation, ``mcmc/calculators/calculators.py:586``).  ``species``: symbol of type 1, 2, ..."""
    text = path.read() if hasattr(path, "read") else open(path).read()
    lines = [l.split("#")[0].strip() for l in text.splitlines()]
    n = lx = ly = lz = None
    xy = xz = yz = 0.0
    start = None
    for k, l in enumerate(lines):
        t = l.split()
        if len(t) == 2 and t[1] == "atoms":
            n = int(t[0])
        elif len(t) == 4 and t[2:] == ["xlo", "xhi"]:
            x0, lx = float(t[0]), float(t[1]) - float(t[0])
        elif len(t) == 4 and t[2:] == ["ylo", "yhi"]:
            y0, ly = float(t[0]), float(t[1]) - float(t[0])
        elif len(t) == 4 and t[2:] == ["zlo", "zhi"]:
            z0, lz = float(t[0]), float(t[1]) - float(t[0])
        elif len(t) == 6 and t[3:] == ["xy", "xz", "yz"]:
            xy, xz, yz = (float(v) for v in t[:3])
        elif t[:1] == ["Atoms"]:
            start = k + 1
            break
    if None in (n, lx, ly, lz, start):
        raise ValueError("not a LAMMPS data file with an Atoms section")
    rows = [l.split() for l in lines[start:] if l][:n]
    rows.sort(key=lambda r: int(r[0]))
    Z = np.array([ATOMIC_NUMBERS[species[int(r[1]) - 1]] for r in rows], np.int32)
    pos = np.array([[float(v) for v in r[2:5]] for r in rows]) - np.array([x0, y0, z0])
    cell = np.array([[lx, 0, 0], [xy, ly, 0], [xz, yz, lz]], float)
    return Structure(Z, pos, cell, np.array(pbc, bool))
