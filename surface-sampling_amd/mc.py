"""Batched semigrand-canonical MC over independent chains.

The reference runs ONE chain: ``MCMC.step_semigrand`` (``mcmc/mcmc.py:233-266``) builds a ``ChangeProposal``
(``mcmc/events/proposal.py:74-106``), applies ``change_site`` (``mcmc/slab.py:235-274``) between ``save_state("before")``
and ``save_state("after")`` (``mcmc/events/event.py:97-105``), relaxes the proposed slab from its UNRELAXED lattice
positions (``mcmc/system.py:359-378,451-470``) and accepts with the Metropolis rule (``mcmc/events/criterion.py:157-168``).
With thousands of chains that Python loop is the serial fraction, so here the same step is array arithmetic over B
chains: site occupations are ``[B, S]`` arrays, proposals / acceptances are drawn from a counter-based generator
(Philox4x32-10 keyed by the seed, counter = (step, global chain id, draw)), so a chain's trajectory does not depend on
how chains are batched or sharded over GPUs, and the B proposed slabs are relaxed in one lock-step device call
(``EnsembleNFFSurface.relax_batch``).  "before"/"after" states are simply the old and the new arrays.

Canonical sampling (``MCMC.step_canonical`` ``mcmc/mcmc.py:190-231``: ``SwitchProposal`` ``mcmc/events/proposal.py:154-197``
-> ``get_complementary_idx`` ``mcmc/slab.py:168-232`` with uniform weights, ``Exchange`` ``mcmc/events/event.py:138-156``) is
the same array arithmetic: two different "types" present on the lattice (a species or "None" for empty sites), one site of
each, adsorbates exchanged.  Site weights: uniform, Boltzmann in the per-atom energies (``compute_boltzmann_weights``,
``mcmc/slab.py:73-113``) and / or distance decay around the first site (``get_complementary_idx_distance_decay``
``mcmc/slab.py:116-165``, matrix ``compute_distance_weight_matrix`` ``mcmc/utils/misc.py:170-190``).  With
``reference_groupby=True`` the candidate sites of a species are those of the reference's ``get_adsorbate_indices``
(``mcmc/slab.py:36-57``): ``itertools.groupby`` over the filled sites in site order keyed by the symbol of the adsorbate's
first atom, a dict that keeps only the LAST consecutive run of every key -- reproduced here run for run; the default
(all sites of a species) is what the reference's docstring describes.

Adsorbates are single atoms or the reference's multi-atom ``ATOM_GROUPS`` ("HO", "H2O"; ``mcmc/slab.py:22-33``, added /
removed as in ``add_atom_group`` / ``remove_atom_group`` ``:313-346,398-436``).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import structures

# ---------------------------------------------------------------------------------------------------------------------
# Counter-based random numbers: Philox4x32-10 (Salmon et al., SC'11), vectorised over any leading shape.
# ---------------------------------------------------------------------------------------------------------------------
_PHILOX_M0, _PHILOX_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PHILOX_W0, _PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32(counter, key, rounds: int = 10) -> np.ndarray:
    """``counter [..., 4]`` and ``key [..., 2]`` (uint32) -> ``[..., 4]`` uint32."""
    c = np.asarray(counter, dtype=np.uint64) & _MASK32
    k = np.asarray(key, dtype=np.uint64) & _MASK32
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = k[..., 0].copy(), k[..., 1].copy()
    for r in range(rounds):
        p0, p1 = _PHILOX_M0 * c0, _PHILOX_M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & _MASK32, p1 >> np.uint64(32), p1 & _MASK32
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        if r + 1 < rounds:
            k0 = (k0 + np.uint64(_PHILOX_W0)) & _MASK32
            k1 = (k1 + np.uint64(_PHILOX_W1)) & _MASK32
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def chain_uniforms(seed: int, chain_ids, step: int, draw: int = 0) -> np.ndarray:
    """Four uniforms in [0, 1) per chain for MC step ``step``: ``[B, 4]`` float64.  A function of
    (seed, global chain id, step, draw) only."""
    ids = np.asarray(chain_ids, dtype=np.uint64)
    ctr = np.empty(ids.shape + (4,), np.uint64)
    ctr[..., 0] = np.uint64(step & 0xFFFFFFFF)
    ctr[..., 1] = np.uint64((step >> 32) & 0xFFFFFFFF)
    ctr[..., 2] = ids & _MASK32
    ctr[..., 3] = ((ids >> np.uint64(32)) << np.uint64(8)) | np.uint64(draw & 0xFF)
    key = np.empty(ids.shape + (2,), np.uint64)
    key[..., 0] = np.uint64(seed & 0xFFFFFFFF)
    key[..., 1] = np.uint64((seed >> 32) & 0xFFFFFFFF)
    return philox4x32(ctr, key).astype(np.float64) * (1.0 / 4294967296.0)


# ---------------------------------------------------------------------------------------------------------------------
# One chain, reference index semantics (mirror of mcmc/slab.py change_site / add_atom / remove_atom for single atoms)
# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class SiteState:
    """Occupation bookkeeping of ONE surface the way ``SurfaceSystem`` keeps it (``mcmc/system.py:99-127``):
    ``numbers``/``positions`` of the real atoms, ``ads_group[i]`` = index of the adsorbate atom i belongs to (0 for
    slab atoms), ``occ[s]`` = atom index of the adsorbate on site s (0 = empty), ``ads_coords [S, 3]``."""

    numbers: np.ndarray
    positions: np.ndarray
    ads_group: np.ndarray
    occ: np.ndarray
    ads_coords: np.ndarray

    def copy(self) -> "SiteState":
        return SiteState(self.numbers.copy(), self.positions.copy(), self.ads_group.copy(), self.occ.copy(),
                         self.ads_coords)

    def __len__(self):
        return len(self.numbers)

    @property
    def symbols(self):
        return [structures.SYMBOLS[int(z)] for z in self.numbers]


# multi-atom adsorbates of the reference (mcmc/slab.py:22-33): symbols and positions relative to the site coordinate
ATOM_GROUPS = {
    "HO": (("O", "H"), ((0.0, 0.0, 0.0), (1.0, 0.0, 0.0))),
    "H2O": (("O", "H", "H"), ((0.0, 0.0, 0.0), (0.5, -np.sqrt(3.0) / 2.0, 0.0), (0.5, np.sqrt(3.0) / 2.0, 0.0))),
}


def adsorbate_atoms(name: str):
    """``(atomic numbers [k], offsets [k, 3])`` of an adsorbate: an element symbol or an ``ATOM_GROUPS`` name."""
    if name in ATOM_GROUPS:
        syms, offs = ATOM_GROUPS[name]
        return np.array([structures.ATOMIC_NUMBERS[x] for x in syms], np.int32), np.array(offs, float)
    return np.array([structures.ATOMIC_NUMBERS[name]], np.int32), np.zeros((1, 3))


def change_site(state: SiteState, site_idx: int, end_ads: str) -> SiteState:
    """``change_site`` of the reference (``mcmc/slab.py:235-274``): remove what sits on the site -- one atom or a whole group,
    i.e. every atom whose ``ads_group`` equals the site's ``occ`` entry; later atom indices and ``occ`` / ``ads_group``
    entries shift down by the number of atoms removed (``:347-395``) -- then append the new adsorbate (atom or
    ``ATOM_GROUPS`` member, ``:291-346``) at the site coordinate with ``occ[site] = ads_group =`` the index of its first atom."""
    if site_idx >= len(state.occ) or site_idx < 0:
        raise IndexError("site index out of range")
    s = state.copy()
    idx = int(s.occ[site_idx])
    if idx != 0:
        assert np.count_nonzero(s.occ == idx) == 1, "adsorbate index must belong to exactly one site"
        members = np.flatnonzero(s.ads_group == idx)
        k = len(members)
        assert k >= 1 and np.array_equal(members, np.arange(idx, idx + k)), "an adsorbate's atoms are contiguous"
        s.numbers = np.delete(s.numbers, members)
        s.positions = np.delete(s.positions, members, axis=0)
        s.ads_group = np.delete(s.ads_group, members)
        s.occ = np.where(s.occ >= idx, s.occ - k, s.occ)
        s.ads_group = np.where(s.ads_group >= idx, s.ads_group - k, s.ads_group)
        s.occ = np.where(s.occ < 0, 0, s.occ)
        s.ads_group = np.where(s.ads_group < 0, 0, s.ads_group)
        s.occ[site_idx] = 0
    if end_ads != "None":
        new_idx = len(s.numbers)
        z, offs = adsorbate_atoms(end_ads)
        s.numbers = np.concatenate([s.numbers, z]).astype(state.numbers.dtype)
        s.positions = np.vstack([s.positions, np.asarray(s.ads_coords[site_idx], float)[None] + offs])
        s.ads_group = np.concatenate([s.ads_group, np.full(len(z), new_idx, s.ads_group.dtype)])
        s.occ[site_idx] = new_idx
    return s


def compute_distance_weight_matrix(ads_coords, distance_decay_factor: float) -> np.ndarray:
    """Row-wise softmax of ``-distance / decay`` between adsorption sites (``mcmc/utils/misc.py:170-190``)."""
    x = np.asarray(ads_coords, float)
    d = np.sqrt(((x[:, None, :] - x[None, :, :]) ** 2).sum(axis=2))
    z = -d / float(distance_decay_factor)
    z -= z.max(axis=1, keepdims=True)
    w = np.exp(z)
    return w / w.sum(axis=1, keepdims=True)


def boltzmann_atom_weights(per_atom_energies, temperature: float) -> np.ndarray:
    """``softmax(per_atom_energies / T)`` over the atoms of one slab (``compute_boltzmann_weights``, ``mcmc/slab.py:104``)."""
    z = np.asarray(per_atom_energies, float) / float(temperature)
    z = z - z.max()
    w = np.exp(z)
    return w / w.sum()


def metropolis_accept(prev_energy, curr_energy, temperature: float, u) -> np.ndarray:
    """Metropolis rule of the reference (``mcmc/events/criterion.py:157-168``), vectorised: accept where
    ``u < exp(-(E_after - E_before) / kT)`` (overflow of the exponential = certain acceptance)."""
    diff = np.asarray(curr_energy, float) - np.asarray(prev_energy, float)
    with np.errstate(over="ignore", invalid="ignore"):
        prob = np.exp(-diff / float(temperature))
    prob = np.where(np.isnan(prob), 0.0, prob)
    return np.asarray(u, float) < prob


def complete_cell(cell) -> np.ndarray:
    """``ase.geometry.complete_cell``: zero-length cell vectors (the non-periodic axis of a slab given without vacuum) are replaced
    by unit vectors orthogonal to the others, so the cell can be inverted."""
    cell = np.array(cell, float).reshape(3, 3)
    missing = np.nonzero(~cell.any(axis=1))[0]
    if len(missing) == 3:
        cell.flat[::4] = 1.0
    elif len(missing) == 2:
        i = 3 - int(missing.sum())            # the one vector present
        _, _, vh = np.linalg.svd(cell[i][None, :])
        cell[missing] = vh[1:]
    elif len(missing) == 1:
        i = int(missing[0])
        n = np.cross(cell[i - 2], cell[i - 1])
        nn = np.linalg.norm(n)
        if nn == 0.0:
            raise ValueError("cell vectors are linearly dependent")
        cell[i] = n / nn
    return cell


def mic_distance_matrix(xa, xb, cell, pbc) -> np.ndarray:
    """Minimum-image distances ``[len(xa), len(xb)]`` (``ase.Atoms.get_all_distances(mic=True)``): fractional differences are
    wrapped into [-1/2, 1/2) along the periodic axes, then the shortest image within a search range is taken.  Degenerate
    (zero-length) vectors of non-periodic axes are completed like ASE does.  The image range per periodic axis is derived from
    the cell: every lattice vector whose length could beat the wrapped difference is covered (range ``ceil(|d|max / h_i)`` with
    ``h_i`` the distance between the lattice planes of axis i) -- ASE reaches the same minimum through a Minkowski reduction;
    1 for every slab cell of the reference, more only for strongly skewed cells."""
    cell = complete_cell(cell)
    pbc = np.asarray(pbc, bool).reshape(3)
    d = np.asarray(xb, float)[None, :, :] - np.asarray(xa, float)[:, None, :]
    frac = d @ np.linalg.inv(cell)
    frac = np.where(pbc[None, None, :], frac - np.round(frac), frac)
    d = frac @ cell
    ortho = np.allclose(cell - np.diag(np.diag(cell)), 0.0)
    if ortho or not pbc.any():
        return np.sqrt((d * d).sum(axis=2))
    # an image shifted by n_i along axis i is at least (|n_i| - 1/2) h_i away (h_i: spacing of that axis' lattice planes; the
    # wrapped difference lies within +- h_i / 2 of its plane): images with (|n_i| - 1/2) h_i > max |d| cannot win
    dmax = float(np.sqrt((d * d).sum(axis=2)).max()) if d.size else 0.0
    heights = abs(np.linalg.det(cell)) / np.array([np.linalg.norm(np.cross(cell[(i + 1) % 3], cell[(i + 2) % 3])) for i in range(3)])
    reach = [max(1, int(np.floor(dmax / heights[i] + 0.5))) if pbc[i] else 0 for i in range(3)]
    if max(reach) > 64:
        raise ValueError("cell too skewed for the minimum-image search: reduce it first")
    best = None
    for i in range(-reach[0], reach[0] + 1):
        for j in range(-reach[1], reach[1] + 1):
            for k in range(-reach[2], reach[2] + 1):
                dd = d + (i * cell[0] + j * cell[1] + k * cell[2])[None, None, :]
                r2 = (dd * dd).sum(axis=2)
                best = r2 if best is None else np.minimum(best, r2)
    return np.sqrt(best)


def filter_distances(slab, ads=("O",), cutoff_distance: float = 1.5) -> bool:
    """``filter_distances`` of the reference (``mcmc/utils/misc.py:118-135``): True when no two atoms whose symbol is in ``ads``
    -- slab atoms of those species included -- are closer than ``cutoff_distance`` under the minimum-image convention (pairs at
    exactly zero distance pass, as there).  ``slab``: anything with ``numbers`` / ``positions`` / ``cell`` / ``pbc``
    (``structures.Structure``) or the ASE accessors."""
    num, pos, cell, pbc = structures.as_arrays(slab)
    ads = [ads] if isinstance(ads, str) else list(ads)
    sel = np.isin(num, [structures.ATOMIC_NUMBERS[a] for a in ads])
    if sel.sum() < 2:
        return True
    d = mic_distance_matrix(pos[sel], pos[sel], cell, pbc)
    iu = np.triu_indices(int(sel.sum()), k=1)
    dd = d[iu]
    return not bool(np.any((dd > 0) & (dd <= cutoff_distance)))


def geometric_schedule(start_temp: float, total_sweeps: int, alpha: float) -> list:
    """Simulated-annealing temperatures ``T_i = start_temp * alpha**i`` for ``total_sweeps`` sweeps."""
    return (float(start_temp) * np.power(float(alpha), np.arange(int(total_sweeps)))).tolist()


# One cooling / reheating cycle of the reference's "multiple anneal" mode as data: (kind, T_from, T_to, sweeps); ``None``
# stands for the temperature the cycle was entered with (``mcmc/utils/sampling.py:48-62`` lists the same four legs).
REFERENCE_ANNEAL_CYCLE = (("ramp", None, 0.10, 100), ("ramp", 0.10, 0.08, 200), ("hold", 0.08, 0.08, 200),
                          ("ramp", 0.08, None, 10))


def cyclic_schedule(start_temp: float, total_sweeps: int, cycle=REFERENCE_ANNEAL_CYCLE) -> list:
    """``start_temp`` followed by repetitions of ``cycle`` (linear ramps / plateaus), cut to ``total_sweeps`` entries."""
    legs = []
    for kind, t0, t1, n in cycle:
        a = start_temp if t0 is None else t0
        b = start_temp if t1 is None else t1
        legs.append(np.full(n, a) if kind == "hold" else np.linspace(a, b, n))
    one = np.concatenate(legs)
    reps = max(1, -(-(int(total_sweeps) - 1) // len(one)))
    return np.concatenate([[float(start_temp)], np.tile(one, reps)])[:int(total_sweeps)].tolist()


def create_anneal_schedule(start_temp: float = 1.0, total_sweeps: int = 1000, alpha: float = 0.99,
                           multiple_anneal: bool = False) -> list:
    """Temperature per sweep with the reference's argument names (``mcmc/utils/sampling.py:10-67``, without its plot /
    csv side effects): geometric cooling, or the cyclic schedule above."""
    if multiple_anneal:
        return cyclic_schedule(start_temp, total_sweeps)
    return geometric_schedule(start_temp, total_sweeps, alpha)


# ---------------------------------------------------------------------------------------------------------------------
# Slabs of many chains as packed arrays (no per-chain Python objects on the MC hot path)
# ---------------------------------------------------------------------------------------------------------------------
class SlabBatch:
    """B slabs of one base cell as the ABI's packed arrays; ``batch[b]`` builds the ``Structure`` of chain b on demand."""

    def __init__(self, n_atoms, numbers, positions, cell, pbc):
        self.n_atoms = np.asarray(n_atoms, dtype=np.int64)
        self.start = np.concatenate([[0], np.cumsum(self.n_atoms)]).astype(np.int64)
        self.numbers, self.positions, self.cell, self.pbc = numbers, positions, cell, pbc

    def __len__(self):
        return len(self.n_atoms)

    def __getitem__(self, b):
        a0, a1 = int(self.start[b]), int(self.start[b + 1])
        return structures.Structure(self.numbers[a0:a1].copy(), self.positions[a0:a1].copy(), self.cell, self.pbc)


class SlabRefs:
    """A list of slabs whose entries are either ``Structure`` objects or ``(SlabBatch, index)`` references that turn into a
    ``Structure`` when they are looked at: the accept / restore bookkeeping of an MC step shuffles 256 references instead of
    building 256 structures nobody reads."""

    def __init__(self, items):
        self.items = list(items)

    @classmethod
    def of_batch(cls, batch: SlabBatch):
        return cls([(batch, b) for b in range(len(batch))])

    def __len__(self):
        return len(self.items)

    def __getitem__(self, b):
        if isinstance(b, slice):
            return [self[i] for i in range(*b.indices(len(self.items)))]
        it = self.items[b]
        if isinstance(it, tuple):
            it = self.items[b] = it[0][it[1]]
        return it

    def __setitem__(self, b, value):
        self.items[b] = value

    def __iter__(self):
        return (self[b] for b in range(len(self.items)))

    def raw(self, b):
        """The entry as stored (a reference stays a reference)."""
        return self.items[b]

    def consolidate(self, max_batches: int = 16):
        """A reference keeps its whole ``SlabBatch`` (the packed slabs of ALL chains of one MC step) alive; a chain that has not
        been accepted for a long time pins an old one.  When more than ``max_batches`` distinct batches are referenced, the
        referenced slabs are copied into ONE fresh batch (still no ``Structure`` objects), bounding the memory at about
        ``max_batches`` steps' worth of positions."""
        batches = {id(it[0]): it[0] for it in self.items if isinstance(it, tuple)}
        if len(batches) <= max_batches:
            return self
        ref_idx = [k for k, it in enumerate(self.items) if isinstance(it, tuple)]
        first = self.items[ref_idx[0]][0]
        n_atoms = np.array([int(self.items[k][0].n_atoms[self.items[k][1]]) for k in ref_idx], np.int64)
        numbers = np.concatenate([self.items[k][0].numbers[self.items[k][0].start[self.items[k][1]]:self.items[k][0].start[self.items[k][1] + 1]]
                                  for k in ref_idx])
        positions = np.concatenate([self.items[k][0].positions[self.items[k][0].start[self.items[k][1]]:self.items[k][0].start[self.items[k][1] + 1]]
                                    for k in ref_idx])
        merged = SlabBatch(n_atoms, numbers, positions, first.cell, first.pbc)
        for j, k in enumerate(ref_idx):
            self.items[k] = (merged, j)
        return self


def _raw_items(seq):
    """The entries as stored (references stay references), in a NEW list."""
    return list(seq.items) if isinstance(seq, SlabRefs) else list(seq)


# ---------------------------------------------------------------------------------------------------------------------
# B chains
# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class ChainState:
    """Occupations of B chains.  ``species[b, s]``: index into ``adsorbates`` or ``n_ads`` for an empty site (the
    position of "None" in the reference's choice list, ``proposal.py:36-38``).  ``order[b, s]``: adsorption sequence
    number of the adsorbate on the site (0 if empty) — adsorbate atoms follow the slab atoms in increasing ``order``,
    which reproduces the reference's atom ordering (append on adsorption, shift down on removal)."""

    species: np.ndarray
    order: np.ndarray
    counter: np.ndarray                     # [B] next sequence number
    energy: np.ndarray = field(default=None)  # [B] surface energy of the current state (None: not evaluated yet)

    def copy(self) -> "ChainState":
        return ChainState(self.species.copy(), self.order.copy(), self.counter.copy(),
                          None if self.energy is None else self.energy.copy())


class ChainEnsemble:
    """B independent semigrand chains on one base slab and one site lattice.

    Args:
        base: the pristine slab (``structures.Structure``; slab atoms only).
        ads_coords: ``[S, 3]`` adsorption-site coordinates (``SurfaceSystem.ads_coords``).
        adsorbates: adsorbate names, e.g. ``("Sr", "O")`` (``ChangeProposal`` default, ``proposal.py:55``); element symbols or
            ``ATOM_GROUPS`` names ("HO", "H2O").
        n_chains: chains owned by this process; ``first_chain`` = global id of the first one (sharding).
        calc: energy backend with ``relax_batch(atoms_list, fixed_indices, relax_steps, fmax)`` (relax=True) or
            ``calculate_batch(atoms_list)`` (relax=False), e.g. ``calculators.EnsembleNFFSurface``.
        surface_energy_fn: ``(energy, structure) -> float``; default: the calculator's chemical-potential / bulk
            reference arithmetic when it has ``chem_pots`` and ``offset_data``, else the plain energy.
        fixed_indices: atom indices held fixed during relaxation (``bulk_idx``, ``mcmc/system.py:288-294``).
    """

    def __init__(self, base, ads_coords, adsorbates, n_chains: int, calc, *, seed: int = 0, first_chain: int = 0,
                 relax: bool = True, relax_steps: int = 20, fmax: float = 0.01, fixed_indices=None,
                 surface_energy_fn=None, temperature: float = 1.0, optimizer: str = "BFGS",
                 reference_groupby: bool = False, require_per_atom_energies: bool = False,
                 require_distance_decay: bool = False, distance_decay_factor: float = 1.0,
                 exchange_by_group_key: bool = False, filter_distance: float = 0.0,
                 distance_adsorbate_types=("Sr", "Ti"), testing: bool = False, acceptance_energy: str = "f64"):
        # Which word of the device's ensemble energy the acceptance test compares: "f64" (default) = the mean as the device forms
        # it, before the narrowing to the float32 result word; "f32" = results["energy"], the float32 word the reference's
        # Metropolis test sees (mcmc/calculators/calculators.py:484) -- for runs that want the reference's accept / reject decisions
        # reproduced even where |dE| is of the order of the float32 spacing at keV energies (1e-4 .. 5e-4 eV).
        if acceptance_energy not in ("f64", "f32"):
            raise ValueError('acceptance_energy must be "f64" or "f32"')
        self.acceptance_energy = acceptance_energy
        self.base = base
        self.ads_coords = np.asarray(ads_coords, float).reshape(-1, 3)
        self.adsorbates = list(adsorbates)
        self.n_ads = len(self.adsorbates)
        self.ads_atoms = [adsorbate_atoms(a) for a in self.adsorbates]          # (numbers [k], offsets [k, 3]) per adsorbate
        self.ads_sizes = np.array([len(z) for z, _ in self.ads_atoms] + [0], np.int64)   # (+ the "None" code)
        kmax = max(1, int(self.ads_sizes.max()))
        self._ads_Z = np.zeros((self.n_ads + 1, kmax), np.int32)          # atoms of every adsorbate, padded (batch_arrays)
        self._ads_off = np.zeros((self.n_ads + 1, kmax, 3))
        for c, (z, offs) in enumerate(self.ads_atoms):
            self._ads_Z[c, :len(z)] = z
            self._ads_off[c, :len(z)] = offs
        self.fast_path = True     # packed-array evaluation when the calculator offers it (evaluate_packed); False: per-slab objects
        # switch-proposal options of the reference's SwitchProposal (mcmc/events/proposal.py:112-160)
        self.reference_groupby = bool(reference_groupby)
        self.require_per_atom_energies = bool(require_per_atom_energies)
        self.require_distance_decay = bool(require_distance_decay)
        self.distance_weight_matrix = (compute_distance_weight_matrix(self.ads_coords, distance_decay_factor)
                                       if require_distance_decay else None)
        # key of an adsorbate in the reference's grouping = symbol of its FIRST atom ("HO" and "O" share the key "O")
        first = [int(z[0]) for z, _ in self.ads_atoms]
        self.group_key = np.array([first.index(f) for f in first] + [self.n_ads], np.int64)
        # Canonical exchange: WHAT is exchanged.  Default (False): the adsorbates that actually sit on the two sites -- the
        # composition is conserved whatever the grouping.  True reproduces the reference to the letter: its Exchange event passes
        # the GROUP KEYS type1 / type2 (the symbol of an adsorbate's first atom, get_complementary_idx mcmc/slab.py:168-232 ->
        # mcmc/events/event.py:138-151) to change_site, so with adsorbates that share a first-atom symbol ("HO" and "O") an
        # exchanged "HO" arrives as the plain atom "O" and the composition changes.  The key must then name an adsorbate of the
        # list (the state arrays cannot hold a species outside it).  A deliberate divergence of the default, DESIGN.md section 7.
        self.exchange_by_group_key = bool(exchange_by_group_key)
        self.key_adsorbate = np.arange(self.n_ads + 1, dtype=np.int64)
        if self.exchange_by_group_key:
            for c, (z, _) in enumerate(self.ads_atoms):
                sym = structures.SYMBOLS[int(z[0])]
                if sym not in self.adsorbates:
                    raise ValueError(f'exchange_by_group_key: the reference would adsorb the plain atom "{sym}" (group key of '
                                     f'"{self.adsorbates[c]}"), which is not in the adsorbate list {self.adsorbates}')
                self.key_adsorbate[c] = self.adsorbates.index(sym)
        # Acceptance criterion, chosen like MCMC.step_semigrand / step_canonical choose it (mcmc/mcmc.py:218-227,253-262):
        # filter_distance > 0 -> DistanceCriterion (mcmc/events/criterion.py:74-114: accept iff filter_distances() holds for the
        # proposed UNRELAXED slab; adsorbate_types keeps the criterion's default ("Sr", "Ti") -- the reference calls it without
        # arguments -- and no energy enters), elif testing -> TestingCriterion (always accept), else Metropolis.
        self.filter_distance = float(filter_distance)
        self.distance_adsorbate_types = tuple(distance_adsorbate_types)
        self.testing = bool(testing)
        self.criterion = "distance" if self.filter_distance > 0 else "testing" if self.testing else "metropolis"
        self.calc = calc
        self.seed, self.first_chain = int(seed), int(first_chain)
        self.relax, self.relax_steps, self.fmax = bool(relax), int(relax_steps), float(fmax)
        self.optimizer = optimizer   # reference SrTiO3 configuration: BFGS (scripts/configs/sample_config_painn.json:26)
        self.fixed_indices = None if fixed_indices is None else np.asarray(fixed_indices, np.int64)
        self.temp = float(temperature)
        self._default_se = surface_energy_fn is None
        self.surface_energy_fn = surface_energy_fn or self._default_surface_energy
        B, S = int(n_chains), len(self.ads_coords)
        self.chain_ids = np.arange(self.first_chain, self.first_chain + B, dtype=np.int64)
        self.state = ChainState(np.full((B, S), self.n_ads, np.int16), np.zeros((B, S), np.int64), np.ones(B, np.int64))
        self.step_count = 0
        self.relaxed = SlabRefs([None] * B)   # relaxed Structure of the current state of every chain (built on demand)
        self.oob = np.zeros(B, bool)        # out-of-bounds flag of the last evaluation of every chain (reference: energy_oob)
        self.per_atom_energies = [None] * B  # of the current state (filled when the backend returns them)
        self.n_evaluations = 0
        self.energy_stale = np.zeros(B, bool)   # criteria without energies: state energies are brought up to date per sweep

    # ---- proposal: vectorised ChangeProposal.get_action ------------------------------------------------------------
    def propose(self, step: int, state: ChainState | None = None, site_idx=None):
        """Per chain: a uniformly random site, and a uniformly random entry of ``adsorbates + ["None"]`` minus what
        is on the site now (``proposal.py:82-106``).  Returns ``(site_idx [B], end_code [B], start_code [B], u_acc [B])``
        where ``u_acc`` is the uniform reserved for the acceptance test of this step."""
        st = state or self.state
        u = chain_uniforms(self.seed, self.chain_ids, step)
        S = st.species.shape[1]
        site = np.minimum((u[:, 0] * S).astype(np.int64), S - 1)
        if site_idx is not None:       # ChangeProposal(site_idx=...): the site is given, the new adsorbate is still drawn
            site = np.broadcast_to(np.asarray(site_idx, dtype=np.int64), site.shape).copy()
            if site.min() < 0 or site.max() >= S:
                raise IndexError("site index out of range")
        start = st.species[np.arange(len(site)), site].astype(np.int64)
        k = np.minimum((u[:, 1] * self.n_ads).astype(np.int64), self.n_ads - 1)   # n_ads + 1 choices minus the current one
        end = np.where(k < start, k, k + 1)
        return site, end, start, u[:, 2]

    # ---- proposal: vectorised SwitchProposal.get_action (uniform weights) -------------------------------------------
    def switch_candidates(self, state: ChainState | None = None) -> np.ndarray:
        """``[B, n_types, S]`` bool: candidate sites of every type code (the last code = empty sites).  Default: all sites
        that carry the type.  ``reference_groupby``: the reference's ``get_adsorbate_indices`` (``mcmc/slab.py:36-57``) --
        filled sites in site order, grouped by CONSECUTIVE equal key (symbol of the adsorbate's first atom) with
        ``itertools.groupby``, collected in a dict, so a later run of a key replaces the earlier ones: the candidates of a
        key are the sites of its last run (types that share a first-atom symbol share one entry)."""
        st = state or self.state
        B, S = st.species.shape
        filled = st.species != self.n_ads
        cand = np.zeros((B, self.n_ads + 1, S), bool)
        cand[:, self.n_ads] = ~filled
        if not self.reference_groupby:
            for c in range(self.n_ads):
                cand[:, c] = st.species == c
            return cand
        key = self.group_key[st.species]                              # [B, S]; n_ads for empty sites
        # key of the previous FILLED site (empty sites do not interrupt a run: groupby only sees the filled ones)
        idx = np.where(filled, np.arange(S)[None, :], -1)
        last_filled = np.maximum.accumulate(idx, axis=1)
        prev = np.concatenate([np.full((B, 1), -1), last_filled[:, :-1]], axis=1)
        prev_key = np.where(prev >= 0, np.take_along_axis(key, np.maximum(prev, 0), axis=1), -1)
        run_start = filled & (key != prev_key)
        run_id = np.cumsum(run_start, axis=1)                         # run number of every filled site (1-based)
        for c in sorted(set(self.group_key[: self.n_ads].tolist())):
            m = filled & (key == c)
            last_run = np.where(m, run_id, 0).max(axis=1)
            cand[:, c] = m & (run_id == last_run[:, None])
        return cand

    def _site_weights(self, cand: np.ndarray, state: ChainState) -> np.ndarray:
        """Per-site selection weights ``[B, S]``: 1, or the Boltzmann weight of the adsorbate's first atom
        (``compute_boltzmann_weights``: softmax of per-atom energies / T over the slab's atoms; empty sites weigh 1)."""
        B, S = state.species.shape
        if not self.require_per_atom_energies:
            return np.ones((B, S))
        occ = self.occ(state)
        w = np.ones((B, S))
        for b in range(B):
            pae = self.per_atom_energies[b]
            if pae is None:
                raise ValueError("require_per_atom_energies is True, but no per_atom_energies were provided")
            bw = boltzmann_atom_weights(pae, self.temp)
            f = state.species[b] != self.n_ads
            w[b, f] = bw[occ[b, f]]
        return w

    def propose_switch(self, step: int, state: ChainState | None = None):
        """Per chain: an ordered pair of DIFFERENT types present on the lattice (``random.sample(types, 2)``,
        ``mcmc/slab.py:60-71``), then one site of each type (``mcmc/slab.py:200-222``): uniformly, or with Boltzmann weights
        (``require_per_atom_energies``), the second site additionally weighted by the distance-decay row of the first
        (``require_distance_decay``, ``mcmc/slab.py:116-165``).  Returns ``(site1 [B], site2 [B], type1 [B], type2 [B],
        valid [B], u_acc [B])``; ``valid`` is False for chains with fewer than two types on the lattice (the reference cannot
        propose there).  Candidate sites: :meth:`switch_candidates`."""
        st = state or self.state
        u = chain_uniforms(self.seed, self.chain_ids, step)
        B, S = st.species.shape
        cand = self.switch_candidates(st)                                                # [B, T, S]
        present = cand.any(axis=2)
        T = present.sum(axis=1)
        valid = T >= 2
        Tm = np.maximum(T, 2)
        k = np.minimum((u[:, 0] * (Tm * (Tm - 1))).astype(np.int64), Tm * (Tm - 1) - 1)   # ordered pairs, uniform
        i1, i2 = k // (Tm - 1), k % (Tm - 1)
        i2 = i2 + (i2 >= i1)
        rank = np.cumsum(present, axis=1) - 1                                           # index among the present types
        type1 = np.argmax(present & (rank == i1[:, None]), axis=1)
        type2 = np.argmax(present & (rank == i2[:, None]), axis=1)
        w = self._site_weights(cand, st)
        rows = np.arange(B)

        def pick(mask, weights, uu):
            ww = np.where(mask, weights, 0.0)
            cum = np.cumsum(ww, axis=1)
            tot = np.maximum(cum[:, -1], 1e-300)
            j = (cum <= (uu * tot)[:, None]).sum(axis=1)           # first site whose cumulative weight exceeds u * total
            j = np.minimum(j, S - 1)
            # guard against round-off landing on a zero-weight site: move to the last candidate at or before j, else the first
            ok = mask[rows, j]
            first = np.argmax(mask, axis=1)
            lastc = S - 1 - np.argmax(mask[:, ::-1], axis=1)
            return np.where(ok, j, np.where(j > lastc, lastc, first))

        site1 = pick(cand[rows, type1], w, u[:, 1])
        w2 = w
        if self.require_distance_decay:
            w2 = w * self.distance_weight_matrix[site1]
        site2 = pick(cand[rows, type2], w2, u[:, 3])
        return site1, site2, type1, type2, valid, u[:, 2]

    # ---- vectorised change_site ------------------------------------------------------------------------------------
    def apply(self, state: ChainState, site, end_code) -> ChainState:
        """The "after" state: remove what is on the site, adsorb ``end_code`` (``n_ads`` = desorb only)."""
        new = state.copy()
        new.energy = None
        b = np.arange(len(site))
        new.species[b, site] = end_code
        adsorb = end_code != self.n_ads
        new.order[b, site] = np.where(adsorb, new.counter, 0)
        new.counter = new.counter + adsorb
        return new

    def occ(self, state: ChainState | None = None) -> np.ndarray:
        """Reference-style ``occ`` ``[B, S]``: atom index of the adsorbate on each site (0 = empty)."""
        st = state or self.state
        filled = st.species != self.n_ads
        key = np.where(filled, st.order, np.iinfo(np.int64).max)
        order = np.argsort(key, axis=1, kind="stable")                       # sites in adsorption order
        sizes = np.take_along_axis(np.where(filled, self.ads_sizes[st.species], 0), order, axis=1)
        first = np.cumsum(sizes, axis=1) - sizes                             # atoms of earlier adsorbates
        start = np.empty_like(first)
        np.put_along_axis(start, order, first, axis=1)
        return np.where(filled, len(self.base) + start, 0)

    def num_adsorbates(self, state: ChainState | None = None) -> np.ndarray:
        st = state or self.state
        return (st.species != self.n_ads).sum(axis=1)

    def num_adsorbate_atoms(self, state: ChainState | None = None) -> np.ndarray:
        st = state or self.state
        return self.ads_sizes[st.species].sum(axis=1)

    def structure(self, b: int, state: ChainState | None = None):
        """Unrelaxed slab of chain b: slab atoms, then adsorbates at their site coordinates in adsorption order
        (``SurfaceSystem.unrelaxed_atoms``, ``mcmc/system.py:349-357``)."""
        st = state or self.state
        sites = np.flatnonzero(st.species[b] != self.n_ads)
        sites = sites[np.argsort(st.order[b, sites], kind="stable")]
        zs, xs = [self.base.numbers], [self.base.positions]
        for site in sites:
            z, offs = self.ads_atoms[int(st.species[b, site])]
            zs.append(z)
            xs.append(self.ads_coords[site][None] + offs)
        return structures.Structure(np.concatenate(zs).astype(np.int32), np.vstack(xs), self.base.cell, self.base.pbc)

    def batch_arrays(self, state: ChainState | None = None, which=None):
        """The unrelaxed slabs of the chains ``which`` (default all) as packed arrays -- ``structure(b)`` for every b without
        a Python loop: ``(n_atoms [b], numbers [sum], positions [sum, 3], ads_chain [n_ads_atoms], ads_numbers)`` where the last
        two name the chain (position in ``which``) and the atomic number of every adsorbate atom (element counts)."""
        st = state or self.state
        sp = st.species if which is None else st.species[np.asarray(which)]
        od = st.order if which is None else st.order[np.asarray(which)]
        b, S = sp.shape
        nb = len(self.base)
        filled = sp != self.n_ads
        key = np.where(filled, od, np.iinfo(np.int64).max)
        by_order = np.argsort(key, axis=1, kind="stable")                    # sites in adsorption order, empty ones last
        cnt = filled.sum(axis=1)
        rows, ranks = np.nonzero(np.arange(S)[None, :] < cnt[:, None])       # adsorbates, chain-major, in adsorption order
        sites = by_order[rows, ranks]
        codes = sp[rows, sites].astype(np.int64)
        sizes = self.ads_sizes[codes]
        tot = int(sizes.sum())
        ent = np.repeat(np.arange(len(rows)), sizes)                         # adsorbate of every adsorbate ATOM
        within = np.arange(tot) - (np.cumsum(sizes) - sizes)[ent]            # its index inside the adsorbate
        z_ads = self._ads_Z[codes[ent], within]
        x_ads = self.ads_coords[sites[ent]] + self._ads_off[codes[ent], within]
        ads_chain = rows[ent]
        n_ads_atoms = np.bincount(ads_chain, minlength=b).astype(np.int64)
        n_atoms = nb + n_ads_atoms
        start = np.concatenate([[0], np.cumsum(n_atoms)]).astype(np.int64)
        numbers = np.empty(int(start[-1]), np.int32)
        positions = np.empty((int(start[-1]), 3))
        base_idx = (start[:-1, None] + np.arange(nb)[None, :]).ravel()
        numbers[base_idx] = np.tile(self.base.numbers, b)
        positions[base_idx] = np.tile(self.base.positions, (b, 1))
        dst = start[ads_chain] + nb + (np.arange(tot) - (np.cumsum(n_ads_atoms) - n_ads_atoms)[ads_chain])
        numbers[dst] = z_ads
        positions[dst] = x_ads
        return n_atoms, numbers, positions, ads_chain, z_ads

    # ---- energies --------------------------------------------------------------------------------------------------
    def _packed_surface_energy(self):
        """How the fast path turns energies into acceptance energies, or None when only the per-slab function can:
        ``("plain",)`` or ``("chem", element order)`` -- the calculator's chemical-potential arithmetic on element counts, valid
        when every adsorbate element already occurs in the base slab (then all chains walk the elements in the base slab's
        order of first appearance, like the scalar function's Counter)."""
        if not self._default_se:      # a user-supplied (energy, structure) -> float function needs the structures
            return None
        if hasattr(self.calc, "surface_energy_of") and getattr(self.calc, "pourbaix_atoms", None):
            return None
        chem_pots, offset_data = getattr(self.calc, "chem_pots", None), getattr(self.calc, "offset_data", None)
        if not (chem_pots and offset_data):
            return ("plain",)
        order = list(dict.fromkeys(int(z) for z in self.base.numbers))
        ads_elements = {int(z) for zs, _ in self.ads_atoms for z in zs}
        if not ads_elements <= set(order):
            return None
        return ("chem", order)

    def _packed_supported(self) -> bool:
        """Whether the calculator's ``evaluate_packed`` serves this ensemble's request (``calc.packed_supported(relax,
        optimizer)``; calculators without the method: everything but relaxations with a host-driven optimizer)."""
        ask = getattr(self.calc, "packed_supported", None)
        if ask is not None:
            return bool(ask(self.relax, self.optimizer))
        return not (self.relax and (not isinstance(self.optimizer, str)
                                    or any(k in self.optimizer for k in ("CG", "LAMMPS", "BFGSLineSearch"))))

    def _evaluate_packed(self, state: ChainState, idx, mode):
        n_atoms, numbers, positions, ads_chain, z_ads = self.batch_arrays(state, idx)
        b, nb = len(idx), len(self.base)
        start = np.concatenate([[0], np.cumsum(n_atoms)]).astype(np.int64)
        cell = np.tile(np.asarray(self.base.cell, float).reshape(1, 9), (b, 1))
        pbc = np.tile(np.asarray(self.base.pbc).astype(np.uint8).reshape(1, 3), (b, 1))
        # fixed_indices None = "the calculator's default" (LAMMPSSurfCalc: the template's bulk group); an explicit list --
        # also an EMPTY one: hold nothing -- always becomes a mask, exactly as the per-slab path passes [indices] * B
        fixed = None
        if self.relax and self.fixed_indices is not None:
            fixed = np.zeros(int(start[-1]), np.uint8)
            if len(self.fixed_indices):
                fixed[(start[:-1, None] + np.asarray(self.fixed_indices, dtype=np.int64)[None, :]).ravel()] = 1
        out = self.calc.evaluate_packed(n_atoms, numbers, positions, cell, pbc, relax=self.relax, fixed_mask=fixed,
                                        relax_steps=self.relax_steps, fmax=self.fmax, optimizer=self.optimizer)
        if self.relax:
            self.oob[idx] = out["oob"]
        ea = out["energy_atoms"]
        self._last_pae = [ea[start[k]:start[k + 1]] for k in range(b)]
        # the acceptance energy is the device's fp64 value where the backend returns it (vssr_batch_energy_f64); "energy"
        # keeps the reference's float32 result word
        raw = np.asarray(out["energy_f64"] if ("energy_f64" in out and self.acceptance_energy == "f64") else out["energy"],
                         dtype=np.float64)
        if mode[0] == "plain":
            energies = raw
        else:
            from .calculators import surface_energy_from_counts

            counts = {}
            for z in mode[1]:
                n_base = int(np.count_nonzero(self.base.numbers == z))
                counts[structures.SYMBOLS[z]] = n_base + np.bincount(ads_chain[z_ads == z], minlength=b)
            energies = surface_energy_from_counts(raw, counts, self.calc.chem_pots, self.calc.offset_data,
                                                  getattr(self.calc, "offset_units", "atomic"))
        self.n_evaluations += b
        relaxed = SlabRefs.of_batch(SlabBatch(n_atoms, numbers, out["positions"], self.base.cell, self.base.pbc))
        return np.asarray(energies, float), relaxed

    def _default_surface_energy(self, energy, struct):
        if hasattr(self.calc, "surface_energy_of") and (getattr(self.calc, "pourbaix_atoms", None)):
            return float(self.calc.surface_energy_of(energy, struct))
        chem_pots, offset_data = getattr(self.calc, "chem_pots", None), getattr(self.calc, "offset_data", None)
        if chem_pots and offset_data:
            from .calculators import surface_energy_from_energy

            return surface_energy_from_energy(energy, struct.get_chemical_symbols(), chem_pots, offset_data,
                                              getattr(self.calc, "offset_units", "atomic"))
        return float(energy)

    def evaluate(self, state: ChainState, which=None):
        """Surface energies of the chains ``which`` (default all) in ``state``: one lock-step batched relaxation
        (or single-point evaluation) of their unrelaxed slabs.  Returns ``(energies, relaxed_structures)``."""
        idx = np.arange(len(state.species)) if which is None else np.asarray(which)
        if self.fast_path and hasattr(self.calc, "evaluate_packed") and len(idx) and self._packed_supported():
            mode = self._packed_surface_energy()
            if mode is not None:
                return self._evaluate_packed(state, idx, mode)
        slabs = [self.structure(int(b), state) for b in idx]
        if self.relax:
            fixed = None if self.fixed_indices is None else [self.fixed_indices] * len(slabs)
            out = self._relax_batch(slabs, fixed)
            # The acceptance energy is the TRUE energy of the relaxed slab: the reference discards optimize_slab's clamped
            # value (mcmc/system.py:466-469 re-evaluates surface_energy on relaxed_atoms) and uses the out-of-bounds flag
            # only to save the offending structure (mcmc/system.py:375-378).  A 240-atom SrTiO3 slab sits near -1870 eV,
            # far beyond the +-1000 eV guard.
            raw = [self._final_energy(o) for o in out]
            relaxed = [o[0] for o in out]
            for b, o in zip(idx, out):
                self.oob[int(b)] = bool(o[3])
            self._last_pae = [o[4].get("per_atom_energies") if len(o) > 4 and isinstance(o[4], dict) else None for o in out]
        else:
            out = self.calc.calculate_batch(slabs)
            raw = [float(o["energy_f64"]) if ("energy_f64" in o and self.acceptance_energy == "f64")
                   else float(np.ravel(o["energy"])[0]) for o in out]
            relaxed = slabs
            self._last_pae = [o.get("per_atom_energies") for o in out]
        self.n_evaluations += len(slabs)
        energies = np.array([self.surface_energy_fn(e, s) for e, s in zip(raw, slabs)], float)
        return energies, relaxed

    def _relax_batch(self, slabs, fixed):
        """``calc.relax_batch`` with this ensemble's optimizer (backends without the keyword use their own)."""
        import inspect

        kw = dict(fixed_indices=fixed, relax_steps=self.relax_steps, fmax=self.fmax)
        try:
            if "optimizer" in inspect.signature(self.calc.relax_batch).parameters:
                kw["optimizer"] = self.optimizer
        except (TypeError, ValueError):
            pass
        return self.calc.relax_batch(slabs, **kw)

    def _final_energy(self, relax_out) -> float:
        """Energy of the relaxed slab from a ``relax_batch`` tuple: the results of its final evaluation when the backend
        returns them, else the tuple's energy field."""
        res = relax_out[4] if len(relax_out) > 4 else None
        if isinstance(res, dict) and "energy_f64" in res and self.acceptance_energy == "f64":
            return float(res["energy_f64"])
        if isinstance(res, dict) and "energy" in res:
            return float(np.ravel(res["energy"])[0])
        return float(relax_out[2])

    def initialize(self):
        """Surface energy of the starting states (the reference evaluates the start state before the first sweep)."""
        self.state.energy, relaxed = self.evaluate(self.state)
        self.relaxed = relaxed if isinstance(relaxed, SlabRefs) else SlabRefs(relaxed)
        self.per_atom_energies = list(self._last_pae)
        return self.state.energy

    # ---- acceptance criteria without energies (DistanceCriterion, TestingCriterion) ---------------------------------------------
    def distance_accept(self, state: ChainState, which=None) -> np.ndarray:
        """``DistanceCriterion`` for the proposed states of the chains ``which``: ``filter_distances`` on the unrelaxed slab
        (``system.real_atoms``) with ``ads = distance_adsorbate_types``.  The base slab's own pairs never change (checked
        once); per chain only the pairs that involve an adsorbate atom are measured."""
        idx = np.arange(len(state.species)) if which is None else np.asarray(which)
        zsel = np.array([structures.ATOMIC_NUMBERS[a] for a in self.distance_adsorbate_types], np.int64)
        if not hasattr(self, "_base_pairs_ok"):
            self._base_pairs_ok = filter_distances(self.base, self.distance_adsorbate_types, self.filter_distance)
            self._base_sel = self.base.positions[np.isin(self.base.numbers, zsel)]
        if not self._base_pairs_ok:
            return np.zeros(len(idx), bool)
        n_atoms, numbers, positions, ads_chain, z_ads = self.batch_arrays(state, idx)
        start = np.concatenate([[0], np.cumsum(n_atoms)]).astype(np.int64)
        nb = len(self.base)
        ok = np.ones(len(idx), bool)
        for k in range(len(idx)):
            za = numbers[start[k] + nb:start[k + 1]]
            xa = positions[start[k] + nb:start[k + 1]][np.isin(za, zsel)]
            if len(xa) == 0:
                continue
            cut = self.filter_distance
            d1 = mic_distance_matrix(xa, self._base_sel, self.base.cell, self.base.pbc) if len(self._base_sel) else np.zeros((len(xa), 0))
            bad = np.any((d1 > 0) & (d1 <= cut))
            if not bad and len(xa) > 1:
                d2 = mic_distance_matrix(xa, xa, self.base.cell, self.base.pbc)[np.triu_indices(len(xa), k=1)]
                bad = np.any((d2 > 0) & (d2 <= cut))
            ok[k] = not bad
        return ok

    def _accept_without_energy(self, before: ChainState, after: ChainState, valid=None) -> np.ndarray:
        """Distance / testing criterion: the accept mask; accepted chains take the proposed occupations, their energies are
        marked stale (the reference evaluates ``get_surface_energy`` once per sweep in these modes, ``mcmc/mcmc.py:290-296``)."""
        B = len(before.species)
        cand = np.ones(B, bool) if valid is None else np.asarray(valid, bool)
        accept = np.zeros(B, bool)
        if self.criterion == "testing":
            accept = cand.copy()
        elif cand.any():
            w = np.flatnonzero(cand)
            accept[w] = self.distance_accept(after, w)
        a2 = accept[:, None]
        energy = None if before.energy is None else before.energy.copy()
        self.state = ChainState(np.where(a2, after.species, before.species), np.where(a2, after.order, before.order),
                                np.where(accept, after.counter, before.counter), energy)
        self.energy_stale |= accept
        return accept

    def refresh_energies(self):
        """Evaluate the chains whose state changed under a criterion that needs no energies (one batched evaluation)."""
        if self.state.energy is None:
            self.initialize()
            self.energy_stale[:] = False
            return self.state.energy
        w = np.flatnonzero(self.energy_stale)
        if len(w):
            e, r = self.evaluate(self.state, w)
            self.state.energy[w] = e
            raw = _raw_items(self.relaxed)
            for b, rb, pb in zip(w, _raw_items(r), self._last_pae):
                raw[int(b)] = rb
                self.per_atom_energies[int(b)] = pb
            self.relaxed = SlabRefs(raw)
            self.energy_stale[:] = False
        return self.state.energy

    # ---- one Change event + Metropolis for every chain --------------------------------------------------------------
    def step_semigrand(self, temperature: float | None = None, site_idx=None, which=None) -> np.ndarray:
        """``MCMC.step_semigrand`` (``mcmc/mcmc.py:233-266``) for all chains at once; returns the accept mask.
        ``site_idx``: an int or ``[B]`` array fixes the site of the change (``ChangeProposal(site_idx=...)``,
        ``mcmc/events/proposal.py:81-84``) instead of drawing it.  ``which``: bool ``[B]``, only these chains take part (the
        others keep their state and are not evaluated) -- how ``prepare_canonical`` stops a chain at its target."""
        temp = self.temp if temperature is None else float(temperature)
        B = len(self.chain_ids)
        part = np.ones(B, bool) if which is None else np.asarray(which, bool)
        if self.criterion != "metropolis":
            self.step_count += 1
            before = self.state
            site, end, start, _ = self.propose(self.step_count, before, site_idx)
            end = np.where(part, end, start)                      # (a chain outside `which` "changes" its site into what is there)
            return self._accept_without_energy(before, self.apply_masked(before, site, end, part), part)
        if self.state.energy is None:
            self.initialize()
        self.step_count += 1
        before = self.state                                   # save_state("before")
        site, end, _, u_acc = self.propose(self.step_count, before, site_idx)
        after = self.apply_masked(before, site, end, part)    # change_site + save_state("after")
        after.energy = before.energy.copy()
        relaxed_after, pae_after = _raw_items(self.relaxed), list(self.per_atom_energies)
        moved = np.flatnonzero(part)
        if len(moved):
            e, r = self.evaluate(after, None if len(moved) == B else moved)    # get_surface_energy(recalculate=True)
            after.energy[moved] = e
            for b, rb, pb in zip(moved, _raw_items(r), self._last_pae):
                relaxed_after[int(b)] = rb
                pae_after[int(b)] = pb
        accept = metropolis_accept(before.energy, after.energy, temp, u_acc) & part
        # accepted chains keep "after", the others are restored to "before" (Event.backward)
        a2 = accept[:, None]
        self.state = ChainState(np.where(a2, after.species, before.species), np.where(a2, after.order, before.order),
                                np.where(accept, after.counter, before.counter),
                                np.where(accept, after.energy, before.energy))
        self.relaxed = SlabRefs([ra if acc else rb for acc, ra, rb in zip(accept, relaxed_after, _raw_items(self.relaxed))]).consolidate()
        self.per_atom_energies = [pa if acc else pb for acc, pa, pb in zip(accept, pae_after, self.per_atom_energies)]
        return accept

    def apply_masked(self, state: ChainState, site, end_code, part) -> ChainState:
        """:meth:`apply` for the chains of the bool mask ``part``; the others are copied unchanged."""
        new = self.apply(state, site, end_code)
        if part.all():
            return new
        p2 = part[:, None]
        return ChainState(np.where(p2, new.species, state.species), np.where(p2, new.order, state.order),
                          np.where(part, new.counter, state.counter), None)

    # ---- MCMC.prepare_canonical (mcmc/mcmc.py:148-188) -----------------------------------------------------------------------
    def even_adsorption_sites(self, n: int) -> np.ndarray:
        """The reference's even seeding (``get_cluster_centers`` + ``find_closest_points_indices``, ``mcmc/utils/clustering.py:160-233``):
        Ward clustering of the in-plane site coordinates into ``n`` clusters, of each the site closest to its centre."""
        from scipy.cluster.hierarchy import fcluster, linkage

        pts = self.ads_coords[:, :2]
        labels = fcluster(linkage(pts, "ward"), int(n), criterion="maxclust")
        out = []
        for i in range(1, int(n) + 1):
            members = np.where(labels == i)[0]
            centre = pts[members].mean(axis=0)
            out.append(members[np.argmin(np.linalg.norm(pts[members] - centre, axis=1))])
        return np.array(out, dtype=np.int64)

    def prepare_canonical(self, num_ads_atoms: int, even_adsorption_sites: bool = False, max_steps: int = 10_000) -> np.ndarray:
        """Bring every chain to the composition a canonical run starts from: semigrand steps (with this ensemble's criterion)
        until a chain holds ``num_ads_atoms`` adsorbates -- it then stops while the others go on -- or, with
        ``even_adsorption_sites``, one semigrand step on each of the evenly spread sites.  Returns the adsorbate counts."""
        if int(num_ads_atoms) <= 0:
            raise ValueError("for canonical runs, need number of adsorbed atoms greater than 0")
        if even_adsorption_sites:
            for site in self.even_adsorption_sites(num_ads_atoms):
                self.step_semigrand(site_idx=int(site))
        else:
            start = self.step_count
            for _ in range(int(max_steps)):
                need = self.num_adsorbates() < int(num_ads_atoms)
                if not need.any():
                    break
                self.step_semigrand(which=need)
            else:
                raise RuntimeError("prepare_canonical: some chains did not reach the requested number of adsorbates")
            # A chain consumed the random numbers of steps start + 1 .. start + k_chain, k_chain <= max_steps, whatever the
            # other chains of this ensemble needed.  The steps AFTER the preparation continue from a fixed index, not from the
            # slowest chain's count: a chain's trajectory stays independent of which chains share its ensemble / group / rank
            # (random numbers are keyed by (seed, global chain id, step)).
            self.step_count = start + int(max_steps)
        if self.criterion != "metropolis":
            self.refresh_energies()
        return self.num_adsorbates()

    # ---- one Exchange event + Metropolis for every chain ------------------------------------------------------------
    def step_canonical(self, temperature: float | None = None) -> np.ndarray:
        """``MCMC.step_canonical`` (``mcmc/mcmc.py:190-231``) for all chains at once: the adsorbates of two sites of
        different type are exchanged (two ``change_site`` calls, ``mcmc/events/event.py:138-156``), the composition is
        conserved.  Chains without two different types keep their state.  Returns the accept mask."""
        temp = self.temp if temperature is None else float(temperature)
        if self.state.energy is None and self.criterion == "metropolis":
            self.initialize()
        self.step_count += 1
        before = self.state
        site1, site2, type1, type2, valid, u_acc = self.propose_switch(self.step_count, before)
        # what actually sits on the two sites is exchanged.  (With reference_groupby, adsorbates that share a first-atom
        # symbol -- "HO" and "O" -- share one candidate entry and type1 / type2 name the entry's FIRST adsorbate: applying
        # that representative would turn a plain O into an HO group and change the composition.)
        rows = np.arange(len(site1))
        code1 = before.species[rows, site1].astype(np.int64)
        code2 = before.species[rows, site2].astype(np.int64)
        if self.exchange_by_group_key:    # the reference's literal behaviour: the group key's own atom arrives (see __init__)
            type1 = np.where(valid, self.key_adsorbate[code1], type1)
            type2 = np.where(valid, self.key_adsorbate[code2], type2)
        else:
            type1 = np.where(valid, code1, type1)
            type2 = np.where(valid, code2, type2)
        after = self.apply(self.apply(before, site1, type2), site2, type1)
        if self.criterion != "metropolis":
            return self._accept_without_energy(before, after, valid)
        moved = np.flatnonzero(valid)
        after.energy = before.energy.copy()
        relaxed_after = _raw_items(self.relaxed)
        pae_after = list(self.per_atom_energies)
        if len(moved):
            e, r = self.evaluate(after, moved)
            after.energy[moved] = e
            for b, rb, pb in zip(moved, _raw_items(r), self._last_pae):
                relaxed_after[int(b)] = rb
                pae_after[int(b)] = pb
        accept = metropolis_accept(before.energy, after.energy, temp, u_acc) & valid
        a2 = accept[:, None]
        self.state = ChainState(np.where(a2, after.species, before.species), np.where(a2, after.order, before.order),
                                np.where(accept, after.counter, before.counter),
                                np.where(accept, after.energy, before.energy))
        self.relaxed = SlabRefs([ra if acc else rb for acc, ra, rb in zip(accept, relaxed_after, _raw_items(self.relaxed))]).consolidate()
        self.per_atom_energies = [pa if acc else pb for acc, pa, pb in zip(accept, pae_after, self.per_atom_energies)]
        return accept

    def sweep(self, i: int = 0, sweep_size: int = 20, temperature: float | None = None, canonical: bool = False) -> dict:
        """``MCMC.sweep`` (``mcmc/mcmc.py:268-299``): ``sweep_size`` steps (semigrand, or exchange moves when
        ``canonical``); per-chain summary."""
        n_acc = np.zeros(len(self.chain_ids), np.int64)
        for _ in range(sweep_size):
            n_acc += self.step_canonical(temperature) if canonical else self.step_semigrand(temperature)
        if self.criterion != "metropolis":
            self.refresh_energies()          # (the reference's sweep ends with surface.get_surface_energy())
        return {"energy": self.state.energy.copy(), "adsorption_count": self.num_adsorbates(),
                "acceptance_rate": n_acc / float(sweep_size), "species": self.state.species.copy()}

    def run(self, total_sweeps: int = 10, sweep_size: int = 20, start_temp: float = 1.0, perform_annealing: bool = True,
            alpha: float = 0.99, multiple_anneal: bool = False, anneal_schedule=None, canonical: bool = False,
            starting_iteration: int = 0, keep_structures: bool = False) -> dict:
        """``MCMC.run`` (``mcmc/mcmc.py:301-392``) without the file outputs: temperature schedule + sweeps
        ``starting_iteration .. total_sweeps - 1``.  The result carries the reference's keys, one entry per sweep, each entry
        an array over the chains: ``energy_hist``, ``frac_accept_hist``, ``adsorption_count_hist``; ``history`` holds the
        ``ChainState`` after the sweep (occupation table, adsorption order and energies of every chain -- ``structure(b, state)``
        turns a row back into the unrelaxed slab) and, with ``keep_structures``, ``trajectories`` the relaxed slabs (``SlabRefs``, one per
        sweep; off by default: B slabs per sweep are the bulk of the memory of a long run).  ``energy`` /
        ``adsorption_count`` / ``acceptance_rate`` are the same lists under this module's own names."""
        if anneal_schedule is not None:
            temps = list(anneal_schedule)
        elif perform_annealing:
            temps = create_anneal_schedule(start_temp, total_sweeps, alpha, multiple_anneal)
        else:
            temps = [start_temp] * total_sweeps
        hist = {"energy": [], "adsorption_count": [], "acceptance_rate": [], "history": [], "temperature": temps}
        if keep_structures:
            hist["trajectories"] = []
        for i in range(starting_iteration, total_sweeps):
            self.temp = float(temps[i])        # (mcmc/mcmc.py:382: the sweep's temperature is the sampler's, also for the proposal weights)
            r = self.sweep(i, sweep_size, temps[i], canonical=canonical)
            for k in ("energy", "adsorption_count", "acceptance_rate"):
                hist[k].append(r[k])
            hist["history"].append(self.state.copy())
            if keep_structures:
                hist["trajectories"].append(self.relaxed)
        hist["energy_hist"] = hist["energy"]
        hist["frac_accept_hist"] = hist["acceptance_rate"]
        hist["adsorption_count_hist"] = hist["adsorption_count"]
        return hist


# ---------------------------------------------------------------------------------------------------------------------
# Chain groups that advance independently: the host work of one group runs while the device evaluates another
# ---------------------------------------------------------------------------------------------------------------------
class ConcurrentChains:
    """Several ``ChainEnsemble`` objects over DISJOINT chains, each with its own calculator (its own engine = its own HIP
    stream), every one advanced by its own host thread.

    Chains never interact and their random numbers are a function of (seed, global chain id, step) only, so a group steps
    through exactly the trajectory its chains have inside one big ensemble (``tests/test_mc.py``: bit for bit).  What the
    grouping buys is overlap ACROSS MC steps: a step is host work (proposal, packing, acceptance: numpy under the GIL) followed
    by a blocking wait on the device (the C ABI call releases the GIL), so while one group waits the other one proposes /
    packs / accepts, and the two engines share the GPU like ``bench.py --streams 2``.  With single-point acceptance energies
    (``relax=False``) the host work is ~30 % of a step of one 256-chain ensemble and is hidden this way (``profiles/r04/bench_mc_groups.txt``:
    17.0 k -> 23.0 k proposals/s with three groups); with relaxations it is 2 % and nothing is gained.

    ``build`` cuts ``n_chains`` into ``len(calcs)`` contiguous ranges of global chain ids."""

    def __init__(self, ensembles):
        self.groups = list(ensembles)
        if not self.groups:
            raise ValueError("no chain groups")
        ids = np.concatenate([g.chain_ids for g in self.groups])
        if len(np.unique(ids)) != len(ids):
            raise ValueError("chain groups overlap: a global chain id may belong to one group only")
        if len({id(g.calc) for g in self.groups}) != len(self.groups):
            raise ValueError("every chain group needs its own calculator (its own engine and HIP stream)")
        self.chain_ids = ids

    @classmethod
    def build(cls, base, ads_coords, adsorbates, n_chains: int, calcs, *, first_chain: int = 0, **kwargs):
        calcs = list(calcs)
        bounds = np.linspace(0, int(n_chains), len(calcs) + 1).astype(int)
        return cls([ChainEnsemble(base, ads_coords, adsorbates, int(bounds[k + 1] - bounds[k]), calc,
                                  first_chain=first_chain + int(bounds[k]), **kwargs)
                    for k, calc in enumerate(calcs) if bounds[k + 1] > bounds[k]])

    def __len__(self):
        return len(self.chain_ids)

    def _each(self, fn):
        """``fn(group)`` for every group, one thread per group; the first exception is raised here."""
        if len(self.groups) == 1:
            return [fn(self.groups[0])]
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(len(self.groups)) as pool:
            return [f.result() for f in [pool.submit(fn, g) for g in self.groups]]

    @property
    def energy(self) -> np.ndarray:
        return np.concatenate([g.state.energy for g in self.groups])

    @property
    def species(self) -> np.ndarray:
        return np.concatenate([g.state.species for g in self.groups])

    @property
    def n_evaluations(self) -> int:
        return sum(g.n_evaluations for g in self.groups)

    def num_adsorbates(self) -> np.ndarray:
        return np.concatenate([g.num_adsorbates() for g in self.groups])

    def initialize(self) -> np.ndarray:
        return np.concatenate(self._each(lambda g: g.initialize()))

    def steps(self, n: int, temperature: float | None = None, canonical: bool = False) -> np.ndarray:
        """``n`` MC steps of every group (each group runs its n steps without waiting for the others); accept counts ``[B]``."""
        def go(g):
            acc = np.zeros(len(g.chain_ids), np.int64)
            for _ in range(n):
                acc += g.step_canonical(temperature) if canonical else g.step_semigrand(temperature)
            return acc
        return np.concatenate(self._each(go))

    def run(self, **kwargs) -> dict:
        """``ChainEnsemble.run`` of every group concurrently; the per-sweep entries are joined along the chain axis."""
        parts = self._each(lambda g: g.run(**kwargs))
        out = {"temperature": parts[0]["temperature"]}
        n_sweeps = len(parts[0]["energy"])
        for k in ("energy", "adsorption_count", "acceptance_rate"):
            out[k] = [np.concatenate([p[k][i] for p in parts]) for i in range(n_sweeps)]
        out["history"] = [ChainState(np.concatenate([p["history"][i].species for p in parts]),
                                     np.concatenate([p["history"][i].order for p in parts]),
                                     np.concatenate([p["history"][i].counter for p in parts]),
                                     np.concatenate([p["history"][i].energy for p in parts])) for i in range(n_sweeps)]
        if "trajectories" in parts[0]:
            out["trajectories"] = [SlabRefs(sum((_raw_items(p["trajectories"][i]) for p in parts), [])) for i in range(n_sweeps)]
        out["energy_hist"], out["frac_accept_hist"] = out["energy"], out["acceptance_rate"]
        out["adsorption_count_hist"] = out["adsorption_count"]
        return out
