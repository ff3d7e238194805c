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
each, adsorbates exchanged.  The per-atom-energy (Boltzmann) and distance-decay site weights of the reference's proposal are
not covered.

Single-atom adsorbates only (the SrTiO3 / GaN configurations of BASELINE.json); the reference's multi-atom
``ATOM_GROUPS`` are not covered.
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


def change_site(state: SiteState, site_idx: int, end_ads: str) -> SiteState:
    """``change_site`` of the reference (``mcmc/slab.py:235-274``) for single-atom adsorbates: remove what sits on the
    site (later atom indices and ``occ`` / ``ads_group`` entries shift down, ``mcmc/slab.py:347-390``), then append the
    new adsorbate at the site coordinate with ``occ[site] = ads_group[-1] =`` its atom index (``:291-309``)."""
    if site_idx >= len(state.occ) or site_idx < 0:
        raise IndexError("site index out of range")
    s = state.copy()
    idx = int(s.occ[site_idx])
    if idx != 0:
        assert np.count_nonzero(s.occ == idx) == 1, "adsorbate index must belong to exactly one site"
        s.numbers = np.delete(s.numbers, idx)
        s.positions = np.delete(s.positions, idx, axis=0)
        s.ads_group = np.delete(s.ads_group, idx)
        s.occ = np.where(s.occ >= idx, s.occ - 1, s.occ)
        s.ads_group = np.where(s.ads_group >= idx, s.ads_group - 1, s.ads_group)
        s.occ = np.where(s.occ < 0, 0, s.occ)
        s.ads_group = np.where(s.ads_group < 0, 0, s.ads_group)
        s.occ[site_idx] = 0
    if end_ads != "None":
        new_idx = len(s.numbers)
        s.numbers = np.append(s.numbers, structures.ATOMIC_NUMBERS[end_ads]).astype(state.numbers.dtype)
        s.positions = np.vstack([s.positions, np.asarray(s.ads_coords[site_idx], float)[None]])
        s.ads_group = np.append(s.ads_group, new_idx)
        s.occ[site_idx] = new_idx
    return s


def metropolis_accept(prev_energy, curr_energy, temperature: float, u) -> np.ndarray:
    """Metropolis rule of the reference (``mcmc/events/criterion.py:157-168``), vectorised: accept where
    ``u < exp(-(E_after - E_before) / kT)`` (overflow of the exponential = certain acceptance)."""
    diff = np.asarray(curr_energy, float) - np.asarray(prev_energy, float)
    with np.errstate(over="ignore", invalid="ignore"):
        prob = np.exp(-diff / float(temperature))
    prob = np.where(np.isnan(prob), 0.0, prob)
    return np.asarray(u, float) < prob


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
        adsorbates: adsorbate symbols, e.g. ``("Sr", "O")`` (``ChangeProposal`` default, ``proposal.py:55``).
        n_chains: chains owned by this process; ``first_chain`` = global id of the first one (sharding).
        calc: energy backend with ``relax_batch(atoms_list, fixed_indices, relax_steps, fmax)`` (relax=True) or
            ``calculate_batch(atoms_list)`` (relax=False), e.g. ``calculators.EnsembleNFFSurface``.
        surface_energy_fn: ``(energy, structure) -> float``; default: the calculator's chemical-potential / bulk
            reference arithmetic when it has ``chem_pots`` and ``offset_data``, else the plain energy.
        fixed_indices: atom indices held fixed during relaxation (``bulk_idx``, ``mcmc/system.py:288-294``).
    """

    def __init__(self, base, ads_coords, adsorbates, n_chains: int, calc, *, seed: int = 0, first_chain: int = 0,
                 relax: bool = True, relax_steps: int = 20, fmax: float = 0.01, fixed_indices=None,
                 surface_energy_fn=None, temperature: float = 1.0, optimizer: str = "BFGS"):
        self.base = base
        self.ads_coords = np.asarray(ads_coords, float).reshape(-1, 3)
        self.adsorbates = list(adsorbates)
        self.n_ads = len(self.adsorbates)
        self.ads_numbers = np.array([structures.ATOMIC_NUMBERS[a] for a in self.adsorbates], np.int32)
        self.calc = calc
        self.seed, self.first_chain = int(seed), int(first_chain)
        self.relax, self.relax_steps, self.fmax = bool(relax), int(relax_steps), float(fmax)
        self.optimizer = optimizer   # reference SrTiO3 configuration: BFGS (scripts/configs/sample_config_painn.json:26)
        self.fixed_indices = None if fixed_indices is None else np.asarray(fixed_indices, np.int64)
        self.temp = float(temperature)
        self.surface_energy_fn = surface_energy_fn or self._default_surface_energy
        B, S = int(n_chains), len(self.ads_coords)
        self.chain_ids = np.arange(self.first_chain, self.first_chain + B, dtype=np.int64)
        self.state = ChainState(np.full((B, S), self.n_ads, np.int16), np.zeros((B, S), np.int64), np.ones(B, np.int64))
        self.step_count = 0
        self.relaxed = [None] * B           # relaxed Structure of the current state of every chain
        self.oob = np.zeros(B, bool)        # out-of-bounds flag of the last evaluation of every chain (reference: energy_oob)
        self.n_evaluations = 0

    # ---- proposal: vectorised ChangeProposal.get_action ------------------------------------------------------------
    def propose(self, step: int, state: ChainState | None = None):
        """Per chain: a uniformly random site, and a uniformly random entry of ``adsorbates + ["None"]`` minus what
        is on the site now (``proposal.py:82-106``).  Returns ``(site_idx [B], end_code [B], start_code [B], u_acc [B])``
        where ``u_acc`` is the uniform reserved for the acceptance test of this step."""
        st = state or self.state
        u = chain_uniforms(self.seed, self.chain_ids, step)
        S = st.species.shape[1]
        site = np.minimum((u[:, 0] * S).astype(np.int64), S - 1)
        start = st.species[np.arange(len(site)), site].astype(np.int64)
        k = np.minimum((u[:, 1] * self.n_ads).astype(np.int64), self.n_ads - 1)   # n_ads + 1 choices minus the current one
        end = np.where(k < start, k, k + 1)
        return site, end, start, u[:, 2]

    # ---- proposal: vectorised SwitchProposal.get_action (uniform weights) -------------------------------------------
    def propose_switch(self, step: int, state: ChainState | None = None):
        """Per chain: an ordered pair of DIFFERENT types present on the lattice (species codes, ``n_ads`` = "None" for
        empty sites; ``random.sample(types, 2)``, ``mcmc/slab.py:60-71``), then a uniformly random site of each type
        (``mcmc/slab.py:217-222``).  Returns ``(site1 [B], site2 [B], type1 [B], type2 [B], valid [B], u_acc [B])``;
        ``valid`` is False for chains with fewer than two types on the lattice (the reference cannot propose there).
        The reference collects the sites of a species with ``itertools.groupby`` over the site order, which keeps only
        the last consecutive run of a species; ALL sites of the species are candidates here."""
        st = state or self.state
        u = chain_uniforms(self.seed, self.chain_ids, step)
        B, S = st.species.shape
        codes = np.arange(self.n_ads + 1)
        counts = (st.species[:, :, None] == codes[None, None, :]).sum(axis=1)          # [B, n_ads + 1]
        present = counts > 0
        T = present.sum(axis=1)
        valid = T >= 2
        Tm = np.maximum(T, 2)
        k = np.minimum((u[:, 0] * (Tm * (Tm - 1))).astype(np.int64), Tm * (Tm - 1) - 1)   # ordered pairs, uniform
        i1, i2 = k // (Tm - 1), k % (Tm - 1)
        i2 = i2 + (i2 >= i1)
        rank = np.cumsum(present, axis=1) - 1                                           # index among the present types
        type1 = np.argmax(present & (rank == i1[:, None]), axis=1)
        type2 = np.argmax(present & (rank == i2[:, None]), axis=1)

        def pick(type_code, uu):
            mask = st.species == type_code[:, None]
            n = np.maximum(mask.sum(axis=1), 1)
            j = np.minimum((uu * n).astype(np.int64), n - 1)
            return np.argmax(mask & ((np.cumsum(mask, axis=1) - 1) == j[:, None]), axis=1)

        site1, site2 = pick(type1, u[:, 1]), pick(type2, u[:, 3])
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
        rank = np.argsort(np.argsort(key, axis=1, kind="stable"), axis=1, kind="stable")
        return np.where(filled, len(self.base) + rank, 0)

    def num_adsorbates(self, state: ChainState | None = None) -> np.ndarray:
        st = state or self.state
        return (st.species != self.n_ads).sum(axis=1)

    def structure(self, b: int, state: ChainState | None = None):
        """Unrelaxed slab of chain b: slab atoms, then adsorbates at their site coordinates in adsorption order
        (``SurfaceSystem.unrelaxed_atoms``, ``mcmc/system.py:349-357``)."""
        st = state or self.state
        sites = np.flatnonzero(st.species[b] != self.n_ads)
        sites = sites[np.argsort(st.order[b, sites], kind="stable")]
        numbers = np.concatenate([self.base.numbers, self.ads_numbers[st.species[b, sites]]]).astype(np.int32)
        positions = np.vstack([self.base.positions, self.ads_coords[sites]]) if len(sites) else self.base.positions.copy()
        return structures.Structure(numbers, positions, self.base.cell, self.base.pbc)

    # ---- energies --------------------------------------------------------------------------------------------------
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
        else:
            out = self.calc.calculate_batch(slabs)
            raw = [float(np.ravel(o["energy"])[0]) for o in out]
            relaxed = slabs
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

    @staticmethod
    def _final_energy(relax_out) -> float:
        """Energy of the relaxed slab from a ``relax_batch`` tuple: the results of its final evaluation when the backend
        returns them, else the tuple's energy field."""
        res = relax_out[4] if len(relax_out) > 4 else None
        if isinstance(res, dict) and "energy" in res:
            return float(np.ravel(res["energy"])[0])
        return float(relax_out[2])

    def initialize(self):
        """Surface energy of the starting states (the reference evaluates the start state before the first sweep)."""
        self.state.energy, self.relaxed = self.evaluate(self.state)
        return self.state.energy

    # ---- one Change event + Metropolis for every chain --------------------------------------------------------------
    def step_semigrand(self, temperature: float | None = None) -> np.ndarray:
        """``MCMC.step_semigrand`` (``mcmc/mcmc.py:233-266``) for all chains at once; returns the accept mask."""
        temp = self.temp if temperature is None else float(temperature)
        if self.state.energy is None:
            self.initialize()
        self.step_count += 1
        before = self.state                                   # save_state("before")
        site, end, _, u_acc = self.propose(self.step_count, before)
        after = self.apply(before, site, end)                 # change_site + save_state("after")
        after.energy, relaxed_after = self.evaluate(after)    # get_surface_energy(recalculate=True)
        accept = metropolis_accept(before.energy, after.energy, temp, u_acc)
        # accepted chains keep "after", the others are restored to "before" (Event.backward)
        a2 = accept[:, None]
        self.state = ChainState(np.where(a2, after.species, before.species), np.where(a2, after.order, before.order),
                                np.where(accept, after.counter, before.counter),
                                np.where(accept, after.energy, before.energy))
        self.relaxed = [ra if acc else rb for acc, ra, rb in zip(accept, relaxed_after, self.relaxed)]
        return accept

    # ---- one Exchange event + Metropolis for every chain ------------------------------------------------------------
    def step_canonical(self, temperature: float | None = None) -> np.ndarray:
        """``MCMC.step_canonical`` (``mcmc/mcmc.py:190-231``) for all chains at once: the adsorbates of two sites of
        different type are exchanged (two ``change_site`` calls, ``mcmc/events/event.py:138-156``), the composition is
        conserved.  Chains without two different types keep their state.  Returns the accept mask."""
        temp = self.temp if temperature is None else float(temperature)
        if self.state.energy is None:
            self.initialize()
        self.step_count += 1
        before = self.state
        site1, site2, type1, type2, valid, u_acc = self.propose_switch(self.step_count, before)
        after = self.apply(self.apply(before, site1, type2), site2, type1)
        moved = np.flatnonzero(valid)
        after.energy = before.energy.copy()
        relaxed_after = list(self.relaxed)
        if len(moved):
            e, r = self.evaluate(after, moved)
            after.energy[moved] = e
            for b, rb in zip(moved, r):
                relaxed_after[int(b)] = rb
        accept = metropolis_accept(before.energy, after.energy, temp, u_acc) & valid
        a2 = accept[:, None]
        self.state = ChainState(np.where(a2, after.species, before.species), np.where(a2, after.order, before.order),
                                np.where(accept, after.counter, before.counter),
                                np.where(accept, after.energy, before.energy))
        self.relaxed = [ra if acc else rb for acc, ra, rb in zip(accept, relaxed_after, self.relaxed)]
        return accept

    def sweep(self, i: int = 0, sweep_size: int = 20, temperature: float | None = None, canonical: bool = False) -> dict:
        """``MCMC.sweep`` (``mcmc/mcmc.py:268-299``): ``sweep_size`` steps (semigrand, or exchange moves when
        ``canonical``); per-chain summary."""
        n_acc = np.zeros(len(self.chain_ids), np.int64)
        for _ in range(sweep_size):
            n_acc += self.step_canonical(temperature) if canonical else self.step_semigrand(temperature)
        return {"energy": self.state.energy.copy(), "adsorption_count": self.num_adsorbates(),
                "acceptance_rate": n_acc / float(sweep_size), "species": self.state.species.copy()}

    def run(self, total_sweeps: int = 10, sweep_size: int = 20, start_temp: float = 1.0, perform_annealing: bool = True,
            alpha: float = 0.99, multiple_anneal: bool = False, anneal_schedule=None, canonical: bool = False) -> dict:
        """``MCMC.run`` (``mcmc/mcmc.py:301-420``) without the file outputs: temperature schedule + sweeps."""
        if anneal_schedule is not None:
            temps = list(anneal_schedule)
        elif perform_annealing:
            temps = create_anneal_schedule(start_temp, total_sweeps, alpha, multiple_anneal)
        else:
            temps = [start_temp] * total_sweeps
        hist = {"energy": [], "adsorption_count": [], "acceptance_rate": [], "temperature": temps}
        for i in range(total_sweeps):
            r = self.sweep(i, sweep_size, temps[i], canonical=canonical)
            for k in ("energy", "adsorption_count", "acceptance_rate"):
                hist[k].append(r[k])
        return hist
