"""Batched MC proposal / change / acceptance — host logic, no GPU.

Mirrors the reference's tests for this path (``tests/test_slab.py:41-87`` change_site, ``tests/events/test_criterion.py:
14-46`` Metropolis) and adds the properties the batched design must keep: the vectorised state is equivalent to the
reference's per-chain index bookkeeping, trajectories do not depend on batching / sharding (counter-based RNG), and the
chain samples the Boltzmann distribution of a toy lattice gas (detailed balance)."""
import itertools

import numpy as np
import pytest

from surface_sampling_amd import mc, structures


# ---------------------------------------------------------------------------------------------------------------------
def test_philox4x32_known_answers():
    """Random123 known-answer vectors for Philox4x32-10."""
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
         (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    for ctr, key, out in kat:
        got = mc.philox4x32(np.array(ctr, np.uint64), np.array(key, np.uint64))
        assert tuple(int(x) for x in got) == out
    # vectorised call = element-wise calls
    ctrs = np.array([k[0] for k in kat], np.uint64)
    keys = np.array([k[1] for k in kat], np.uint64)
    assert np.array_equal(mc.philox4x32(ctrs, keys), np.array([k[2] for k in kat], np.uint32))


def test_chain_uniforms_depend_only_on_seed_chain_step():
    u_all = mc.chain_uniforms(7, np.arange(16), step=5)
    u_part = mc.chain_uniforms(7, np.arange(8, 16), step=5)
    assert np.array_equal(u_all[8:], u_part)
    assert (u_all >= 0).all() and (u_all < 1).all()
    assert not np.array_equal(u_all, mc.chain_uniforms(7, np.arange(16), step=6))
    assert not np.array_equal(u_all, mc.chain_uniforms(8, np.arange(16), step=5))
    big = mc.chain_uniforms(1, np.arange(20000), step=1)
    assert abs(big.mean() - 0.5) < 0.01 and abs(big.var() - 1 / 12) < 0.005


# ---- change_site: the reference's fixture (tests/test_slab.py:21-33) ------------------------------------------------
@pytest.fixture()
def system():
    Z = structures.ATOMIC_NUMBERS
    return mc.SiteState(numbers=np.array([Z["Ga"], Z["As"], Z["Ga"], Z["As"]], np.int32),
                        positions=np.array([[0, 0, 0], [0, 0, 3], [1, 1, 1], [1, 1, 4]], float),
                        ads_group=np.array([0, 1, 2, 0]), occ=np.array([1, 2, 0]),
                        ads_coords=np.array([(0, 0, 3), (1, 1, 1), (2, 2, 5)], float))


def test_change_site_with_existing_adsorbate(system):
    new = mc.change_site(system, 0, "O")
    assert len(new) == 4
    assert np.allclose(new.occ, [3, 1, 0])
    assert new.symbols[3] == "O"
    assert np.allclose(new.ads_group, [0, 1, 0, 3])
    assert np.allclose(new.positions[3], [0, 0, 3])


def test_change_site_with_empty_site(system):
    new = mc.change_site(system, 2, "Ir")
    assert len(new) == 5
    assert np.allclose(new.occ, [1, 2, 4])
    assert new.symbols[4] == "Ir"
    assert np.allclose(new.ads_group, [0, 1, 2, 0, 4])


def test_change_site_with_desorption(system):
    new = mc.change_site(system, 0, "None")
    assert len(new) == 3
    assert np.allclose(new.occ, [0, 1, 0])
    assert new.symbols[2] == "As"
    assert np.allclose(new.ads_group, [0, 1, 0])
    assert len(system) == 4 and np.allclose(system.occ, [1, 2, 0])   # the "before" state is untouched


def test_change_site_with_invalid_site_index(system):
    with pytest.raises(IndexError):
        mc.change_site(system, 5, "As")


# ---- Metropolis (tests/events/test_criterion.py:14-46) --------------------------------------------------------------
def test_metropolis_criterion_accept_reject_equal():
    kT = 0.0257
    assert mc.metropolis_accept(10.0, 5.0, kT, 0.999999)            # downhill: always
    assert not mc.metropolis_accept(10.0, 15.0, kT, 1e-12)          # +5 eV at 300 K: exp(-194) never
    assert mc.metropolis_accept(5.0, 5.0, kT, 0.999999)             # no energy change: exp(0) = 1
    # vectorised, overflow (huge downhill) and +inf (out-of-bounds clamp) handled
    acc = mc.metropolis_accept(np.array([0.0, 0.0, 0.0]), np.array([-1e6, 1e6, 0.01]), kT, np.array([0.9, 0.0, 0.5]))
    assert acc.tolist() == [True, False, True]
    p = np.exp(-0.01 / kT)
    u = np.linspace(0, 1, 10001)[:-1]
    assert abs(mc.metropolis_accept(0.0, 0.01, kT, u).mean() - p) < 2e-4


def test_anneal_schedule_matches_reference_recurrence():
    t = mc.create_anneal_schedule(start_temp=1.0, total_sweeps=5, alpha=0.9)
    assert np.allclose(t, [1.0, 0.9, 0.81, 0.729, 0.6561])
    t = mc.create_anneal_schedule(start_temp=0.2, total_sweeps=520, multiple_anneal=True)
    assert len(t) == 520 and t[0] == 0.2 and abs(t[100] - 0.10) < 1e-12 and abs(t[300] - 0.08) < 1e-12
    assert abs(t[501] - 0.08) < 1e-12 and t[510] == pytest.approx(0.2)


# ---------------------------------------------------------------------------------------------------------------------
class LatticeGasCalc:
    """Toy energy backend with the calculators' batch interface: E = sum_ads eps[Z] + J * (# adsorbate pairs closer than
    r0).  Counts calls so that tests can check the lock-step batching."""

    def __init__(self, n_base, eps, J=0.0, r0=1.5):
        self.n_base, self.eps, self.J, self.r0 = n_base, eps, J, r0
        self.calls = 0

    def _energy(self, s):
        z, x = s.numbers[self.n_base:], s.positions[self.n_base:]
        e = sum(self.eps[int(k)] for k in z)
        for i, j in itertools.combinations(range(len(z)), 2):
            if np.linalg.norm(x[i] - x[j]) < self.r0:
                e += self.J
        return e

    def calculate_batch(self, slabs):
        self.calls += 1
        return [{"energy": np.array([self._energy(s)])} for s in slabs]

    def relax_batch(self, slabs, fixed_indices=None, relax_steps=20, fmax=0.01):
        self.calls += 1
        return [(s.copy(), None, self._energy(s), False, {}) for s in slabs]


def _toy(n_chains, n_sites=6, first_chain=0, seed=3, relax=False, **kw):
    Z = structures.ATOMIC_NUMBERS
    base = structures.Structure(np.array([Z["Ti"], Z["Ti"]], np.int32), np.array([[0, 0, 0], [2.0, 0, 0]], float),
                                np.diag([20.0, 20.0, 20.0]), np.array([True, True, False]))
    coords = np.array([[1.0 * s, 0.0, 2.0] for s in range(n_sites)], float)
    calc = LatticeGasCalc(2, {Z["Sr"]: -0.05, Z["O"]: 0.02}, J=0.03)
    ens = mc.ChainEnsemble(base, coords, ("Sr", "O"), n_chains, calc, seed=seed, first_chain=first_chain, relax=relax,
                           temperature=0.05, **kw)
    return ens, calc


def test_proposals_follow_the_reference_rule():
    ens, _ = _toy(4000)
    rng = np.random.default_rng(0)
    ens.state.species[:] = rng.integers(0, 3, ens.state.species.shape)   # 0 = Sr, 1 = O, 2 = empty
    site, end, start, u = ens.propose(step=1)
    assert ((site >= 0) & (site < 6)).all() and ((end >= 0) & (end <= 2)).all()
    assert (end != start).all()                           # never the adsorbate already on the site / "None" for empty
    assert (start == ens.state.species[np.arange(4000), site]).all()
    # uniform over sites and over the two remaining choices
    assert np.abs(np.bincount(site, minlength=6) / 4000 - 1 / 6).max() < 0.03
    for s0 in range(3):
        sel = start == s0
        others = [c for c in range(3) if c != s0]
        frac = (end[sel] == others[0]).mean()
        assert abs(frac - 0.5) < 0.06
    assert ((u >= 0) & (u < 1)).all()


def test_batched_state_equals_reference_bookkeeping():
    """Random change sequences: the [B, S] arrays and the reference's per-chain occ / atom-order bookkeeping agree."""
    ens, _ = _toy(5)
    B, S = ens.state.species.shape
    singles = [mc.SiteState(ens.base.numbers.copy(), ens.base.positions.copy(), np.zeros(len(ens.base), np.int64),
                            np.zeros(S, np.int64), ens.ads_coords) for _ in range(B)]
    state = ens.state
    names = ens.adsorbates + ["None"]
    for step in range(1, 60):
        site, end, _, _ = ens.propose(step, state)
        state = ens.apply(state, site, end)
        singles = [mc.change_site(s, int(site[b]), names[int(end[b])]) for b, s in enumerate(singles)]
        occ = ens.occ(state)
        for b in range(B):
            st = ens.structure(b, state)
            assert np.array_equal(st.numbers, singles[b].numbers)
            assert np.allclose(st.positions, singles[b].positions)
            assert np.array_equal(occ[b], singles[b].occ)
    assert (ens.num_adsorbates(state) == [np.count_nonzero(s.occ) for s in singles]).all()


def test_trajectories_do_not_depend_on_batching_or_sharding():
    whole, calc_w = _toy(8)
    lo, _ = _toy(4, first_chain=0)
    hi, _ = _toy(4, first_chain=4)
    for ens in (whole, lo, hi):
        ens.initialize()
    acc_w = [whole.step_semigrand() for _ in range(40)]
    acc_l = [lo.step_semigrand() for _ in range(40)]
    acc_h = [hi.step_semigrand() for _ in range(40)]
    assert np.array_equal(np.array(acc_w), np.hstack([np.array(acc_l), np.array(acc_h)]))
    assert np.array_equal(whole.state.species, np.vstack([lo.state.species, hi.state.species]))
    assert np.allclose(whole.state.energy, np.hstack([lo.state.energy, hi.state.energy]))
    assert 0 < np.mean(acc_w) < 1
    assert calc_w.calls == 41                              # one lock-step batched evaluation per MC step (+ the start)
    # rejected chains keep the energy of the restored "before" state
    e_check, _ = whole.evaluate(whole.state)
    assert np.allclose(e_check, whole.state.energy)


def test_relaxation_path_uses_relax_batch_and_fixed_atoms():
    ens, calc = _toy(3, relax=True, fixed_indices=[0, 1], relax_steps=7, fmax=0.02)
    seen = {}
    orig = calc.relax_batch

    def spy(slabs, fixed_indices=None, relax_steps=20, fmax=0.01):
        seen.update(n=len(slabs), fixed=fixed_indices, steps=relax_steps, fmax=fmax)
        return orig(slabs, fixed_indices, relax_steps, fmax)

    calc.relax_batch = spy
    ens.initialize()
    ens.step_semigrand()
    assert seen["n"] == 3 and seen["steps"] == 7 and seen["fmax"] == 0.02
    assert all(np.array_equal(f, [0, 1]) for f in seen["fixed"])
    assert all(r is not None for r in ens.relaxed)


def test_acceptance_uses_the_true_energy_beyond_the_out_of_bounds_guard():
    """Slabs with |E| > 1000 eV (any SrTiO3 slab above ~130 atoms): the reference re-evaluates the surface energy on the
    relaxed slab and uses the clamped 1000 only as a flag (mcmc/system.py:375-378,466-469).  A backend that clamps like
    optimize_slab (mcmc/dynamics.py:159-168) must not flatten the Metropolis energies."""
    ens, calc = _toy(32, relax=True, seed=5)
    shift = -1870.0

    def clamping_relax(slabs, fixed_indices=None, relax_steps=20, fmax=0.01):
        out = []
        for s_ in slabs:
            e = calc._energy(s_) + shift
            oob = abs(e) > 1000.0
            out.append((s_.copy(), None, 1000.0 if oob else e, oob, {"energy": np.array([e], np.float32)}))
        return out

    calc.relax_batch = clamping_relax
    e0 = ens.initialize()
    assert np.allclose(e0, shift, atol=1e-3) and ens.oob.all()
    for _ in range(6):
        ens.step_semigrand()
    expected = np.array([calc._energy(ens.structure(b)) + shift for b in range(32)])
    assert np.allclose(ens.state.energy, expected, atol=2e-3)      # float32 results, true energies
    assert len(np.unique(np.round(ens.state.energy, 3))) > 1           # not the constant 1000


def test_detailed_balance_on_a_two_site_lattice_gas():
    """2 sites, adsorbates {Sr, O}: 9 states with known energies -> visit frequencies follow exp(-E/kT)."""
    ens, calc = _toy(512, n_sites=2, seed=11)
    ens.temp = 0.05
    ens.initialize()
    for _ in range(150):                                   # burn-in
        ens.step_semigrand()
    counts = np.zeros((3, 3))
    for _ in range(400):
        ens.step_semigrand()
        np.add.at(counts, (ens.state.species[:, 0], ens.state.species[:, 1]), 1)
    freq = counts / counts.sum()
    Z = structures.ATOMIC_NUMBERS
    eps = [calc.eps[Z["Sr"]], calc.eps[Z["O"]], 0.0]
    E = np.array([[eps[a] + eps[b] + (calc.J if a < 2 and b < 2 else 0.0) for b in range(3)] for a in range(3)])
    boltz = np.exp(-E / ens.temp)
    boltz /= boltz.sum()
    assert np.abs(freq - boltz).max() < 0.02, (freq, boltz)


def test_run_with_annealing_returns_per_chain_history():
    ens, _ = _toy(6)
    hist = ens.run(total_sweeps=3, sweep_size=5, start_temp=0.1, alpha=0.5)
    assert np.allclose(hist["temperature"], [0.1, 0.05, 0.025])
    assert len(hist["energy"]) == 3 and hist["energy"][0].shape == (6,)
    assert ((hist["acceptance_rate"][0] >= 0) & (hist["acceptance_rate"][0] <= 1)).all()
    assert ens.step_count == 15
    # the reference's result keys (mcmc/mcmc.py:383-390), one entry per sweep
    for ref_key, own in (("energy_hist", "energy"), ("frac_accept_hist", "acceptance_rate"), ("adsorption_count_hist", "adsorption_count")):
        assert hist[ref_key] is hist[own] and len(hist[ref_key]) == 3
    assert len(hist["history"]) == 3 and np.array_equal(hist["history"][-1].species, ens.state.species)
    assert np.array_equal(hist["history"][-1].energy, hist["energy_hist"][-1]) and "trajectories" not in hist
    slab = ens.structure(2, hist["history"][0])                      # a history row turns back into its slab
    assert len(slab) == len(ens.base) + int(hist["adsorption_count_hist"][0][2])
    # resuming at sweep 2 runs only the last sweep, at that sweep's temperature; relaxed slabs are kept on request
    ens2, _ = _toy(6)
    h2 = ens2.run(total_sweeps=3, sweep_size=5, start_temp=0.1, alpha=0.5, starting_iteration=2, keep_structures=True)
    assert len(h2["energy_hist"]) == 1 and ens2.step_count == 5 and ens2.temp == 0.025
    assert len(h2["trajectories"]) == 1 and len(h2["trajectories"][0]) == 6


# ---- canonical (switch) moves: mirrors tests/events/test_proposal.py:79-110, tests/test_slab.py:153-183, ----------------
# ---- tests/events/test_event.py:84-131 of the reference ----------------------------------------------------------------
def _seeded_ensemble(n_chains=64, n_sites=8, seed=11):
    ens, _ = _toy(n_chains, n_sites=n_sites, seed=seed)
    rng = np.random.default_rng(5)
    # random starting occupations: species codes 0..n_ads (n_ads = empty), adsorption order = site order
    sp = rng.integers(0, ens.n_ads + 1, size=(n_chains, n_sites)).astype(np.int16)
    sp[0] = ens.n_ads                      # chain 0: empty lattice (one type only)
    sp[1] = 0                              # chain 1: full lattice of one species (one type only)
    order = np.where(sp != ens.n_ads, np.cumsum(sp != ens.n_ads, axis=1), 0)
    ens.state = mc.ChainState(sp, order.astype(np.int64), (order.max(axis=1) + 1).astype(np.int64))
    return ens


def test_switch_proposal_picks_two_sites_of_different_type():
    ens = _seeded_ensemble()
    for step in range(1, 30):
        s1, s2, t1, t2, valid, u = ens.propose_switch(step)
        b = np.arange(len(s1))
        assert not valid[0] and not valid[1] and valid[2:].sum() > 0
        v = np.flatnonzero(valid)
        assert (t1[v] != t2[v]).all() and (s1[v] != s2[v]).all()
        assert (ens.state.species[b, s1][v] == t1[v]).all() and (ens.state.species[b, s2][v] == t2[v]).all()
        assert ((u >= 0) & (u < 1)).all()


def test_switch_proposal_is_uniform_over_ordered_type_pairs_and_sites():
    ens, _ = _toy(1, n_sites=5, seed=2)
    ens.state.species[0] = [0, 1, 1, ens.n_ads, 0]      # types present: 0 (sites 0, 4), 1 (sites 1, 2), None (site 3)
    ens.state.order[0] = [1, 2, 3, 0, 4]
    ens.state.counter[0] = 5
    pairs, sites = {}, {}
    n = 6000
    for step in range(n):
        s1, s2, t1, t2, valid, _ = ens.propose_switch(step)
        assert valid[0]
        pairs[(int(t1[0]), int(t2[0]))] = pairs.get((int(t1[0]), int(t2[0])), 0) + 1
        sites[int(s1[0])] = sites.get(int(s1[0]), 0) + 1
    assert len(pairs) == 6 and all(abs(c / n - 1 / 6) < 0.02 for c in pairs.values())
    # site of type 1 / 2: uniform inside the type -> sites 0, 1, 2, 4 each 1/6, site 3 ("None", alone) 1/3
    for site, want in {0: 1 / 6, 1: 1 / 6, 2: 1 / 6, 4: 1 / 6, 3: 1 / 3}.items():
        assert abs(sites[site] / n - want) < 0.02


def test_exchange_conserves_composition_and_is_restored_on_rejection():
    ens = _seeded_ensemble(n_chains=32)
    ens.initialize()
    before = ens.state.copy()
    comp0 = np.sort(before.species, axis=1)
    acc = ens.step_canonical(temperature=1e-9)              # practically only downhill moves are accepted
    assert np.array_equal(np.sort(ens.state.species, axis=1), comp0)             # composition conserved
    rej = ~acc
    assert np.array_equal(ens.state.species[rej], before.species[rej])           # Event.backward
    assert np.array_equal(ens.state.order[rej], before.order[rej])
    assert np.allclose(ens.state.energy[rej], before.energy[rej])
    changed = (ens.state.species != before.species).any(axis=1)
    assert np.array_equal(changed, acc)                                          # an accepted exchange moves two sites
    assert ((ens.state.species != before.species).sum(axis=1)[acc] == 2).all()
    assert not acc[0] and not acc[1]                                             # one type only: nothing to exchange
    # always-accept limit (TestingCriterion): every valid chain moves
    ens2 = _seeded_ensemble(n_chains=32)
    ens2.initialize()
    acc2 = ens2.step_canonical(temperature=1e12)
    assert acc2[2:].all()


def test_canonical_trajectories_do_not_depend_on_batching():
    whole = _seeded_ensemble(n_chains=16, seed=4)
    whole.run(total_sweeps=2, sweep_size=5, perform_annealing=False, canonical=True)
    part, _ = _toy(8, n_sites=8, first_chain=8, seed=4)
    ref = _seeded_ensemble(n_chains=16, seed=4)
    part.state = mc.ChainState(ref.state.species[8:].copy(), ref.state.order[8:].copy(), ref.state.counter[8:].copy())
    part.run(total_sweeps=2, sweep_size=5, perform_annealing=False, canonical=True)
    assert np.array_equal(whole.state.species[8:], part.state.species)
    assert np.allclose(whole.state.energy[8:], part.state.energy)


# ---- multi-atom adsorbates: mirrors the reference's tests/test_slab_groups.py:41-87 (same fixture, same expected arrays) ----
def _group_fixture():
    Z = structures.ATOMIC_NUMBERS
    return mc.SiteState(np.array([Z["Ga"], Z["As"], Z["Ga"], Z["As"]]),
                        np.array([[0, 0, 0], [0, 0, 3], [1, 1, 1], [1, 1, 4]], float), np.array([0, 1, 2, 0]),
                        np.array([1, 2, 0]), np.array([(0, 0, 3), (1, 1, 1), (2, 2, 5)], float))


def test_change_site_groups_follow_the_reference_sequence():
    s0 = _group_fixture()
    a = mc.change_site(s0, 0, "HO")                       # existing adsorbate -> group
    assert len(a) == 5 and np.array_equal(a.occ, [3, 1, 0]) and a.symbols[3:] == ["O", "H"]
    assert np.array_equal(a.ads_group, [0, 1, 0, 3, 3])
    assert np.allclose(a.positions[3], [0, 0, 3]) and np.allclose(a.positions[4], [1, 0, 3])   # group offsets at the site
    b = mc.change_site(a, 2, "Ir")                        # empty site -> atom, behind the group
    assert len(b) == 6 and np.array_equal(b.occ, [3, 1, 5]) and b.symbols[5] == "Ir"
    assert np.array_equal(b.ads_group, [0, 1, 0, 3, 3, 5])
    c = mc.change_site(b, 0, "None")                      # remove the group: two atoms leave, later indices drop by 2
    assert len(c) == 4 and np.array_equal(c.occ, [0, 1, 3]) and c.symbols[3] == "Ir"
    assert np.array_equal(c.ads_group, [0, 1, 0, 3])
    d = mc.change_site(c, 1, "None")                      # remove a single atom
    assert len(d) == 3 and np.array_equal(d.occ, [0, 0, 2]) and d.symbols[2] == "Ir"
    assert np.array_equal(d.ads_group, [0, 0, 2])
    w = mc.change_site(s0, 2, "H2O")
    assert w.symbols[4:] == ["O", "H", "H"] and np.array_equal(w.ads_group[4:], [4, 4, 4]) and w.occ[2] == 4
    assert np.allclose(np.linalg.norm(w.positions[5:] - w.positions[4], axis=1), 1.0)


def _group_ensemble(n_chains=40, **kw):
    Z = structures.ATOMIC_NUMBERS
    base = structures.Structure(np.array([Z["Ti"], Z["Ti"]], np.int32), np.array([[0, 0, 0], [2.0, 0, 0]], float),
                                np.diag([20.0, 20.0, 20.0]), np.array([True, True, False]))
    coords = np.array([[1.5 * s, 0.0, 2.0] for s in range(7)], float)
    calc = LatticeGasCalc(2, {Z["Sr"]: -0.05, Z["O"]: 0.02, Z["H"]: 0.01}, J=0.0)
    return mc.ChainEnsemble(base, coords, ("Sr", "O", "HO", "H2O"), n_chains, calc, seed=9, relax=False, temperature=0.05,
                            **kw), calc


def test_batched_groups_equal_reference_bookkeeping():
    """Random sequences of changes with atoms and groups: the batched state reproduces, chain by chain, the atom order,
    positions, ``occ`` and the group membership of the reference-style single-chain bookkeeping."""
    ens, _ = _group_ensemble()
    B, S = ens.state.species.shape
    refs = [mc.SiteState(ens.base.numbers.copy(), ens.base.positions.copy(), np.zeros(len(ens.base), np.int64),
                         np.zeros(S, np.int64), ens.ads_coords) for _ in range(B)]
    names = ens.adsorbates + ["None"]
    state = ens.state
    for step in range(1, 30):
        site, end, _, _ = ens.propose(step, state)
        state = ens.apply(state, site, end)
        refs = [mc.change_site(r, int(si), names[int(e)]) for r, si, e in zip(refs, site, end)]
    occ = ens.occ(state)
    assert (ens.num_adsorbate_atoms(state) > ens.num_adsorbates(state)).any()          # groups are present
    for b in range(B):
        st = ens.structure(b, state)
        assert np.array_equal(st.numbers, refs[b].numbers) and np.allclose(st.positions, refs[b].positions)
        assert np.array_equal(occ[b], refs[b].occ)
        assert len(st) == len(ens.base) + ens.num_adsorbate_atoms(state)[b]


# ---- switch proposals: the reference's candidate sets and weights --------------------------------------------------------
def test_reference_groupby_keeps_the_last_consecutive_run_of_a_species():
    """``get_adsorbate_indices`` (mcmc/slab.py:36-57): groupby over the filled sites in site order, later runs of a key
    replace earlier ones.  Sites: Sr . O Sr Sr . O  ->  reference candidates Sr: {3, 4}, O: {6}, None: {1, 5}; the default mode
    offers every site of the species."""
    ens, _ = _toy(1, n_sites=7)
    SR, O, E = 0, 1, ens.n_ads
    sp = np.array([[SR, E, O, SR, SR, E, O]], np.int16)
    order = np.where(sp != E, np.cumsum(sp != E, axis=1), 0)
    st = mc.ChainState(sp, order.astype(np.int64), np.array([order.max() + 1]))
    cand = ens.switch_candidates(st)[0]
    assert [np.flatnonzero(c).tolist() for c in cand] == [[0, 3, 4], [2, 6], [1, 5]]
    ens.reference_groupby = True
    cand = ens.switch_candidates(st)[0]
    assert [np.flatnonzero(c).tolist() for c in cand] == [[3, 4], [6], [1, 5]]
    # the reference's own fixture (tests/test_slab.py:84-87): occ [1, 2, 0] -> {"As": [0], "Ga": [1], "None": [2]}
    sp2 = np.array([[O, SR, E]], np.int16)
    st2 = mc.ChainState(sp2, np.array([[1, 2, 0]]), np.array([3]))
    ens2, _ = _toy(1, n_sites=3)
    ens2.reference_groupby = True
    assert [np.flatnonzero(c).tolist() for c in ens2.switch_candidates(st2)[0]] == [[1], [0], [2]]
    # proposals in reference mode only ever use the reference's candidates
    ens.chain_ids = np.arange(500)
    big = mc.ChainState(np.repeat(sp, 500, 0), np.repeat(order, 500, 0).astype(np.int64), np.full(500, order.max() + 1))
    s1, s2, t1, t2, valid, _ = ens.propose_switch(3, big)
    allowed = {SR: {3, 4}, O: {6}, E: {1, 5}}
    assert valid.all() and all(int(a) in allowed[int(t)] for a, t in zip(s1, t1))
    assert all(int(a) in allowed[int(t)] for a, t in zip(s2, t2)) and (t1 != t2).all()
    assert {int(x) for x in s1} | {int(x) for x in s2} == {1, 3, 4, 5, 6}


def test_group_and_atom_share_the_first_atom_key_in_reference_mode():
    """The reference keys adsorbates by the symbol of their first atom: an "HO" group next to an "O" atom extends the O run."""
    ens, _ = _group_ensemble(1, reference_groupby=True)
    SR, O, HO, H2O, E = 0, 1, 2, 3, ens.n_ads
    sp = np.array([[O, HO, SR, H2O, E, E, E]], np.int16)
    order = np.where(sp != E, np.cumsum(sp != E, axis=1), 0)
    cand = ens.switch_candidates(mc.ChainState(sp, order.astype(np.int64), np.array([5])))[0]
    assert np.flatnonzero(cand[O]).tolist() == [3]          # runs of key "O": [0, 1] then [3]: the last one survives
    assert np.flatnonzero(cand[SR]).tolist() == [2] and not cand[HO].any() and not cand[H2O].any()


def test_canonical_exchange_conserves_composition_with_shared_first_atom_keys():
    """("HO", "O") share the candidate key "O" in reference mode and propose_switch names the key's first adsorbate: the
    exchange must move what actually sits on the two sites -- a plain O never becomes an HO group, the counts of every
    adsorbate stay what they were (advisor finding, round 2)."""
    Z = structures.ATOMIC_NUMBERS
    base = structures.Structure(np.array([Z["Ti"], Z["Ti"]], np.int32), np.array([[0, 0, 0], [2.0, 0, 0]], float),
                                np.diag([20.0, 20.0, 20.0]), np.array([True, True, False]))
    coords = np.array([[1.5 * s, 0.0, 2.0] for s in range(7)], float)
    calc = LatticeGasCalc(2, {Z["Sr"]: -0.05, Z["O"]: 0.02, Z["H"]: 0.01}, J=0.0)
    n = 64
    ens = mc.ChainEnsemble(base, coords, ("HO", "O", "Sr"), n, calc, seed=3, relax=False, temperature=50.0,
                           reference_groupby=True)
    HO, O, SR, E = 0, 1, 2, ens.n_ads
    sp = np.tile(np.array([[SR, O, E, SR, E, HO, O]], np.int16), (n, 1))
    order = np.where(sp != E, np.cumsum(sp != E, axis=1), 0).astype(np.int64)
    ens.state = mc.ChainState(sp, order, np.full(n, 6, np.int64))
    ens.initialize()
    counts0 = np.stack([(ens.state.species == c).sum(axis=1) for c in range(ens.n_ads + 1)], axis=1)
    atoms0 = ens.num_adsorbate_atoms()
    moved = 0
    for _ in range(12):
        acc = ens.step_canonical()
        moved += int(acc.sum())
        counts = np.stack([(ens.state.species == c).sum(axis=1) for c in range(ens.n_ads + 1)], axis=1)
        assert np.array_equal(counts, counts0)
        assert np.array_equal(ens.num_adsorbate_atoms(), atoms0)
    assert moved > n                                                     # T = 50: nearly every exchange is accepted


def test_exchange_by_group_key_reproduces_the_reference_to_the_letter():
    """``exchange_by_group_key=True``: the reference's Exchange event hands the GROUP KEYS to ``change_site``
    (``mcmc/slab.py:168-232`` -> ``mcmc/events/event.py:138-151``), so an exchanged "HO" arrives as the plain atom "O": hydrogens
    are lost, the number of adsorbates is not.  The default (above) conserves the composition instead -- the documented
    divergence (advisor r3)."""
    Z = structures.ATOMIC_NUMBERS
    base = structures.Structure(np.array([Z["Ti"], Z["Ti"]], np.int32), np.array([[0, 0, 0], [2.0, 0, 0]], float),
                                np.diag([20.0, 20.0, 20.0]), np.array([True, True, False]))
    coords = np.array([[1.5 * s, 0.0, 2.0] for s in range(7)], float)
    calc = LatticeGasCalc(2, {Z["Sr"]: -0.05, Z["O"]: 0.02, Z["H"]: 0.01}, J=0.0)
    n = 64
    ens = mc.ChainEnsemble(base, coords, ("HO", "O", "Sr"), n, calc, seed=3, relax=False, temperature=50.0,
                           reference_groupby=True, exchange_by_group_key=True)
    HO, O, SR, E = 0, 1, 2, ens.n_ads
    assert ens.key_adsorbate.tolist() == [O, O, SR, E]                    # "HO" and "O" both arrive as "O"
    sp = np.tile(np.array([[SR, O, E, SR, E, HO, O]], np.int16), (n, 1))
    order = np.where(sp != E, np.cumsum(sp != E, axis=1), 0).astype(np.int64)
    ens.state = mc.ChainState(sp, order, np.full(n, 6, np.int64))
    ens.initialize()
    filled0 = ens.num_adsorbates().copy()
    for _ in range(12):
        ens.step_canonical()
        assert np.array_equal(ens.num_adsorbates(), filled0)              # sites stay filled ...
    lost = (ens.state.species == HO).sum(axis=1) == 0
    assert lost.any()                                                     # ... but groups have turned into plain atoms
    with pytest.raises(ValueError):                                       # a key outside the adsorbate list cannot be represented
        mc.ChainEnsemble(base, coords, ("HO", "Sr"), 2, calc, relax=False, reference_groupby=True, exchange_by_group_key=True)


def test_boltzmann_and_distance_decay_weights_match_the_reference_numbers():
    """``compute_boltzmann_weights`` on the reference's fixture (tests/test_slab.py:90-113): per-atom energies
    [1.0, 0.5, 1.0, 0.6], T = 1 -> As 0.1850956, Ga 0.30517106, empty 1; ``compute_distance_weight_matrix``
    (mcmc/utils/misc.py:170-190): rows sum to 1 and decay with distance."""
    bw = mc.boltzmann_atom_weights([1.0, 0.5, 1.0, 0.6], 1.0)
    assert np.allclose(bw[[1, 2]], [0.1850956, 0.30517106])
    D = mc.compute_distance_weight_matrix([(0, 0, 3), (1, 1, 1), (2, 2, 5)], 1.0)
    assert np.allclose(D.sum(axis=1), 1.0) and D[0, 0] > D[0, 1] > D[0, 2]
    # site weights inside the ensemble: filled sites carry the weight of their adsorbate's first atom, empty sites 1
    Z = structures.ATOMIC_NUMBERS
    base = structures.Structure(np.array([Z["Ga"]], np.int32), np.zeros((1, 3)), np.diag([20.0] * 3), np.array([True] * 3))
    ens = mc.ChainEnsemble(base, [(0, 0, 3), (1, 1, 1), (2, 2, 5)], ("As", "Ga"), 1, LatticeGasCalc(1, {}), relax=False,
                           temperature=1.0, require_per_atom_energies=True)
    st = mc.ChainState(np.array([[0, 1, 2]], np.int16), np.array([[1, 2, 0]]), np.array([3]))
    ens.per_atom_energies = [np.array([1.0, 0.5, 1.0])]     # slab atom, As, Ga
    w = ens._site_weights(ens.switch_candidates(st), st)[0]
    ref = mc.boltzmann_atom_weights([1.0, 0.5, 1.0], 1.0)
    assert np.allclose(w, [ref[1], ref[2], 1.0])
    ens.per_atom_energies = [None]
    with pytest.raises(ValueError):
        ens._site_weights(ens.switch_candidates(st), st)


def test_weighted_switch_proposals_follow_their_weights():
    """Two candidate sites of a species with Boltzmann weights 3 : 1 are proposed 3 : 1; with distance decay the partner
    site concentrates near the first site."""
    Z = structures.ATOMIC_NUMBERS
    base = structures.Structure(np.array([Z["Ti"]], np.int32), np.zeros((1, 3)), np.diag([30.0] * 3), np.array([True] * 3))
    coords = [(0.0, 0, 2), (1.0, 0, 2), (2.0, 0, 2), (9.0, 0, 2)]
    n = 6000
    ens = mc.ChainEnsemble(base, coords, ("Sr",), n, LatticeGasCalc(1, {}), relax=False, temperature=1.0, seed=4,
                           require_per_atom_energies=True, require_distance_decay=True, distance_decay_factor=1.0)
    sp = np.tile(np.array([[0, 1, 1, 0]], np.int16), (n, 1))          # Sr on sites 0 and 3, sites 1 and 2 empty
    st = mc.ChainState(sp, np.tile(np.array([[1, 0, 0, 2]]), (n, 1)), np.full(n, 3))
    e = np.array([0.0, np.log(3.0), 0.0])                               # atoms: slab, Sr@0 (weight 3), Sr@3 (weight 1)
    ens.per_atom_energies = [e] * n
    s1, s2, t1, t2, valid, _ = ens.propose_switch(1, st)
    assert valid.all()
    sr_first = t1 == 0
    frac0 = (s1[sr_first] == 0).mean()
    assert abs(frac0 - 0.75) < 0.03                                      # Boltzmann 3 : 1 between the two Sr sites
    # empty partner of Sr@0: sites 1 (distance 1) and 2 (distance 2): exp(-1) : exp(-2)
    part = s2[sr_first & (s1 == 0)]
    assert set(part.tolist()) <= {1, 2}
    assert abs((part == 1).mean() - np.exp(-1) / (np.exp(-1) + np.exp(-2))) < 0.04


# ---- packed-array fast path (no per-slab Python objects on the MC hot path) --------------------------------------------------
class PackedLatticeGasCalc(LatticeGasCalc):
    """The toy backend with the calculators' packed interface (``EnsembleNFFSurface.evaluate_packed``): same energies as the
    per-slab methods, computed from the packed arrays."""

    chem_pots: dict = {}
    offset_data: dict = {}

    def evaluate_packed(self, n_atoms, Z, pos, cell, pbc, relax=False, fixed_mask=None, relax_steps=20, fmax=0.01, optimizer=None):
        self.calls += 1
        self.packed_calls = getattr(self, "packed_calls", 0) + 1
        start = np.concatenate([[0], np.cumsum(n_atoms)]).astype(np.int64)
        e = np.array([self._energy(structures.Structure(Z[a0:a1], pos[a0:a1], np.reshape(cell[k], (3, 3))))
                      for k, (a0, a1) in enumerate(zip(start[:-1], start[1:]))], np.float64)
        return {"energy": e, "energy_std": np.zeros_like(e), "forces": np.zeros((len(Z), 3), np.float32),
                "energy_atoms": np.arange(len(Z), dtype=np.float32), "positions": np.array(pos, float), "cfg_start": start,
                "saturated": np.zeros(len(e), bool), "oob": np.zeros(len(e), bool),
                "n_steps": np.zeros(len(e), np.int32), "converged": np.ones(len(e), bool)}


def test_acceptance_energy_word_is_selectable():
    """``acceptance_energy="f64"`` (default) compares the fp64 word of the backend, ``"f32"`` the float32 result word the
    reference's Metropolis test sees (mcmc/calculators/calculators.py:484)."""
    class TwoWordCalc(PackedLatticeGasCalc):
        def evaluate_packed(self, *a, **k):
            out = super().evaluate_packed(*a, **k)
            out["energy_f64"] = out["energy"] + 1000.0      # (distinguishable words)
            out["energy"] = out["energy"].astype(np.float32)
            return out

    Z = structures.ATOMIC_NUMBERS
    base = structures.Structure(np.array([Z["Ti"], Z["Ti"]], np.int32), np.array([[0, 0, 0], [2.0, 0, 0]], float),
                                np.diag([20.0, 20.0, 20.0]), np.array([True, True, False]))
    coords = np.array([[1.0 * s, 0.0, 2.0] for s in range(4)], float)
    got = {}
    for word in ("f64", "f32"):
        calc = TwoWordCalc(2, {Z["Sr"]: -0.05, Z["O"]: 0.02}, J=0.03)
        ens = mc.ChainEnsemble(base, coords, ("Sr", "O"), 8, calc, seed=1, relax=False, acceptance_energy=word)
        ens.initialize()
        got[word] = ens.state.energy.copy()
    assert np.all(got["f64"] - got["f32"] > 999.0)
    with pytest.raises(ValueError):
        mc.ChainEnsemble(base, coords, ("Sr", "O"), 8, PackedLatticeGasCalc(2, {}, J=0.0), acceptance_energy="f16")


def test_batch_arrays_equal_the_per_chain_structures():
    """``batch_arrays`` builds what ``structure(b)`` builds for every chain -- atoms, groups ("HO", "H2O"), adsorption order."""
    ens, _ = _group_ensemble(48)
    state = ens.state
    for s in range(1, 30):
        site, end, _, _ = ens.propose(s, state)
        state = ens.apply(state, site, end)
    for which in (None, np.array([5, 0, 17, 40])):
        n_atoms, numbers, positions, ads_chain, z_ads = ens.batch_arrays(state, which)
        idx = np.arange(48) if which is None else which
        start = np.concatenate([[0], np.cumsum(n_atoms)])
        for k, b in enumerate(idx):
            ref = ens.structure(int(b), state)
            assert np.array_equal(numbers[start[k]:start[k + 1]], ref.numbers)
            assert np.array_equal(positions[start[k]:start[k + 1]], ref.positions)
            assert np.array_equal(np.sort(z_ads[ads_chain == k]), np.sort(ref.numbers[len(ens.base):]))
    assert ens.num_adsorbates(state).max() > 3 and (ens.num_adsorbate_atoms(state) > ens.num_adsorbates(state)).any()


def test_surface_energy_from_counts_equals_the_scalar_function_bit_for_bit(golden):
    from surface_sampling_amd import calculators as calcs

    rng = np.random.default_rng(5)
    chem = {"Sr": -2.0, "Ti": 0.3, "O": -0.7}
    base = ["Sr"] * 48 + ["Ti"] * 48 + ["O"] * 144
    energies, counts, want = [], {"Sr": [], "Ti": [], "O": []}, []
    for _ in range(64):
        extra = list(rng.choice(["Sr", "Ti", "O"], size=int(rng.integers(0, 30))))
        sym = base + extra
        e = float(np.float32(-1800.0 + rng.normal() * 40))
        energies.append(e)
        for k in counts:
            counts[k].append(sym.count(k))
        want.append(calcs.surface_energy_from_energy(e, sym, chem, golden.offset_data, "atomic"))
    got = calcs.surface_energy_from_counts(np.array(energies), {k: np.array(v) for k, v in counts.items()}, chem,
                                           golden.offset_data, "atomic")
    assert np.array_equal(got, np.array(want))


def test_packed_fast_path_gives_the_trajectories_of_the_per_slab_path():
    """Same seed, same toy energies: the packed path (arrays in, arrays out, lazy slabs) and the per-slab path accept the
    same proposals and end in the same states; the relaxed slabs are built only when somebody looks."""
    Z = structures.ATOMIC_NUMBERS
    base = structures.Structure(np.array([Z["Ti"], Z["Ti"]], np.int32), np.array([[0, 0, 0], [2.0, 0, 0]], float),
                                np.diag([20.0, 20.0, 20.0]), np.array([True, True, False]))
    coords = np.array([[1.0 * s, 0.0, 2.0] for s in range(6)], float)
    runs = []
    for fast in (True, False):
        for relax in (False, True):
            calc = PackedLatticeGasCalc(2, {Z["Sr"]: -0.05, Z["O"]: 0.02, Z["H"]: 0.01}, J=0.03)
            ens = mc.ChainEnsemble(base, coords, ("Sr", "O", "HO"), 32, calc, seed=4, relax=relax, temperature=0.05,
                                   fixed_indices=np.array([0]))
            ens.fast_path = fast
            acc = np.stack([ens.step_semigrand() for _ in range(8)] + [ens.step_canonical() for _ in range(4)])
            assert (getattr(calc, "packed_calls", 0) > 0) == fast
            runs.append((fast, relax, acc, ens.state.species.copy(), ens.state.energy.copy(), ens))
    for relax in (False, True):
        a = next(r for r in runs if r[0] and r[1] == relax)
        b = next(r for r in runs if not r[0] and r[1] == relax)
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
        assert a[2].any() and not a[2].all()
        fast_ens, slow_ens = a[5], b[5]
        assert isinstance(fast_ens.relaxed, mc.SlabRefs) and any(isinstance(fast_ens.relaxed.raw(k), tuple) for k in range(32))
        for k in range(32):
            assert np.array_equal(fast_ens.relaxed[k].numbers, slow_ens.relaxed[k].numbers)
            assert np.array_equal(fast_ens.relaxed[k].positions, slow_ens.relaxed[k].positions)
            # the stored slab is the slab of the KEPT state (the toy backend does not move atoms): a rejected chain must not
            # end up with the slab of its rejected proposal
            kept = fast_ens.structure(k)
            assert np.array_equal(fast_ens.relaxed[k].numbers, kept.numbers) and np.array_equal(fast_ens.relaxed[k].positions, kept.positions)
        assert not isinstance(fast_ens.relaxed.raw(0), tuple)        # looked at: now a Structure
    # a user-supplied energy function needs the structures: the per-slab path serves it
    calc = PackedLatticeGasCalc(2, {Z["Sr"]: -0.05, Z["O"]: 0.02}, J=0.0)
    ens = mc.ChainEnsemble(base, coords, ("Sr", "O"), 4, calc, relax=False, surface_energy_fn=lambda e, s: float(e) + len(s))
    ens.initialize()
    assert getattr(calc, "packed_calls", 0) == 0


# ---- DistanceCriterion / filter_distances (reference tests/test_filter_distance.py, tests/events/test_criterion.py:8-11) ---------
def _filter_fixture():
    import os

    F = np.load(os.path.join(os.path.dirname(__file__), "golden", "filter_distance.npz"))
    unit = structures.Structure(F["unit_numbers"], F["unit_positions"], F["unit_cell"], F["unit_pbc"])
    failed = structures.Structure(F["failed_numbers"], F["failed_positions"], F["failed_cell"], F["failed_pbc"])
    return unit.repeat((2, 2, 1)), failed, {k: F[k] for k in ("ase_bridge", "ase_top1", "ase_top2")}


def _with_adsorbed(slab, coords, element="O"):
    s = slab.copy()
    for x in coords:
        s.numbers = np.append(s.numbers, structures.ATOMIC_NUMBERS[element]).astype(np.int32)
        s.positions = np.vstack([s.positions, np.asarray(x, float)[None]])
    return s


def test_filter_distances_reference_cases():
    """The five cases of the reference's tests/test_filter_distance.py on its own fixtures (the 2 x 2 tiling of its unit-cell
    slab with O adsorbed at the bridge / top coordinates the test file defines; the 'distance failed' CIF)."""
    pristine, failed, c = _filter_fixture()
    assert not mc.filter_distances(_with_adsorbed(pristine, [c["ase_bridge"]]), ads=["O"], cutoff_distance=1.5)       # test_one_O_fail
    assert mc.filter_distances(_with_adsorbed(pristine, [c["ase_top1"]]), ads=["O"], cutoff_distance=1.5)             # test_one_O_pass
    assert not mc.filter_distances(_with_adsorbed(pristine, [c["ase_bridge"], c["ase_top1"]]), ads=["O"], cutoff_distance=1.5)   # two_O_fail
    assert mc.filter_distances(_with_adsorbed(pristine, [c["ase_top1"], c["ase_top2"]]), ads=["O"], cutoff_distance=1.5)         # two_O_pass
    assert not mc.filter_distances(failed, ads=["O"], cutoff_distance=1.5)                                             # cell_distance_failed
    # the reference's DistanceCriterion fixture (tests/events/test_fixtures.py:12-24): Ga atoms sqrt(3) apart pass at 1.5
    Z = structures.ATOMIC_NUMBERS
    gaas = structures.Structure(np.array([Z["Ga"], Z["As"], Z["Ga"], Z["As"]], np.int32),
                                np.array([[0, 0, 0], [0, 0, 3], [1, 1, 1], [1, 1, 4]], float), np.eye(3) * 30, np.array([False] * 3))
    assert mc.filter_distances(gaas, ["Ga"], 1.5) is True and mc.filter_distances(gaas, ["Ga"], 1.8) is False
    # minimum image in a skewed (hexagonal) cell: two atoms 0.9 A apart across the periodic boundary
    hexc = np.array([[3.0, 0, 0], [-1.5, 1.5 * np.sqrt(3.0), 0], [0, 0, 20.0]])
    a, b = np.array([0.05, 0.05, 5.0]) @ hexc, (np.array([0.95, 0.95, 5.0]) @ hexc)
    d = mc.mic_distance_matrix(a[None], b[None], hexc, [True, True, False])[0, 0]
    brute = min(np.linalg.norm(b - a + i * hexc[0] + j * hexc[1]) for i in (-2, -1, 0, 1, 2) for j in (-2, -1, 0, 1, 2))
    assert d == pytest.approx(brute, abs=1e-12) and d < 1.0


def test_distance_and_testing_criteria_replace_the_energy_test():
    """``filter_distance > 0`` selects the reference's DistanceCriterion (mcmc/mcmc.py:218-227,253-262): acceptance = the
    proposed unrelaxed slab passes ``filter_distances`` for the criterion's atom types, no energy enters, state energies are
    brought up to date once per sweep; ``testing=True`` accepts everything."""
    Z = structures.ATOMIC_NUMBERS
    base = structures.Structure(np.array([Z["Ti"], Z["Ti"]], np.int32), np.array([[0, 0, 0], [2.0, 0, 0]], float),
                                np.diag([20.0, 20.0, 20.0]), np.array([True, True, False]))
    coords = np.array([[1.0 * s, 0.0, 2.0] for s in range(6)], float)       # sites 1 A apart: neighbouring Sr adsorbates collide
    calc = PackedLatticeGasCalc(2, {Z["Sr"]: -0.05, Z["O"]: 0.02}, J=0.03)
    ens = mc.ChainEnsemble(base, coords, ("Sr", "O"), 64, calc, seed=2, relax=False, temperature=0.05, filter_distance=1.5)
    assert ens.criterion == "distance"
    n_acc = np.zeros(64, int)
    for _ in range(30):
        before = ens.state.copy()
        acc = ens.step_semigrand()
        n_acc += acc
        for b in range(64):
            s = ens.structure(b)
            assert mc.filter_distances(s, ("Sr", "Ti"), 1.5)                  # every kept state passes the filter ...
            if not acc[b]:
                assert np.array_equal(ens.state.species[b], before.species[b])
    assert calc.calls == 0                                                    # ... and no energy was asked for
    assert n_acc.min() < 30 and n_acc.sum() > 0                               # some proposals collide, some do not
    SR = 0
    sr = ens.state.species == SR
    assert not (sr[:, 1:] & sr[:, :-1]).any()                                 # never two Sr on neighbouring sites (1 A apart)
    out = ens.sweep(0, sweep_size=5)
    assert calc.calls >= 1 and not ens.energy_stale.any() and np.isfinite(out["energy"]).all()
    for b in range(64):
        assert out["energy"][b] == pytest.approx(calc._energy(ens.structure(b)), abs=1e-12)
    # O is not among the criterion's default types: O adsorbates may sit on neighbouring sites
    o = ens.state.species == 1
    assert (o[:, 1:] & o[:, :-1]).any()
    # exchange moves under the same criterion
    acc = np.stack([ens.step_canonical() for _ in range(10)])
    assert acc.any()
    for b in range(64):
        assert mc.filter_distances(ens.structure(b), ("Sr", "Ti"), 1.5)
    # testing criterion: always accept
    ens2 = mc.ChainEnsemble(base, coords, ("Sr", "O"), 16, calc, seed=2, relax=False, testing=True)
    assert ens2.criterion == "testing" and all(ens2.step_semigrand().all() for _ in range(5))


def test_prepare_canonical_and_fixed_site_steps():
    """``MCMC.prepare_canonical`` (mcmc/mcmc.py:148-188): semigrand steps until every chain holds the requested number of
    adsorbates -- a chain that has reached it stops while the others go on -- or one step on each of the evenly spread sites
    (Ward clustering of the in-plane site coordinates, the site closest to each cluster centre); ``step_semigrand(site_idx=...)``
    changes the given site (``ChangeProposal(site_idx=...)``)."""
    Z = structures.ATOMIC_NUMBERS
    base = structures.Structure(np.array([Z["Ti"], Z["Ti"]], np.int32), np.array([[0, 0, 0], [2.0, 0, 0]], float),
                                np.diag([20.0, 20.0, 20.0]), np.array([True, True, False]))
    coords = np.array([[2.0 * i, 2.0 * j, 2.0] for i in range(4) for j in range(3)], float)
    calc = PackedLatticeGasCalc(2, {Z["Sr"]: -0.05, Z["O"]: 0.02}, J=0.0)
    ens = mc.ChainEnsemble(base, coords, ("Sr", "O"), 24, calc, seed=6, relax=False, temperature=5.0)
    ens.initialize()
    hist = []
    orig = ens.step_semigrand

    def spy(*a, **k):
        hist.append(None if k.get("which") is None else k["which"].copy())
        return orig(*a, **k)

    ens.step_semigrand = spy
    counts = ens.prepare_canonical(4)
    assert (counts >= 4).all() and (counts == 4).all()          # a change adds at most one adsorbate, finished chains are frozen
    assert len(hist) >= 4 and hist[-1].sum() < 24               # ... the last steps ran for the stragglers only
    frozen = ens.state.species.copy()
    ens.step_semigrand = orig
    acc = ens.step_semigrand(which=np.zeros(24, bool))
    assert not acc.any() and np.array_equal(ens.state.species, frozen)
    # fixed site: only that site may differ afterwards
    before = ens.state.species.copy()
    ens.step_semigrand(site_idx=7)
    changed = np.argwhere(ens.state.species != before)
    assert set(changed[:, 1].tolist()) <= {7}
    with pytest.raises(IndexError):
        ens.step_semigrand(site_idx=99)
    # even seeding: n well separated sites, one per cluster, the same for every chain
    ens2 = mc.ChainEnsemble(base, coords, ("Sr", "O"), 8, calc, seed=6, relax=False, testing=True)
    sites = ens2.even_adsorption_sites(4)
    assert len(set(sites.tolist())) == 4
    from scipy.cluster.hierarchy import fcluster, linkage
    labels = fcluster(linkage(coords[:, :2], "ward"), 4, criterion="maxclust")
    assert sorted(labels[sites].tolist()) == [1, 2, 3, 4]
    n = ens2.prepare_canonical(4, even_adsorption_sites=True)
    filled = ens2.state.species != ens2.n_ads
    assert (n == 4).all() and all(set(np.flatnonzero(f).tolist()) == set(sites.tolist()) for f in filled)   # testing criterion: every step accepted
    with pytest.raises(ValueError):
        ens2.prepare_canonical(0)


def test_slab_references_do_not_pin_an_unbounded_number_of_batches():
    """A chain that is never accepted keeps referencing the packed batch of an old step; beyond 16 distinct batches the
    referenced slabs are copied into one (``SlabRefs.consolidate``) -- the slabs themselves are unchanged."""
    Z = structures.ATOMIC_NUMBERS
    base = structures.Structure(np.array([Z["Ti"], Z["Ti"]], np.int32), np.array([[0, 0, 0], [2.0, 0, 0]], float),
                                np.diag([20.0, 20.0, 20.0]), np.array([True, True, False]))
    coords = np.array([[1.0 * s, 0.0, 2.0] for s in range(6)], float)
    calc = PackedLatticeGasCalc(2, {Z["Sr"]: -0.05, Z["O"]: 0.02}, J=0.03)
    ens = mc.ChainEnsemble(base, coords, ("Sr", "O"), 48, calc, seed=8, relax=False, temperature=0.02)
    ens.initialize()
    seen = 0
    for _ in range(60):
        ens.step_semigrand()
        n = len({id(it[0]) for it in ens.relaxed.items if isinstance(it, tuple)})
        seen = max(seen, n)
        assert n <= 17
    assert seen > 4                                                # (references to several steps do coexist)
    for b in range(48):
        kept = ens.structure(b)
        assert np.array_equal(ens.relaxed[b].numbers, kept.numbers) and np.array_equal(ens.relaxed[b].positions, kept.positions)


def test_concurrent_chain_groups_walk_the_trajectories_of_one_ensemble():
    """``ConcurrentChains``: 3 groups of chains (own calculators, own host threads) step through exactly what the same 10
    chains do inside one ensemble -- accept counts, energies, occupations, every sweep of the history."""
    Z = structures.ATOMIC_NUMBERS
    whole, _ = _toy(10)
    want = whole.run(total_sweeps=3, sweep_size=4, start_temp=0.1, alpha=0.5, keep_structures=True)
    calcs = [LatticeGasCalc(2, {Z["Sr"]: -0.05, Z["O"]: 0.02}, J=0.03) for _ in range(3)]
    groups = mc.ConcurrentChains.build(whole.base, whole.ads_coords, ("Sr", "O"), 10, calcs, seed=3, relax=False, temperature=0.05)
    assert [len(g.chain_ids) for g in groups.groups] == [3, 3, 4] and np.array_equal(groups.chain_ids, np.arange(10))
    got = groups.run(total_sweeps=3, sweep_size=4, start_temp=0.1, alpha=0.5, keep_structures=True)
    for k in ("energy_hist", "frac_accept_hist", "adsorption_count_hist"):
        assert len(got[k]) == 3
        for a, b in zip(got[k], want[k]):
            assert np.array_equal(a, b), k
    for a, b in zip(got["history"], want["history"]):
        assert np.array_equal(a.species, b.species) and np.array_equal(a.order, b.order) and np.array_equal(a.counter, b.counter)
    for a, b in zip(got["trajectories"], want["trajectories"]):
        assert len(a) == len(b) == 10
        assert all(np.array_equal(x.numbers, y.numbers) and np.array_equal(x.positions, y.positions) for x, y in zip(a, b))
    assert np.array_equal(groups.energy, whole.state.energy) and np.array_equal(groups.species, whole.state.species)
    assert groups.n_evaluations == whole.n_evaluations
    # steps(): the same again, continuing both
    acc = groups.steps(5)
    ref = np.zeros(10, np.int64)
    for _ in range(5):
        ref += whole.step_semigrand()
    assert np.array_equal(acc, ref) and np.array_equal(groups.energy, whole.state.energy)
    # guards: overlapping chains, a shared calculator; an exception inside a group's thread surfaces
    a, ca = _toy(4)
    b, cb = _toy(4, first_chain=2)
    with pytest.raises(ValueError, match="overlap"):
        mc.ConcurrentChains([a, b])
    c = mc.ChainEnsemble(a.base, a.ads_coords, ("Sr", "O"), 4, ca, seed=3, first_chain=10, relax=False)
    with pytest.raises(ValueError, match="own calculator"):
        mc.ConcurrentChains([a, c])

    class Broken(LatticeGasCalc):
        def calculate_batch(self, *args, **kw):
            raise RuntimeError("device lost")
    bad = mc.ConcurrentChains.build(whole.base, whole.ads_coords, ("Sr", "O"), 4,
                                    [LatticeGasCalc(2, {Z["Sr"]: -0.05}, J=0.0), Broken(2, {Z["Sr"]: -0.05}, J=0.0)],
                                    seed=3, relax=False)
    with pytest.raises(RuntimeError, match="device lost"):
        bad.initialize()


# ---- advisor findings, round 4 ----------------------------------------------------------------------------------------------
def test_explicit_empty_fixed_indices_hold_nothing_on_both_paths():
    """``fixed_indices=[]`` means "hold nothing" on the packed path as on the per-slab path (a mask of zeros reaches the
    calculator); only ``None`` means "the calculator's default group" (``fixed_mask=None``)."""
    Z = structures.ATOMIC_NUMBERS
    base = structures.Structure(np.array([Z["Ti"], Z["Ti"]], np.int32), np.array([[0, 0, 0], [2.0, 0, 0]], float),
                                np.diag([20.0, 20.0, 20.0]), np.array([True, True, False]))
    coords = np.array([[1.0 * s, 0.0, 2.0] for s in range(6)], float)
    seen = {}

    class Spy(PackedLatticeGasCalc):
        def evaluate_packed(self, *a, fixed_mask=None, **k):
            seen["packed"] = None if fixed_mask is None else np.array(fixed_mask, copy=True)
            return super().evaluate_packed(*a, fixed_mask=fixed_mask, **k)

        def relax_batch(self, slabs, fixed_indices=None, relax_steps=20, fmax=0.01):
            seen["slab"] = fixed_indices
            return super().relax_batch(slabs, fixed_indices, relax_steps, fmax)

    for fixed, packed_is_none in ((np.array([], np.int64), False), (None, True), (np.array([1]), False)):
        for fast in (True, False):
            calc = Spy(2, {Z["Sr"]: -0.05, Z["O"]: 0.02}, J=0.03)
            ens = mc.ChainEnsemble(base, coords, ("Sr", "O"), 5, calc, seed=4, relax=True, temperature=0.05, fixed_indices=fixed)
            ens.fast_path = fast
            ens.initialize()
            if fast:
                assert (seen["packed"] is None) == packed_is_none
                if fixed is not None:
                    assert seen["packed"].dtype == np.uint8 and int(seen["packed"].sum()) == 5 * len(fixed)
            else:
                assert (seen["slab"] is None) == (fixed is None)
                if fixed is not None:
                    assert all(len(f) == len(fixed) for f in seen["slab"])


def test_steps_after_prepare_canonical_do_not_depend_on_the_other_chains():
    """The random numbers are keyed by (seed, global chain id, step): after ``prepare_canonical`` every ensemble continues from
    the SAME step index whatever its slowest chain needed, so chains prepared together walk exactly the trajectories they
    walk when prepared in two separate ensembles (grouping / sharding independence holds through the preparation)."""
    Z = structures.ATOMIC_NUMBERS
    base = structures.Structure(np.array([Z["Ti"], Z["Ti"]], np.int32), np.array([[0, 0, 0], [2.0, 0, 0]], float),
                                np.diag([20.0, 20.0, 20.0]), np.array([True, True, False]))
    coords = np.array([[2.0 * i, 2.0 * j, 2.0] for i in range(4) for j in range(3)], float)

    def run(n, first):
        calc = PackedLatticeGasCalc(2, {Z["Sr"]: -0.05, Z["O"]: 0.02}, J=0.0)
        ens = mc.ChainEnsemble(base, coords, ("Sr", "O"), n, calc, seed=6, first_chain=first, relax=False, temperature=5.0)
        ens.initialize()
        counts = ens.prepare_canonical(4, max_steps=500)
        prepared = ens.state.species.copy()
        after_prepare = ens.step_count
        acc = np.stack([ens.step_canonical() for _ in range(6)] + [ens.step_semigrand() for _ in range(3)])
        return counts, prepared, after_prepare, acc, ens.state.species.copy(), ens.state.energy.copy()

    whole = run(24, 0)
    parts = [run(10, 0), run(14, 10)]
    assert whole[2] == parts[0][2] == parts[1][2] == 500           # a fixed continuation index
    assert np.array_equal(whole[1], np.concatenate([p[1] for p in parts]))
    assert np.array_equal(whole[3], np.concatenate([p[3] for p in parts], axis=1)) and whole[3].any()
    assert np.array_equal(whole[4], np.concatenate([p[4] for p in parts]))
    assert np.array_equal(whole[5], np.concatenate([p[5] for p in parts]))


def test_minimum_image_distances_in_skewed_and_degenerate_cells():
    """``mic_distance_matrix`` = brute force over many images for strongly skewed cells (ASE gets there by a Minkowski
    reduction), and a zero-length vector on a non-periodic axis is completed instead of raising (``ase.geometry.complete_cell``)."""
    rng = np.random.default_rng(0)
    for trial in range(12):
        cell = np.array([[4, 0, 0], [rng.uniform(-9, 9), 3, 0], [rng.uniform(-3, 3), rng.uniform(-3, 3), 10]])
        pbc = [True, True, trial % 2 == 0]
        xa, xb = rng.uniform(0, 4, (5, 3)), rng.uniform(0, 4, (6, 3))
        got = mc.mic_distance_matrix(xa, xb, cell, pbc)
        best = np.full((5, 6), np.inf)
        span = range(-14, 15)
        for i in span:
            for j in span:
                for k in (span if pbc[2] else [0]):
                    dd = xb[None] - xa[:, None] + i * cell[0] + j * cell[1] + k * cell[2]
                    best = np.minimum(best, np.sqrt((dd * dd).sum(2)))
        assert np.allclose(got, best, atol=1e-12)
    flat = np.array([[4.0, 0, 0], [1.0, 3.0, 0], [0, 0, 0]])
    d = mc.mic_distance_matrix(np.zeros((1, 3)), np.array([[3.9, 0.0, 1.0]]), flat, [True, True, False])
    assert d[0, 0] == pytest.approx(np.hypot(0.1, 1.0))
    assert np.allclose(mc.complete_cell(flat)[2], [0, 0, 1])
    assert np.allclose(mc.complete_cell(np.zeros((3, 3))), np.eye(3))
    one = mc.complete_cell(np.array([[0, 0, 0], [0, 2.0, 0], [0, 0, 0]]))
    assert abs(np.linalg.det(one)) == pytest.approx(2.0)
