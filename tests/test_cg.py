"""LAMMPS-style conjugate-gradient minimisation (reference ``optimizer: "LAMMPS"`` for GaN, ``min_style cg``):
the numpy restatement on CPU, the device state machine against it on the GPU."""

import os

import numpy as np
import pytest

from cg_oracle import cg_minimize
from conftest import GOLDEN


def _gan(golden, sigma=0.05, seed=4):
    from surface_sampling_amd.structures import Structure

    g = golden.structure("GaN_3x3_pristine")
    rng = np.random.default_rng(seed)
    types = np.array([0 if z == 31 else 1 for z in g.numbers], np.int32)
    return Structure(g.numbers, g.positions + rng.normal(0, sigma, g.positions.shape), g.cell, g.pbc), types


def test_cg_restatement_minimises_tersoff(golden, oracle_mod):
    """Invariants of the restatement on the rattled GaN slab (bulk atoms held like the reference's `fix 2 bulk setforce 0`):
    accepted energies decrease monotonically, it stops on the energy tolerance within the reference's 100 iterations, held
    atoms do not move, and the minimum agrees with a long BFGS relaxation of the same start."""
    from bfgs_oracle import bfgs_relax

    s, types = _gan(golden)
    fixed = np.arange(12, 36)

    def fn(pos):
        E, _, F = oracle_mod.tersoff(golden.tersoff_params, types, pos, s.cell, [1, 1, 1])
        return E, F

    pos, e, niter, neval, why, trace = cg_minimize(fn, s.positions, fixed=fixed, max_iter=100)
    assert why == 1 and 2 <= niter <= 100 and neval >= niter
    assert all(b <= a + 1e-12 for a, b in zip(trace, trace[1:])) and trace[-1] < trace[0] - 0.1
    assert np.array_equal(pos[fixed], s.positions[fixed])
    pos2, tr2, _, _ = bfgs_relax(fn, s.positions, fixed=fixed, max_steps=200, fmax=1e-4)
    assert 0.0 <= e - tr2[-1][0] < 2e-2      # etol 1e-5 relative of ~143 eV: stops when an iteration gains < 1.4 meV, ~8 meV above the minimum
    # tight tolerances reach the same minimum to 1e-7 eV
    _, e_tight, _, _, why_t, _ = cg_minimize(fn, s.positions, fixed=fixed, max_iter=2000, etol=0.0, ftol=1e-6)
    assert why_t == 2 and abs(e_tight - tr2[-1][0]) < 1e-6


def test_cg_restatement_stop_conditions():
    K = np.diag([1.0, 4.0, 9.0, 1.0, 2.0, 3.0])

    def fn(pos):
        x = pos.reshape(-1)
        return 0.5 * x @ K @ x, -(K @ x).reshape(-1, 3)

    x0 = np.array([[0.3, -0.2, 0.1], [0.05, 0.4, -0.3]])
    pos, e, niter, neval, why, _ = cg_minimize(fn, x0, etol=0.0, ftol=1e-10, max_iter=200)
    assert why == 2 and np.abs(pos).max() < 1e-9
    assert cg_minimize(fn, x0, max_iter=1)[4] == 3
    assert cg_minimize(fn, x0, max_eval=3, etol=0.0, ftol=0.0)[4] == 4
    assert cg_minimize(fn, np.zeros((2, 3)))[4] in (5, 6)      # zero force: nothing to follow


@pytest.mark.gpu
def test_cg_gpu_follows_the_restatement(golden, oracle_mod):
    """Device CG (vssr_batch_relax_cg) on a batch of three chains -- two rattled GaN slabs (different starts, different
    iteration counts) on Tersoff and the Cu(100) slab with two adatoms on EAM in a second handle -- against the restatement
    driven by the fp64 oracles: same iteration / evaluation counts and stop reasons, energies to 1e-9 eV."""
    import eam_oracle
    from surface_sampling_amd import backend, eam, structures

    fixed = np.arange(12, 36)
    cases = [_gan(golden, 0.05, 4), _gan(golden, 0.02, 9)]
    packs = [(t, s.positions, s.cell, np.ones(3, np.uint8)) for s, t in cases]
    mask = np.zeros(72, np.uint8)
    mask[fixed] = 1
    mask[36 + fixed] = 1
    eng = backend.TersoffEngine(golden.tersoff_params, device=0)
    e, ea, f, pos, it, ev, why = eng.relax_cg_f64(packs, fixed=mask, max_iter=100)
    for b, (s, types) in enumerate(cases):
        def fn(p, types=types, s=s):
            E, _, F = oracle_mod.tersoff(golden.tersoff_params, types, p, s.cell, [1, 1, 1])
            return E, F

        pref, eref, niter, neval, reason, _ = cg_minimize(fn, s.positions, fixed=fixed, max_iter=100)
        assert (it[b], ev[b], why[b]) == (niter, neval, reason), (b, it[b], ev[b], why[b], niter, neval, reason)
        assert abs(e[b] - eref) < 1e-9 and np.abs(pos[36 * b:36 * b + 36] - pref).max() < 1e-9
        assert np.array_equal(pos[36 * b:36 * b + 36][fixed], s.positions[fixed])
    assert it[0] != it[1]                                         # the chains really ran different numbers of iterations
    eng.close()
    # EAM: Cu(100) + two adatoms, bottom layer held
    d = np.load(os.path.join(GOLDEN, "cu100.npz"))
    fl = eam.read_funcfl(os.path.join(GOLDEN, "Cu_u3.eam"))
    pos0 = np.vstack([d["positions"], d["ads_coords"][[5, 12]]])
    e2 = backend.EAMEngine(fl, device=0)
    out = e2.relax_cg_f64([(np.zeros(10, np.int32), pos0, d["cell"], d["pbc"].astype(np.uint8))], fixed=np.array([1] * 4 + [0] * 6, np.uint8),
                          max_iter=60)

    def fn2(p):
        E, _, F = eam_oracle.eam(fl, p, d["cell"], d["pbc"])
        return E, F

    pref, eref, niter, neval, reason, _ = cg_minimize(fn2, pos0, fixed=np.arange(4), max_iter=60)
    assert (out[4][0], out[5][0], out[6][0]) == (niter, neval, reason)
    assert abs(out[0][0] - eref) < 1e-9 and eref < eam_oracle.eam(fl, pos0, d["cell"], d["pbc"])[0] - 0.5
    e2.close()


@pytest.mark.gpu
def test_live_chain_compaction_changes_nothing_but_the_work(golden, monkeypatch):
    """96 GaN chains (36 + 12 atoms, different rattles: 30 .. 150 evaluations each) minimised in lock step with and without the
    live-chain compaction of the resident batch (``VSSR_RELAX_COMPACT``): positions, energies, per-atom energies, forces,
    iteration / evaluation counts and stop reasons identical bit for bit; with it the driver dispatches far fewer
    chain-evaluations for the same number of lock-step launches (``vssr_batch_relax_counts``).  A second relaxation and a plain
    evaluation on the same handle afterwards see the original batch again."""
    from surface_sampling_amd import backend

    packs, mask = _gan_mc_like_batch(golden, 96, 12)
    eng = backend.TersoffEngine(golden.tersoff_params, device=0)
    monkeypatch.setenv("VSSR_CG_FUSED", "0")     # the lock-step driver (chains of this size take the chain-resident minimiser by default)
    runs = {}
    for flag in ("0", "2"):      # off / for every batch size (the default compacts resident batches of >= 65 536 atoms only)
        monkeypatch.setenv("VSSR_RELAX_COMPACT", flag)
        out = eng.relax_cg_f64(packs, fixed=mask, max_iter=100)
        runs[flag] = (out, eng.last_relax_counts)
    (a, ca), (b, cb) = runs["0"], runs["2"]
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    ev = a[5]
    assert ev.min() < 0.6 * ev.max()                               # the chains really stop at different times
    assert ca[0] == cb[0] and ca[1] == ca[0] * 96                    # same lock-step launches; without compaction every launch is full
    needed = int(ev.sum()) + 96
    assert needed <= cb[1] < 0.8 * ca[1], (needed, cb, ca)         # ... with it the dispatched chain-evaluations follow the live set
    assert np.array_equal(a[3][mask.astype(bool)], np.concatenate([p[1] for p in packs])[mask.astype(bool)])   # held atoms did not move
    # the handle holds the original batch again: a static evaluation of the relaxed geometries equals the returned results
    n_atoms, T, pos, cell, pbc = backend.pack_batch(packs)
    e2, ea2, f2 = eng.evaluate_arrays_f64(n_atoms, T, b[3], cell, pbc)
    assert np.array_equal(e2, b[0]) and np.array_equal(f2, b[2])
    eng.close()


def _gan_mc_like_batch(golden, n_chains, seed):
    """GaN 3 x 3 slabs + 12 adatoms each, different rattles (chains need 20 .. 150 evaluations); bulk layers held."""
    g = golden.structure("GaN_3x3_pristine")
    ztop = g.positions[:, 2].max()
    rng = np.random.default_rng(seed)
    types36 = np.array([0 if z == 31 else 1 for z in g.numbers], np.int32)
    packs, mask = [], []
    for b in range(n_chains):
        ads = np.array([(rng.uniform(), rng.uniform(), 0.0) for _ in range(12)]) @ g.cell
        ads[:, 2] = ztop + rng.uniform(1.6, 2.4, 12)
        pos = np.vstack([g.positions + rng.normal(0, 0.01 + 0.05 * rng.uniform(), g.positions.shape), ads])
        packs.append((np.concatenate([types36, np.zeros(12, np.int32)]), pos, g.cell, np.ones(3, np.uint8)))
        m = np.zeros(48, np.uint8)
        m[:36][g.positions[:, 2] < ztop - 3.0] = 1
        mask.append(m)
    return packs, np.concatenate(mask)


@pytest.mark.gpu
def test_chain_resident_minimiser_equals_the_lock_step_driver(golden, monkeypatch):
    """``chain_min.hip`` (one workgroup minimises one chain from its first evaluation to its stop criterion; the default for
    Tersoff chains of <= 256 atoms) against the lock-step driver of ``relax.hip`` (``VSSR_CG_FUSED=0``) on 80 GaN chains of 48
    atoms and a ragged batch (36 / 48 / 96-atom chains, a chain without held atoms): minimised positions, energies, per-atom
    energies, forces, iteration / evaluation counts and stop reasons identical BIT FOR BIT; one launch instead of ~150 lock-step
    evaluations, and exactly the chain-evaluations the chains need.  Also with a neighbor capacity that is too small at first
    (the pools are enlarged and the chains continue where they stopped) and with the one-thread site kernel."""
    from surface_sampling_amd import backend

    packs, mask = _gan_mc_like_batch(golden, 80, 21)
    g = golden.structure("GaN_3x3_pristine")
    big = g.repeat((2, 1, 1))                                                # 72 atoms + 24 adatoms: two site tiles
    rng = np.random.default_rng(3)
    tb = np.array([0 if z == 31 else 1 for z in big.numbers], np.int32)
    ads = np.array([(rng.uniform(), rng.uniform(), 0.0) for _ in range(24)]) @ big.cell
    ads[:, 2] = big.positions[:, 2].max() + rng.uniform(1.6, 2.4, 24)
    ragged = [packs[0], (tb[:36], g.positions + rng.normal(0, 0.04, g.positions.shape), g.cell, np.ones(3, np.uint8)),
              (np.concatenate([tb, np.zeros(24, np.int32)]), np.vstack([big.positions + rng.normal(0, 0.03, big.positions.shape), ads]),
               big.cell, np.ones(3, np.uint8)), packs[5]]
    rmask = np.concatenate([mask[:48], np.zeros(36, np.uint8), np.zeros(96, np.uint8), mask[5 * 48:6 * 48]])
    eng = backend.TersoffEngine(golden.tersoff_params, device=0)
    for batch, fx in ((packs, mask), (ragged, rmask)):
        out = {}
        for flag in ("0", "1"):      # lock-step driver / chain-resident minimiser
            monkeypatch.setenv("VSSR_CG_FUSED", flag)
            out[flag] = (eng.relax_cg_f64(batch, fixed=fx, max_iter=100), eng.last_relax_counts)
        (a, ca), (b, cb) = out["0"], out["1"]
        for k, (x, y) in enumerate(zip(a, b)):
            assert np.array_equal(x, y), k
        assert cb[0] == 1 and ca[0] > 20                                       # one launch against the lock-step evaluations
        assert cb[1] == int(b[5].sum()) + len(batch) and cb[1] < ca[1]           # the chains' own evaluations (+ the setup one), nothing else
        assert len(set(b[5].tolist())) > 2 and (b[6] > 0).all()
    # slot pools too small at first: enlarged, the chains resume; same results
    ref = out["1"][0]
    eng.debug_capacity(slots_per_atom=4)
    again = eng.relax_cg_f64(ragged, fixed=rmask, max_iter=100)
    assert eng.debug_capacity() >= 1                                          # (regrows of the last relaxation)
    for x, y in zip(ref, again):
        assert np.array_equal(x, y)
    # crowded chains (adatoms pushed into the slab: rows of more than 16 slots): the one-thread site path inside the kernel
    monkeypatch.setenv("VSSR_CG_FUSED", "1")
    crowded = [(t.copy(), p.copy(), c, b_) for t, p, c, b_ in packs[:8]]
    for k, (t, p, c, b_) in enumerate(crowded):
        p[36:40] = p[20] + np.array([[0.9, 0.0, 0.3], [-0.9, 0.2, 0.4], [0.1, 0.95, -0.2], [0.2, -0.9, 0.5]]) * (1.0 + 0.05 * k)
    got = eng.relax_cg_f64(crowded, fixed=mask[:8 * 48], max_iter=30)
    monkeypatch.setenv("VSSR_CG_FUSED", "0")
    want = eng.relax_cg_f64(crowded, fixed=mask[:8 * 48], max_iter=30)
    for x, y in zip(want, got):
        assert np.array_equal(x, y)
    eng.close()


@pytest.mark.gpu
def test_cg_is_refused_for_the_fp32_painn_path(golden):
    """The line search compares energies at the 1e-8 level: only the fp64 potentials offer it (the reference uses the
    LAMMPS minimiser with LAMMPS calculators only)."""
    import ctypes as C

    from surface_sampling_amd import backend

    table, const = golden.offset_table()
    eng = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
    s = golden.structure("SrTiO3_2x2_pristine")
    eng.upload([(s.numbers, s.positions, s.cell, s.pbc)])
    p = backend.CgParams.default()
    rc = eng._lib.vssr_batch_relax_cg(eng._h, C.byref(p), None, 3, None, None, None, None)
    assert rc == -5 and b"fp64" in eng._lib.vssr_last_error(eng._h)
    eng.close()


@pytest.mark.gpu
def test_gan_batched_mc_with_lammps_style_relaxation(golden, oracle_mod):
    """The reference's GaN configuration (``optimizer: "LAMMPS"``: CG relaxation inside every MC step) through the batched MC:
    ``TersoffSurfCalc.relax_batch`` packs all slabs into one ``vssr_batch_relax_cg`` call (advisor finding, round 2: the
    analytic calculators had no ``relax_batch``).  Every stored energy is the oracle's energy of the stored relaxed slab,
    relaxation lowers the energy of every proposal, held atoms stay, two runs agree."""
    from surface_sampling_amd import mc
    from surface_sampling_amd.calculators import TersoffSurfCalc

    g = golden.structure("GaN_3x3_pristine")
    ztop = g.positions[:, 2].max()
    a, b = g.cell[0], g.cell[1]
    coords = np.array([(i + 0.5) / 3 * a + (j + 0.5) / 3 * b for i in range(3) for j in range(3)], float)
    coords[:, 2] = ztop + 1.8
    fixed = np.flatnonzero(g.positions[:, 2] < ztop - 3.0)
    runs = []
    for _ in range(2):
        calc = TersoffSurfCalc(golden.tersoff_params, ["Ga", "N"], device="cuda:0")
        calc.set(relax_steps=30)
        out = calc.relax_batch([g, g.copy()], fixed_indices=[fixed, fixed])
        assert out[0][2] == out[1][2] and out[0][3] is False and out[0][4]["stop"] in (
            "energy tolerance", "force tolerance", "max iterations", "forces are zero", "linesearch alpha is zero")
        ens = mc.ChainEnsemble(g, coords, ("Ga", "N"), 6, calc, seed=2, relax=True, relax_steps=30, fixed_indices=fixed,
                               temperature=0.3, optimizer="LAMMPS")
        ens.initialize()
        for _ in range(3):
            ens.step_semigrand()
        assert (ens.num_adsorbates() > 0).any()
        static = calc.calculate_batch([ens.structure(b) for b in range(6)])
        for b in range(6):
            r = ens.relaxed[b]
            types = np.array([0 if z == 31 else 1 for z in r.numbers], np.int32)
            E, _, _ = oracle_mod.tersoff(golden.tersoff_params, types, r.positions, r.cell, [1, 1, 1])
            assert abs(E - ens.state.energy[b]) <= 1e-9 * max(1.0, abs(E))
            assert ens.state.energy[b] <= static[b]["energy"] + 1e-9            # relaxed <= unrelaxed proposal
            assert np.array_equal(r.positions[fixed], g.positions[fixed])
        runs.append((ens.state.species.copy(), ens.state.energy.copy()))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("relax,optimizer", [(True, "LAMMPS"), (True, "FIRE"), (False, "LAMMPS")])
def test_gan_packed_mc_path_equals_the_per_slab_path(golden, relax, optimizer):
    """``TersoffSurfCalc.evaluate_packed`` (packed arrays in and out, what ``ChainEnsemble`` calls every step) against the
    per-slab path (``relax_batch`` / ``calculate_batch`` with one object per slab): same seed, identical accept masks,
    occupations, energies (bit for bit), per-atom energies and relaxed geometries -- semigrand and canonical steps."""
    from surface_sampling_amd import mc
    from surface_sampling_amd.calculators import TersoffSurfCalc

    g = golden.structure("GaN_3x3_pristine")
    ztop = g.positions[:, 2].max()
    coords = np.array([(i + 0.5) / 3 * g.cell[0] + (j + 0.5) / 3 * g.cell[1] for i in range(3) for j in range(3)], float)
    coords[:, 2] = ztop + 1.8
    fixed = np.flatnonzero(g.positions[:, 2] < ztop - 3.0)
    runs = []
    for fast in (True, False):
        calc = TersoffSurfCalc(golden.tersoff_params, ["Ga", "N"], device="cuda:0")
        calc.set(relax_steps=25)
        ens = mc.ChainEnsemble(g, coords, ("Ga", "N"), 7, calc, seed=6, relax=relax, relax_steps=25, fmax=0.05,
                               fixed_indices=fixed, temperature=0.4, optimizer=optimizer)
        ens.fast_path = fast
        assert ens._packed_supported()
        ens.initialize()
        acc = np.stack([ens.step_semigrand() for _ in range(4)] + [ens.step_canonical() for _ in range(2)])
        runs.append((acc, ens))
    (a, ea), (b, eb) = runs
    assert np.array_equal(a, b) and np.array_equal(ea.state.species, eb.state.species)
    assert np.array_equal(ea.state.energy, eb.state.energy) and np.array_equal(ea.oob, eb.oob)
    for k in range(7):
        assert np.array_equal(ea.relaxed[k].numbers, eb.relaxed[k].numbers)
        assert np.array_equal(ea.relaxed[k].positions, eb.relaxed[k].positions)
        assert np.array_equal(ea.per_atom_energies[k], eb.per_atom_energies[k])
    assert a.any()
    # a host-driven optimizer is not served by the packed path, and asking for it directly fails loudly
    from surface_sampling_amd import backend
    assert not calc.packed_supported(True, "BFGSLineSearch") and calc.packed_supported(False, "BFGSLineSearch")
    n_atoms, numbers, positions, _, _ = ea.batch_arrays(ea.state, np.arange(2))
    with pytest.raises(backend.BackendError, match="on the device only"):
        calc.evaluate_packed(n_atoms, numbers, positions, np.tile(np.ravel(g.cell), (2, 1)), np.ones((2, 3), np.uint8),
                             relax=True, optimizer="BFGSLineSearch")
    with pytest.raises(ValueError, match="not covered"):
        calc.evaluate_packed([1], [38], [[0.0, 0.0, 0.0]], np.ravel(g.cell)[None], np.ones((1, 3), np.uint8))
