"""Batched MC step through the real PaiNN calculator on the GPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _site_grid(base, n=4, dz=1.5):
    ztop = base.positions[:, 2].max()
    a, b = base.cell[0], base.cell[1]
    return np.array([(i + 0.5) / n * a + (j + 0.5) / n * b + np.array([0.0, 0.0, ztop + dz - ((i + 0.5) / n * a + (j + 0.5) / n * b)[2]])
                     for i in range(n) for j in range(n)], float)


def test_semigrand_steps_with_device_relaxation(golden):
    """8 chains, 4 MC steps, every proposed slab relaxed on the device in lock-step: the stored energies are the
    surface energies of the stored states, rejected chains are restored, two runs give identical trajectories."""
    from surface_sampling_amd import mc
    from surface_sampling_amd.calculators import EnsembleNFFSurface

    base = golden.structure("SrTiO3_2x2_pristine")
    coords = _site_grid(base)
    fixed = np.flatnonzero(base.positions[:, 2] < base.positions[:, 2].max() - 4.0)   # bulk_idx-like FixAtoms mask
    runs = []
    for _ in range(2):
        calc = EnsembleNFFSurface(golden.blobs, device="cuda:0", model_units="kcal/mol", prediction_units="eV",
                                  offset_units="atomic")
        calc.set(offset=True, offset_data=golden.offset_data, chem_pots={"Sr": -2, "Ti": 0, "O": 0})
        ens = mc.ChainEnsemble(base, coords, ("Sr", "O"), 8, calc, seed=5, relax=True, relax_steps=5, fmax=0.05,
                               fixed_indices=fixed, temperature=0.5)
        e0 = ens.initialize()
        assert np.allclose(e0, e0[0]) and np.isfinite(e0).all()          # every chain starts from the pristine slab
        accepts, states = [], []
        for _ in range(4):
            before = ens.state.copy()
            acc = ens.step_semigrand()
            accepts.append(acc)
            states.append(ens.state.species.copy())
            rej = ~acc
            assert np.array_equal(ens.state.species[rej], before.species[rej])
            assert np.array_equal(ens.state.energy[rej], before.energy[rej])
        e_check, _ = ens.evaluate(ens.state)
        assert np.allclose(e_check, ens.state.energy, atol=2e-4)          # energy of the kept state, re-relaxed
        assert ens.n_evaluations == 8 * 6
        for b in range(8):
            assert len(ens.relaxed[b]) == len(base) + ens.num_adsorbates()[b]
            moved = np.abs(ens.relaxed[b].positions[: len(base)][fixed] - base.positions[fixed]).max()
            assert moved == 0.0                                            # FixAtoms mask honoured
        runs.append((np.array(accepts), np.array(states), ens.state.energy.copy()))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    assert np.array_equal(runs[0][2], runs[1][2])
    assert runs[0][0].any()                                                # T = 0.5 eV: something is accepted


def test_mc_state_energies_match_the_oracle(golden, oracle_mod):
    """The energies the Metropolis test uses are the reference's numbers: after 3 semigrand steps with device relaxation the stored
    surface energy of every chain equals the fp64 oracle's ensemble energy AT THE STORED RELAXED GEOMETRY, passed through the
    same surface-energy bookkeeping (reference calculators.py:379-446), within the GPU-vs-oracle tolerance."""
    from surface_sampling_amd import mc
    from surface_sampling_amd.calculators import EnsembleNFFSurface, surface_energy_from_energy

    base = golden.structure("SrTiO3_2x2_pristine")
    coords = _site_grid(base)
    fixed = np.flatnonzero(base.positions[:, 2] < base.positions[:, 2].max() - 4.0)
    chem = {"Sr": -2, "Ti": 0, "O": 0}
    calc = EnsembleNFFSurface(golden.blobs, device="cuda:0", model_units="kcal/mol", prediction_units="eV", offset_units="atomic")
    calc.set(offset=True, offset_data=golden.offset_data, chem_pots=chem)
    ens = mc.ChainEnsemble(base, coords, ("Sr", "O"), 6, calc, seed=11, relax=True, relax_steps=4, fmax=0.05,
                           fixed_indices=fixed, temperature=0.5)
    ens.initialize()
    for _ in range(3):
        ens.step_semigrand()
    table, const = golden.offset_table()
    worst = 0.0
    for b in range(6):
        slab = ens.relaxed[b]
        ref = oracle_mod.ensemble(golden.blobs, slab.numbers, slab.positions, slab.cell, slab.pbc, 64, table, const)
        want = surface_energy_from_energy(ref["energy"], slab.get_chemical_symbols(), chem, golden.offset_data, "atomic")
        worst = max(worst, abs(want - float(ens.state.energy[b])))
    assert worst <= 2e-4, worst


@pytest.mark.parametrize("relax", [True, False])
def test_packed_fast_path_equals_the_per_slab_path_on_the_device(golden, relax):
    """``ChainEnsemble`` talks to the calculator in packed arrays (``evaluate_packed``; no per-slab objects, surface energies
    from element counts, relaxed slabs built on demand, the chains split over two concurrent engines).  Same seed through the per-slab path (``relax_batch`` /
    ``calculate_batch`` + ``surface_energy_from_energy`` per slab): identical accept masks, species, energies (bit for bit),
    out-of-bounds flags and relaxed geometries."""
    from surface_sampling_amd import mc
    from surface_sampling_amd.calculators import EnsembleNFFSurface

    base = golden.structure("SrTiO3_2x2_pristine")
    coords = _site_grid(base)
    fixed = np.flatnonzero(base.positions[:, 2] < base.positions[:, 2].max() - 4.0)
    runs = []
    for fast in (True, False):
        calc = EnsembleNFFSurface(golden.blobs, device="cuda:0", model_units="kcal/mol", prediction_units="eV", offset_units="atomic")
        calc.set(offset=True, offset_data=golden.offset_data, chem_pots={"Sr": -2, "Ti": 0, "O": 0})
        ens = mc.ChainEnsemble(base, coords, ("Sr", "O"), 7, calc, seed=21, relax=relax, relax_steps=4, fmax=0.05,
                               fixed_indices=fixed, temperature=0.5)
        ens.fast_path = fast
        calc.MIN_CHAINS_PER_STREAM = 2      # the packed path splits these 7 chains over two engines (two HIP streams, two host threads)
        ens.initialize()
        acc = np.stack([ens.step_semigrand() for _ in range(3)] + [ens.step_canonical() for _ in range(2)])
        assert len(calc.__dict__.get("_extra_engines", [])) == (1 if fast and relax else 0)
        runs.append((acc, ens.state.species.copy(), ens.state.energy.copy(), ens.oob.copy(), ens))
    a, b = runs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
    assert np.array_equal(a[2], b[2])                                      # bit for bit
    for k in range(7):
        ra, rb = a[4].relaxed[k], b[4].relaxed[k]
        assert np.array_equal(ra.numbers, rb.numbers) and np.array_equal(ra.positions, rb.positions)
        assert np.array_equal(a[4].per_atom_energies[k], b[4].per_atom_energies[k])
    assert a[0].any() or not relax      # (unrelaxed adsorbates 1.5 A above the surface are rarely accepted at T = 0.5 eV)


@pytest.mark.parametrize("relax", [False, True])
def test_concurrent_chain_groups_on_the_device(golden, relax):
    """``mc.ConcurrentChains``: two groups of chains with their own calculators (own engines / HIP streams), advanced by their
    own host threads -- the device work of one overlaps the host work of the other -- end in the states of ONE ensemble over
    the same chains: accept counts, occupations and energies bit for bit."""
    from surface_sampling_amd import mc
    from surface_sampling_amd.calculators import EnsembleNFFSurface

    base = golden.structure("SrTiO3_2x2_pristine")
    coords = _site_grid(base)
    fixed = np.flatnonzero(base.positions[:, 2] < base.positions[:, 2].max() - 4.0)

    def new_calc():
        c = EnsembleNFFSurface(golden.blobs, device="cuda:0", model_units="kcal/mol", prediction_units="eV", offset_units="atomic")
        c.set(offset=True, offset_data=golden.offset_data, chem_pots={"Sr": -2, "Ti": 0, "O": 0})
        return c
    kw = dict(seed=9, relax=relax, relax_steps=3, fmax=0.05, fixed_indices=fixed, temperature=0.5)
    whole = mc.ChainEnsemble(base, coords, ("Sr", "O"), 9, new_calc(), **kw)
    whole.initialize()
    want = np.zeros(9, np.int64)
    for _ in range(4):
        want += whole.step_semigrand()
    groups = mc.ConcurrentChains.build(base, coords, ("Sr", "O"), 9, [new_calc(), new_calc()], **kw)
    groups.initialize()
    got = groups.steps(4)
    assert np.array_equal(got, want) and np.array_equal(groups.species, whole.state.species)
    assert np.array_equal(groups.energy, whole.state.energy)
    assert groups.n_evaluations == whole.n_evaluations
