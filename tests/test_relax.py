"""Lock-step FIRE relaxation (reference mcmc/dynamics.py:83-170)."""

import numpy as np
import pytest

from conftest import top_layer
from fire_oracle import fire_relax


def test_fire_restatement_invariants_cpu(golden, oracle_mod):
    """CPU: the numpy FIRE restatement on oracle forces lowers the energy and the max force (free = top layer)."""
    s = golden.structure("O44Sr12Ti16")
    table, const = golden.offset_table()
    free = top_layer(s)
    fixed = np.setdiff1d(np.arange(len(s)), free)

    def force_fn(pos):
        r = oracle_mod.ensemble(golden.blobs, s.numbers, pos, s.cell, s.pbc, 64, table, const)
        return r["energy"], r["forces"]

    pos, energies, steps, conv = fire_relax(force_fn, s.positions, fixed=fixed, max_steps=6)
    assert steps == 6 and not conv
    assert abs(energies[0] - (-570.127991)) < 2e-4            # reference BFGS step 0 print
    assert energies[-1] < energies[0] - 5e-3                  # relaxing the DL-TiO2 termination gains > 5 meV in 6 steps
    assert np.abs(pos[fixed] - s.positions[fixed]).max() == 0.0
    assert np.abs(pos[free] - s.positions[free]).max() < 0.25


def test_fire_params_struct_layout():
    import ctypes

    from surface_sampling_amd import backend

    p = backend.FireParams.default(20, 0.01)
    assert ctypes.sizeof(backend.FireParams) == 40 and p.max_steps == 20 and abs(p.fmax - 0.01) < 1e-9
    assert (p.dt, p.nmin) == (pytest.approx(0.1), 5)


@pytest.mark.gpu
def test_fire_gpu_matches_restatement_and_batches(golden, oracle_mod):
    from surface_sampling_amd import backend, structures

    table, const = golden.offset_table()
    eng = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
    slabs = [golden.structure("O44Sr12Ti16"), golden.structure("O40Sr16Ti12"),
             structures.synth_chain(golden.structure("SrTiO3_2x2_pristine"), 3, grid=(4, 4))]
    fixed_idx = [np.setdiff1d(np.arange(len(s)), top_layer(s)) for s in slabs[:2]] + [np.arange(60)]
    mask = np.zeros(sum(len(s) for s in slabs), np.uint8)
    o = 0
    for s, idx in zip(slabs, fixed_idx):
        mask[o + idx] = 1
        o += len(s)
    eng.upload([(s.numbers, s.positions, s.cell, s.pbc) for s in slabs])
    e0 = eng.evaluate([(s.numbers, s.positions, s.cell, s.pbc) for s in slabs])["energy"].astype(np.float64)
    eng.upload([(s.numbers, s.positions, s.cell, s.pbc) for s in slabs])
    info = eng.relax_fire(fixed=mask, max_steps=8, fmax=0.01)
    res = eng.download()
    cs = res["cfg_start"]
    assert (info["n_steps"] == 8).all() and not info["converged"].any()
    assert (res["energy"].astype(np.float64) < e0 - 1e-3).all()     # every chain went downhill
    o = 0
    for b, (s, idx) in enumerate(zip(slabs, fixed_idx)):
        p = info["positions"][cs[b]:cs[b + 1]]
        assert np.abs(p[idx] - s.positions[idx]).max() == 0.0        # FixAtoms respected exactly

        def force_fn(pos, s=s):
            r = oracle_mod.ensemble(golden.blobs, s.numbers, pos, s.cell, s.pbc, 64, table, const)
            return r["energy"], r["forces"]

        pref, eref, steps, conv = fire_relax(force_fn, s.positions, fixed=idx, max_steps=8)
        assert np.abs(p - pref).max() < 2e-3                          # same algorithm, fp32 vs fp64 forces
        assert abs(float(res["energy"][b]) - eref[-1]) < 5e-4
    # a chain that starts converged does not move and reports 0 steps; relaxing it changes nothing
    info2 = eng.relax_fire(fixed=np.ones_like(mask), max_steps=5, fmax=0.01)
    assert (info2["n_steps"] == 0).all() and info2["converged"].all()
    assert np.array_equal(info2["positions"], info["positions"])
    eng.close()


@pytest.mark.gpu
def test_relax_batch_front_end(golden):
    """EnsembleNFFSurface.relax_batch returns the reference's (slab, traj, energy, energy_oob) tuple per slab."""
    from surface_sampling_amd.calculators import EnsembleNFFSurface

    calc = EnsembleNFFSurface(golden.blobs, device="cuda:0")
    calc.set(offset=True, offset_data=golden.offset_data, relax_steps=20)
    s = golden.structure("O36Sr12Ti12")
    fixed = np.setdiff1d(np.arange(len(s)), top_layer(s))
    out = calc.relax_batch([s, s.copy()], fixed_indices=[fixed, fixed], relax_steps=20, fmax=0.01)
    (slab, traj, energy, oob, r), (slab2, _, energy2, _, _) = out
    assert traj is None and oob is False and energy == energy2
    assert energy < -467.525604 + 1e-4          # not above the unrelaxed reference energy
    assert energy > -467.56                      # reference BFGS reaches -467.534088 (tests/test_SrTiO3_terms.ipynb:208-210)
    assert np.abs(slab.positions[fixed] - s.positions[fixed]).max() == 0.0
    assert r["n_steps"] <= 20


@pytest.mark.gpu
@pytest.mark.parametrize("optimizer", ["BFGS", "FIRE"])
def test_relax_batch_records_the_trajectory_like_the_reference_observer(golden, optimizer):
    """``relax_batch(save_traj=True, record_interval=k)`` = ``optimize_slab(save_traj=True, record_interval=k)`` (reference
    mcmc/dynamics.py:131-151): the observer fires after 0, k, 2k, ... optimizer steps.  Checked against relaxations of the
    same starts that STOP at those step counts, and against the single-point evaluation of the start."""
    from surface_sampling_amd.calculators import EnsembleNFFSurface

    calc = EnsembleNFFSurface(golden.blobs, device="cuda:0")
    calc.set(offset=True, offset_data=golden.offset_data)
    slabs = [golden.structure("O36Sr12Ti12"), golden.structure("O44Sr12Ti16"), golden.structure("O40Sr16Ti12")]
    fixed = [np.setdiff1d(np.arange(len(s)), top_layer(s)) for s in slabs]
    k, steps = 2, 5
    out = calc.relax_batch(slabs, fixed_indices=fixed, relax_steps=steps, fmax=0.01, optimizer=optimizer, save_traj=True,
                           record_interval=k)
    start = calc.calculate_batch(slabs)
    plain = calc.relax_batch(slabs, fixed_indices=fixed, relax_steps=steps, fmax=0.01, optimizer=optimizer)
    for b, (slab, traj, energy, oob, r) in enumerate(out):
        assert plain[b][1] is None and plain[b][2] == energy                # recording does not change the relaxation
        assert set(traj) == {"atoms", "energies", "forces"}
        n_rec = r["n_steps"] // k + 1
        assert len(traj["atoms"]) == len(traj["energies"]) == len(traj["forces"]) == n_rec >= 2
        assert traj["energies"][0] == float(start[b]["energy"][0])           # record 0 = the start, before any step
        assert np.array_equal(traj["atoms"][0].positions, slabs[b].positions)
        f0 = start[b]["forces"].copy()
        f0[fixed[b]] = 0.0                                                   # atoms.get_forces() applies FixAtoms
        assert np.array_equal(traj["forces"][0], f0)
        for t in traj["forces"]:
            assert not t[fixed[b]].any()
    for rec in (1, 2):                                                       # records after k and 2k steps
        part = calc.relax_batch(slabs, fixed_indices=fixed, relax_steps=rec * k, fmax=0.01, optimizer=optimizer)
        for b in range(len(slabs)):
            if len(out[b][1]["atoms"]) <= rec:
                continue
            assert np.abs(out[b][1]["atoms"][rec].positions - part[b][0].positions).max() < 1e-9
            assert abs(out[b][1]["energies"][rec] - float(part[b][4]["energy"][0])) < 1e-5


@pytest.mark.gpu
def test_multi_pass_neighbor_sums_inside_a_relaxation_with_regrow_and_drop_out(golden, monkeypatch):
    """The multi-pass forms of the neighbor-sum kernels (what chains of more than 405 / 557 atoms take; here forced onto 260-atom
    chains with ranges of 100 atoms) inside the lock-step BFGS driver: a neighbor capacity that is too small at first (buffers and
    per-pass bundle tables are rebuilt), a chain that converges early and drops out (activity mask), FixAtoms -- same relaxed
    geometries and energies as the default single-pass kernels to fp32 rounding, identical when repeated."""
    from surface_sampling_amd import backend, structures

    table, const = golden.offset_table()
    big = golden.structure("SrTiO3_2x2_pristine").repeat((2, 2, 1))
    chains = [structures.synth_chain(big, 5), structures.synth_chain(big, 9), structures.synth_chain(big, 13)]
    packs = [(c.numbers, c.positions, c.cell, c.pbc) for c in chains]
    mask = np.concatenate([(c.positions[:, 2] < c.positions[:240, 2].max() - 4.0).astype(np.uint8) for c in chains])
    mask[len(chains[0]):len(chains[0]) + len(chains[1])] = 1          # chain 1: everything held -> converged from the start, drops out
    runs = []
    for knobs in ({}, {"VSSR_EDGE_FS16_MAX": "0", "VSSR_EDGE_FS8_MAX": "0", "VSSR_EDGE_BWD_MPASS": "2", "VSSR_EDGE_SUB_CHUNK": "100"}):
        for k, v in knobs.items():
            monkeypatch.setenv(k, v)
        eng = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
        for k in knobs:
            monkeypatch.delenv(k)
        out = []
        for rep in range(2):
            eng.debug_capacity(slots_per_atom=16, tight=1)           # too small: the relaxation has to regrow
            eng.upload(packs)
            info = eng.relax_bfgs(fixed=mask, max_steps=6, fmax=0.01)
            res = eng.download()
            out.append((info["positions"].copy(), res["energy_f64"].copy(), res["forces"].copy(), info["n_steps"].copy()))
            assert eng.debug_capacity() >= 1
        assert all(np.array_equal(a, b) for a, b in zip(out[0], out[1]))
        runs.append(out[0])
        eng.close()
    (p0, e0, f0, n0), (p1, e1, f1, n1) = runs
    assert np.array_equal(n0, n1) and n0[1] == 0 and n0[0] == 6 and n0[2] == 6
    assert np.abs(p0 - p1).max() < 2e-4 and np.abs(e0 - e1).max() < 2e-4 and np.abs(f0 - f1).max() < 2e-3
    assert np.array_equal(p0[mask.astype(bool)], np.concatenate([c.positions for c in chains])[mask.astype(bool)])


@pytest.mark.gpu
def test_stats_after_a_relaxation_need_a_full_run(golden):
    """A lock-step relaxation in which a chain converged EARLY leaves a resident graph that covers only the chains of its last
    iteration: the introspection calls say so instead of returning partial data (advisor finding, round 2); one full run makes
    them valid again.  A relaxation that uses its whole step budget with every chain running -- the normal case with the
    reference's settings -- ends on a complete, unmasked evaluation and leaves everything valid (advisor finding, round 4)."""
    from surface_sampling_amd import backend

    table, const = golden.offset_table()
    eng = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
    s = golden.structure("O36Sr12Ti12")
    rattled = s.positions + np.random.default_rng(5).normal(0, 0.15, s.positions.shape)
    packs = [(s.numbers, s.positions, s.cell, s.pbc), (s.numbers, rattled, s.cell, s.pbc)]
    eng.upload(packs)
    eng.run()
    before = eng.stats()
    f = np.linalg.norm(eng.download()["forces"].astype(np.float64), axis=1)
    f_a, f_b = f[:len(s.numbers)].max(), f[len(s.numbers):].max()
    assert f_b > 3 * f_a
    # (i) nobody converges within the budget: complete graph, introspection valid without another run
    info = eng.relax("BFGS", max_steps=3, fmax=1e-4)
    assert not info["converged"].any() and (info["n_steps"] == 3).all()
    assert np.isfinite(eng.download()["energy"]).all()
    assert eng.stats()["atoms"] == before["atoms"] and len(eng.neighbors()[0]) > 0
    assert eng.debug_read("e_atom", 0).size == before["atoms"]
    st, _ = eng.stress()
    assert np.isfinite(st).all()
    # (ii) chain 0 is converged from the start, chain 1 keeps stepping: the mask goes in at the first poll
    eng.upload(packs)
    info = eng.relax("FIRE", max_steps=8, fmax=1.05 * f_a)
    assert info["converged"][0] and info["n_steps"][0] == 0 and info["n_steps"][1] >= 4
    res = eng.download()                                                      # results stay available
    assert np.isfinite(res["energy"]).all()
    for call in (eng.stats, eng.neighbors, lambda: eng.debug_read("e_atom", 0)):
        with pytest.raises(backend.BackendError, match="last relaxation iteration"):
            call()
    eng.run()
    assert eng.stats()["atoms"] == before["atoms"]
    eng.close()


@pytest.mark.gpu
def test_tersoff_relaxation(golden, oracle_mod):
    """GaN (config 2): run_lammps_opt (LAMMPS-style CG on the device, like the reference's LAMMPS minimiser; FIRE / BFGS on
    request) lowers the energy of a rattled slab back towards the pristine minimum."""
    from surface_sampling_amd.calculators import TersoffSurfCalc
    from surface_sampling_amd.structures import Structure

    g = golden.structure("GaN_3x3_pristine")
    rng = np.random.default_rng(2)
    rattled = Structure(g.numbers, g.positions + rng.normal(0, 0.04, g.positions.shape), g.cell, g.pbc)
    calc = TersoffSurfCalc(golden.tersoff_params, ["Ga", "N"], device="cuda:0")
    calc.set(relax_steps=60)
    e0 = calc.get_potential_energy(rattled)
    relaxed, e1, ea = calc.run_lammps_opt(rattled, fixed_indices=np.arange(0, 12))
    assert e1 < e0 - 0.05 and e1 >= -144.059 - 0.6      # towards (a relaxed variant of) the pristine slab
    assert abs(ea.sum() - e1) < 1e-9
    assert np.abs(relaxed.positions[:12] - rattled.positions[:12]).max() == 0.0
    types = np.array([0 if z == 31 else 1 for z in g.numbers], np.int32)
    E, _, F = oracle_mod.tersoff(golden.tersoff_params, types, relaxed.positions, g.cell, [1, 1, 1])
    assert abs(E - e1) <= 1e-9 * abs(E)
    assert calc.last_opt["optimizer"] == "CG" and calc.last_opt["stop"] in ("energy tolerance", "force tolerance", "max iterations")
    _, e_free, _ = calc.run_lammps_opt(rattled)              # no FixAtoms mask, like the calculator property
    assert calc.get_property("relaxed_energy", rattled) == pytest.approx(e_free, abs=1e-9)
    assert e_free <= e1 + 1e-9
    # the ASE-style optimizers on the same path land in the same basin
    _, e_fire, _ = calc.run_lammps_opt(rattled, fixed_indices=np.arange(0, 12), optimizer="FIRE")
    calc.set(relax_steps=40)                                 # (device BFGS: at most 46 steps, the reference uses 20)
    _, e_bfgs, _ = calc.run_lammps_opt(rattled, fixed_indices=np.arange(0, 12), optimizer="BFGS", fmax=1e-3)
    assert abs(e_fire - e1) < 0.05 and abs(e_bfgs - e1) < 0.05 and e_bfgs <= e1 + 1e-6
