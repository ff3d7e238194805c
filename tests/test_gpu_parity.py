"""GPU (-m gpu): parity of the HIP path, called through the C ABI, against the CPU oracle.

Stated tolerances: GPU fp32 vs the fp64 oracle |dE| <= 1e-4 eV per structure
(N <= ~270), max|dF| <= 2e-4 eV/A; vs the reference's printed known answers |dE| <= 2e-4 eV,
|d fmax| <= 1e-5 eV/A; neighbor edge sets identical; Tersoff (fp64 on device) <= 1e-9 relative.
"""

import os

import numpy as np
import pytest

from conftest import top_layer

pytestmark = pytest.mark.gpu

E_TOL = 1e-4      # eV, GPU fp32 vs fp64 oracle
E_TOL_LARGE = 1e-4   # eV, chains of 300 .. 1 462 atoms (beyond the survey's range; |E| up to 3.8 keV) on the fp64 output word: measured <= 1.4e-5 (profiles/r05/energy_words.jsonl)
F_TOL = 2e-4      # eV/A
STD_TOL = 2e-4


@pytest.fixture(scope="module")
def engine(golden):
    from surface_sampling_amd import backend

    table, const = golden.offset_table()
    eng = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
    yield eng
    eng.close()


def _arrays(s):
    return (s.numbers, s.positions, s.cell, s.pbc)


def _oracle(golden, oracle_mod, s, bits=64):
    table, const = golden.offset_table()
    return oracle_mod.ensemble(golden.blobs, s.numbers, s.positions, s.cell, s.pbc, bits, table, const)


def test_kat_structures_vs_reference_prints_and_oracle(golden, oracle_mod, engine):
    from surface_sampling_amd.calculators import EnsembleNFFSurface

    tol = golden.kat["tolerance"]
    calc32 = EnsembleNFFSurface(golden.blobs, device=0, position_dtype="float32")
    calc32.set(offset=True, offset_data=golden.offset_data)
    for case in golden.kat["painn_ensemble"]:
        s = golden.structure(case["structure"])
        res = engine.evaluate([_arrays(s)])
        ref = _oracle(golden, oracle_mod, s)
        e = float(res["energy"][0])
        assert abs(e - case["energy"]) <= tol["energy_abs"], (case["structure"], e)
        assert abs(e - ref["energy"]) <= E_TOL
        free = top_layer(s) if case["free_atoms"] == "top_layer" else np.array(case["free_atoms"])
        fmax = np.linalg.norm(res["forces"][free].astype(np.float64), axis=1).max()
        assert abs(fmax - case["fmax"]) <= 2e-5, (case["structure"], fmax)  # fp64 positions vs the reference's fp32 run
        # the reference evaluates fp32 positions (nff AtomsBatch tensors): the calculator's position_dtype="float32" reproduces that
        # rounding, and through it the print is met at the tolerance SURVEY.md section 8(c) states
        calc32.calculate(s, properties=("energy", "forces"))
        fmax32 = np.linalg.norm(calc32.results["forces"][free].astype(np.float64), axis=1).max()
        print(f"{case['structure']}: |fmax - print| fp64 positions {abs(fmax - case['fmax']):.2e}, "
              f"EnsembleNFFSurface(position_dtype='float32') {abs(fmax32 - case['fmax']):.2e}")
        assert abs(fmax32 - case["fmax"]) <= tol["fmax_abs"], (case["structure"], fmax32)
        assert abs(float(calc32.results["energy"][0]) - case["energy"]) <= tol["energy_abs"]
        assert np.abs(res["forces"] - ref["forces"]).max() <= F_TOL
        assert abs(float(res["energy_std"][0]) - ref["energy_std"]) <= STD_TOL
        assert np.abs(res["forces_std"] - ref["forces_std"]).max() <= STD_TOL
        assert np.abs(res["energy_models"][0] - ref["energy_models"]).max() <= 2e-4


def test_neighbor_edge_sets_identical(golden, oracle_mod, engine):
    from surface_sampling_amd import structures

    base = golden.structure("SrTiO3_2x2_pristine")
    big = base.repeat((2, 2, 1))
    shifted = big.copy()
    shifted.positions[::5] += 3 * big.cell[0] - 2 * big.cell[1]  # atoms far outside the cell
    batch = [base, golden.structure("O44Sr12Ti16"), structures.synth_chain(big, 4), shifted,
             golden.structure("SrTiO3_2x2x4_pristine")]
    engine.evaluate([_arrays(s) for s in batch])
    ei, ej, eS, er = engine.neighbors()
    start = 0
    got_all = sorted(zip(ei.tolist(), ej.tolist(), map(tuple, eS.tolist())))
    want_all = []
    for s in batch:
        oi, oj, oS, orr = oracle_mod.neighbors(s.positions, s.cell, s.pbc, 5.0)
        want_all += list(zip((oi + start).tolist(), (oj + start).tolist(), map(tuple, oS.tolist())))
        start += len(s)
    assert got_all == sorted(want_all)
    st = engine.stats()
    assert st["edges"] == len(want_all) and st["atoms"] == start and st["slots"] >= st["edges"]
    # stored fp32 edge vectors agree with the fp64 ones
    s = batch[0]
    oi, oj, oS, orr = oracle_mod.neighbors(s.positions, s.cell, s.pbc, 5.0)
    n0 = len(oi)
    key_g = np.lexsort((eS[:n0, 2], eS[:n0, 1], eS[:n0, 0], ej[:n0], ei[:n0]))
    key_o = np.lexsort((oS[:, 2], oS[:, 1], oS[:, 0], oj, oi))
    assert np.abs(er[:n0][key_g] - orr[key_o]).max() < 2e-6


def test_per_layer_intermediates_match_oracle(golden, oracle_mod, engine):
    """Every stored activation of every layer vs the fp64 oracle (model 2 of the ensemble)."""
    from surface_sampling_amd import structures

    s = structures.synth_chain(golden.structure("SrTiO3_2x2_pristine"), 2, grid=(4, 4))
    engine.evaluate([_arrays(s)])
    m = 1
    E, G, d = oracle_mod.painn(golden.blobs[m], s.numbers, s.positions, s.cell, s.pbc, 64, dump=True)
    n = len(s)
    report = []
    for l in range(3):
        for name, shape in (("phi", (n, 384)), ("s_msg", (n, 128)), ("v_msg", (n, 3, 128)),
                            ("s_upd", (n, 128)), ("v_upd", (n, 3, 128))):
            try:
                got = engine.debug_read(f"{name}{l}", m).reshape(shape).astype(np.float64)
            except Exception as exc:   # phi0 is not materialised when layer 0 is evaluated by species factorisation, and
                # nothing consumes the vector output of the last block (checked below on a handle that keeps it)
                assert "not materialised" in str(exc) and ((name == "phi" and l == 0) or (name == "v_upd" and l == 2))
                continue
            want = d[name][l]
            scale = max(1.0, np.abs(want).max())
            report.append((f"{name}{l}", np.abs(got - want).max() / scale))
    got = engine.debug_read("e_atom", m).astype(np.float64)
    report.append(("e_atom", np.abs(got - d["e_atom"]).max() / max(1.0, np.abs(d["e_atom"]).max())))
    for name, shape in (("sbar_msg", (n, 128)), ("vbar_msg", (n, 3, 128))):
        got = engine.debug_read(f"{name}0", m).reshape(shape).astype(np.float64)
        want = d[name][0]
        report.append((f"{name}0", np.abs(got - want).max() / max(1.0, np.abs(want).max())))
    # the latent embedding the reference's clustering helpers read = the final scalar features (vssr_batch_embedding)
    emb = engine.embedding()
    assert emb.shape == (3, n, 128) and np.array_equal(emb[m], engine.embedding(m))
    assert np.array_equal(emb[m].reshape(-1), engine.debug_read("s_upd2", m))
    report.append(("embedding", np.abs(emb[m].astype(np.float64) - d["s_upd"][2]).max() / max(1.0, np.abs(d["s_upd"][2]).max())))
    bad = [(k, v) for k, v in report if not v < 2e-5]
    assert not bad, f"intermediates off: {bad}; all: {report}"
    assert len(report) >= 17


def test_last_block_vector_output_on_request(golden, oracle_mod, monkeypatch):
    """VSSR_DEBUG_KEEP=1 materialises the one activation the product path drops (the last update block's vector output: only
    its scalar output reaches the readout): it matches the oracle, and keeping it changes no result."""
    from surface_sampling_amd import backend, structures

    s = structures.synth_chain(golden.structure("SrTiO3_2x2_pristine"), 2, grid=(4, 4))
    table, const = golden.offset_table()
    monkeypatch.setenv("VSSR_DEBUG_KEEP", "1")
    keep = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
    monkeypatch.delenv("VSSR_DEBUG_KEEP")
    lean = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
    rk, rl = keep.evaluate([_arrays(s)]), lean.evaluate([_arrays(s)])
    for k in ("energy", "forces", "energy_std", "forces_std"):
        assert np.array_equal(rk[k], rl[k]), k
    _, _, d = oracle_mod.painn(golden.blobs[1], s.numbers, s.positions, s.cell, s.pbc, 64, dump=True)
    got = keep.debug_read("v_upd2", 1).reshape(len(s), 3, 128).astype(np.float64)
    want = d["v_upd"][2]
    assert np.abs(got - want).max() / max(1.0, np.abs(want).max()) < 2e-5
    with pytest.raises(backend.BackendError, match="not materialised"):
        lean.debug_read("v_upd2", 1)
    keep.close()
    lean.close()


def test_embedding_through_the_calculator_and_helpers(golden, oracle_mod):
    """``EnsembleNFFSurface(properties=(..., "embedding"))`` + the reference's helper functions
    (``get_results_single / get_embeddings_single / get_std_devs_single``, mcmc/calculators/calculators.py:34-135; call
    pattern of scripts/clustering.py:236-249) against the oracle's final scalar features of model 0."""
    from surface_sampling_amd import calculators as calcs

    s = golden.structure("O40Sr16Ti12")
    single = calcs.EnsembleNFFSurface(golden.blobs[:1], device="cuda:0", properties=("energy", "forces", "embedding"))
    ens = calcs.EnsembleNFFSurface(golden.blobs, device="cuda:0")
    for c in (single, ens):   # (calculate() with the default property list includes the surface energy, like the reference's)
        c.set(offset=True, offset_data=golden.offset_data, chem_pots={"Sr": -2, "Ti": 0, "O": 0})
    res = calcs.get_results_single(s, single)
    emb = calcs.get_embeddings_single(s, single, results_cache=res, flatten=True, flatten_axis=0)
    _, _, d = oracle_mod.painn(golden.blobs[0], s.numbers, s.positions, s.cell, s.pbc, 64, dump=True)
    want = d["s_upd"][2]
    ref = _oracle(golden, oracle_mod, s)
    assert res["embedding"].shape == (len(s), 128) and emb.shape == (128,)
    assert np.abs(res["embedding"] - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
    assert np.allclose(emb, want.mean(axis=0), atol=2e-5)
    assert "embedding" not in calcs.get_results_single(s, ens)           # not requested: not computed
    fstd = calcs.get_std_devs_single(s, ens)
    assert abs(float(fstd) - float(ref["forces_std"].mean())) <= 1e-5
    assert calcs.get_std_devs_single(s, single) == 0.0
    batch = ens.calculate_batch([s, golden.structure("O36Sr12Ti12")], want_embedding=True)
    assert batch[0]["embedding_models"].shape == (3, len(s), 128) and batch[1]["embedding"].shape == (60, 128)
    assert np.array_equal(batch[0]["embedding"], res["embedding"])       # model 0 of the ensemble = the single-model calculator


def test_batched_ragged_chains_vs_oracle_and_vs_single(golden, oracle_mod, engine):
    from surface_sampling_amd import structures

    big = golden.structure("SrTiO3_2x2_pristine").repeat((2, 2, 1))
    chains = [structures.synth_chain(big, c) for c in (0, 7, 24)] + [golden.structure("O40Sr16Ti12")]
    res = engine.evaluate([_arrays(s) for s in chains])
    for b, s in enumerate(chains):
        a0, a1 = res["cfg_start"][b], res["cfg_start"][b + 1]
        assert a1 - a0 == len(s)
        ref = _oracle(golden, oracle_mod, s)
        assert abs(float(res["energy"][b]) - ref["energy"]) <= E_TOL, (b, float(res["energy"][b]), ref["energy"])
        assert np.abs(res["forces"][a0:a1] - ref["forces"]).max() <= F_TOL
        single = engine.evaluate([_arrays(s)])
        assert float(single["energy"][0]) == float(res["energy"][b])  # batching changes nothing, bit for bit
        assert np.array_equal(single["forces"], res["forces"][a0:a1])
    # committed fp64 vectors (guards against oracle and GPU drifting together)
    f = golden.fine
    for name in ("S240", "chain17"):
        r = engine.evaluate([(f[f"{name}.numbers"], f[f"{name}.positions"], f[f"{name}.cell"], f[f"{name}.pbc"])])
        assert abs(float(r["energy"][0]) - float(f[f"{name}.energy"])) <= E_TOL
        assert np.abs(r["forces"] - f[f"{name}.forces"]).max() <= F_TOL


def test_run_to_run_determinism_and_energy_only(golden, engine):
    from surface_sampling_amd import backend, structures

    big = golden.structure("SrTiO3_2x2_pristine").repeat((2, 2, 1))
    chains = [_arrays(structures.synth_chain(big, c)) for c in range(6)]
    a = engine.evaluate(chains)
    b = engine.evaluate(chains)
    for k in ("energy", "forces", "energy_std", "forces_std"):
        assert np.array_equal(a[k], b[k]), k
    engine.upload(chains)
    engine.run(backend.WANT_ENERGY)
    e_only = engine.download(backend.WANT_ENERGY)
    assert np.array_equal(e_only["energy"], a["energy"])


def test_full_size_properties_256_chains(golden, engine):
    """BASELINE config 4 size (256 chains x ~260 atoms): size-independent physical properties."""
    from surface_sampling_amd import structures

    big = golden.structure("SrTiO3_2x2_pristine").repeat((2, 2, 1))
    chains = [structures.synth_chain(big, c) for c in range(256)]
    res = engine.evaluate([_arrays(s) for s in chains])
    assert np.isfinite(res["energy"]).all() and np.isfinite(res["forces"]).all()
    assert not res["saturated"].any()
    cs = res["cfg_start"]
    # (1) translation invariance: net force on every chain vanishes
    net = np.array([res["forces"][cs[b]:cs[b + 1]].astype(np.float64).sum(0) for b in range(256)])
    assert np.abs(net).max() < 5e-4
    # (2) chains c and c+25k share the adsorbate count rule; identical inputs give identical outputs
    dup = engine.evaluate([_arrays(chains[3]), _arrays(chains[3])])
    assert dup["energy"][0] == dup["energy"][1] == res["energy"][3]
    # (3) rigid translation + lattice-vector wrap + atom permutation leave E unchanged, permute F
    s = chains[11]
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(s))
    moved = structures.Structure(s.numbers[perm], s.positions[perm] + np.array([1.234, -0.77, 0.0]) + 2 * s.cell[1],
                                 s.cell, s.pbc)
    r2 = engine.evaluate([_arrays(moved)])
    assert abs(float(r2["energy"][0]) - float(res["energy"][11])) < 2e-4
    assert np.abs(r2["forces"] - res["forces"][cs[11]:cs[12]][perm]).max() < 3e-4
    st = engine.stats()
    assert st["atoms"] == len(moved)


def test_forces_are_energy_gradient_on_gpu(golden, engine):
    """Central finite differences of the GPU energy along random directions vs GPU forces."""
    from surface_sampling_amd import backend, structures

    s = structures.synth_chain(golden.structure("SrTiO3_2x2_pristine"), 1, grid=(4, 4))
    res = engine.evaluate([_arrays(s)])
    rng = np.random.default_rng(5)
    h = 2e-3
    for _ in range(3):
        dirn = rng.normal(size=s.positions.shape)
        dirn /= np.linalg.norm(dirn)
        plus = engine.evaluate([(s.numbers, s.positions + h * dirn, s.cell, s.pbc)], backend.WANT_ENERGY | backend.WANT_PER_MODEL)
        minus = engine.evaluate([(s.numbers, s.positions - h * dirn, s.cell, s.pbc)], backend.WANT_ENERGY | backend.WANT_PER_MODEL)
        fd = -(float(plus["energy"][0]) - float(minus["energy"][0])) / (2 * h)
        an = float((res["forces"].astype(np.float64) * dirn).sum())
        assert abs(fd - an) < 5e-2 * max(1.0, abs(an)), (fd, an)  # fp32 energies: coarse FD only


def test_tersoff_gpu_vs_oracle(golden, oracle_mod):
    from surface_sampling_amd import backend

    eng = backend.TersoffEngine(golden.tersoff_params, device=0)
    s = golden.structure("GaN_3x3_pristine")
    types = np.array([0 if z == 31 else 1 for z in s.numbers], np.int32)
    f = golden.fine
    batch = [(types, s.positions, s.cell, [1, 1, 1]), (f["GaN_rattled.types"], f["GaN_rattled.positions"], s.cell, [1, 1, 1])]
    rng = np.random.default_rng(11)
    ads = np.r_[s.positions, s.positions[18:30] + [0.3, 0.2, 1.9]] + rng.normal(0, 0.03, (48, 3))
    batch.append((np.r_[types, np.zeros(12, np.int32)], ads, s.cell, [1, 1, 1]))  # 36 + 12 Ga (config 2)
    e, ea, F = eng.evaluate_f64(batch)
    assert abs(e[0] - golden.kat["tersoff"]["energy"]) <= golden.kat["tolerance"]["tersoff_energy_abs"]
    o = 0
    for b, (t, p, c, pbc) in enumerate(batch):
        E0, ea0, F0 = oracle_mod.tersoff(golden.tersoff_params, t, p, c, pbc)
        n = len(t)
        assert abs(e[b] - E0) <= 1e-9 * abs(E0)
        assert np.abs(ea[o:o + n] - ea0).max() <= 1e-9
        assert np.abs(F[o:o + n] - F0).max() <= 1e-8
        o += n
    eng.close()


def test_tersoff_silicon_literature_values_on_the_device(oracle_mod):
    """Tersoff's published Si(C) set (n = 0.78734, beta = 1.1e-6: the b_ij branch GaN.tersoff never takes) on the device: the
    diamond lattice at the published a0 = 5.432 A gives the published cohesive energy of 4.63 eV per atom, the lattice-constant scan
    has its minimum there, and a rattled 64-atom supercell agrees with the fp64 oracle (tests/test_oracle_kat.py pins the oracle on
    the same values)."""
    from conftest import SI_T3, SI_T3_A0, SI_T3_ECOH, diamond_cell
    from surface_sampling_amd import backend

    eng = backend.TersoffEngine(SI_T3, device=0)
    scan_a = (5.430, 5.431, 5.432, 5.433, 5.434)
    batch = [diamond_cell(a) + ([1, 1, 1],) for a in scan_a]
    rng = np.random.default_rng(8)
    t, x, c = diamond_cell(SI_T3_A0)
    X = np.concatenate([x + np.array([i, j, k]) * SI_T3_A0 for i in range(2) for j in range(2) for k in range(2)])
    X = X + rng.normal(0, 0.08, X.shape)
    batch.append((np.zeros(len(X), np.int32), X, c * 2, [1, 1, 1]))
    e, ea, F = eng.evaluate_f64(batch)
    per_atom = e[:5] / 8
    assert abs(per_atom[2] - SI_T3_ECOH) <= 5e-4, per_atom
    assert int(np.argmin(per_atom)) == 2, per_atom
    assert np.abs(F[:40]).max() < 1e-9
    E0, ea0, F0 = oracle_mod.tersoff(SI_T3, *batch[5])
    assert abs(e[5] - E0) <= 1e-9 * abs(E0)
    assert np.abs(ea[40:] - ea0).max() <= 1e-9 and np.abs(F[40:] - F0).max() <= 1e-8
    eng.close()


from conftest import synthetic_tersoff as _synthetic_tersoff  # noqa: E402


def test_tersoff_synthetic_parameter_sets_and_long_rows(oracle_mod):
    """Tersoff beyond GaN.tersoff: three species with m = 3 / lam3 != 0 / n != 1 / extreme beta entries, dense random
    configurations whose rows run from a few to > 16 slots (the four-lanes-per-centre kernel takes rows of <= 16 slots, the
    one-thread form the longer ones -- both in one launch), a sparse one (every row short) and an isolated atom.  Energies,
    per-atom energies and forces against the fp64 oracle."""
    from surface_sampling_amd import backend

    def random_box(n, box, dmin, seed, nt):
        rng = np.random.default_rng(seed)
        pts = []
        while len(pts) < n:
            x = rng.uniform(0, box, 3)
            if all(np.linalg.norm((x - y + box / 2) % box - box / 2) >= dmin for y in pts):
                pts.append(x)
        return rng.integers(0, nt, n).astype(np.int32), np.array(pts), np.eye(3) * box

    for nt, seed in ((3, 1), (2, 2), (4, 3)):
        P = _synthetic_tersoff(nt, seed)
        eng = backend.TersoffEngine(P, device=0)
        batch = []
        for n, box, dmin, sd in ((60, 8.0, 1.7, 10 + seed), (24, 9.0, 2.0, 20 + seed), (90, 8.5, 1.55, 30 + seed)):
            t, x, c = random_box(n, box, dmin, sd, nt)
            batch.append((t, x, c, [1, 1, 1] if n != 24 else [1, 1, 0]))
        batch.append((np.zeros(1, np.int32), np.zeros((1, 3)), np.eye(3) * 20.0, [0, 0, 0]))
        e, ea, F = eng.evaluate_f64(batch)
        st = eng.stats()
        o, longest = 0, 0
        for b, (t, x, c, pbc) in enumerate(batch):
            E0, ea0, F0 = oracle_mod.tersoff(P, t, x, c, pbc)
            n = len(t)
            scale = max(1.0, np.abs(ea0).max())
            assert abs(e[b] - E0) <= 1e-9 * max(1.0, abs(E0)), (nt, b, e[b], E0)
            assert np.abs(ea[o:o + n] - ea0).max() <= 1e-9 * scale
            assert np.abs(F[o:o + n] - F0).max() <= 1e-8 * max(1.0, np.abs(F0).max())
            o += n
        assert st["edges"] / st["atoms"] > 8        # dense: rows well beyond the 16-slot limit exist next to short ones
        eng.close()


def test_calculator_front_end_end_to_end(golden, oracle_mod):
    """The reference-shaped calculator on plain Atoms-like objects (scripts/sample_surface.py:168-183)."""
    import copy

    from surface_sampling_amd.calculators import EnsembleNFFSurface, TersoffSurfCalc

    calc = EnsembleNFFSurface(golden.blobs, device="cuda:0", model_units="kcal/mol", prediction_units="eV",
                              offset_units="atomic")
    calc.set(offset=True, offset_data=golden.offset_data, chem_pots={"Sr": -2, "Ti": 0, "O": 0}, relax_steps=20)
    s = golden.structure("O36Sr12Ti12")
    s.calc = calc
    e = s.get_potential_energy()
    assert e.shape == (1,) and abs(float(e[0]) - (-467.525604)) <= 2e-4
    f = s.get_forces()
    assert f.shape == (len(s), 3)
    se = calc.get_property("surface_energy", atoms=s)
    ref = _oracle(golden, oracle_mod, s)
    from surface_sampling_amd.calculators import surface_energy_from_energy

    want = surface_energy_from_energy(ref["energy"], s.get_chemical_symbols(), calc.chem_pots, golden.offset_data)
    assert abs(float(np.ravel(se)[0]) - want) < 2e-4
    assert "energy" in s.results and "forces_std" in calc.results
    twin = copy.deepcopy(calc)
    assert abs(float(twin.get_potential_energy(s)[0]) - float(e[0])) == 0.0
    outs = calc.calculate_batch([s, golden.structure("O44Sr12Ti16")], want_surface_energy=True)
    assert abs(float(outs[0]["energy"][0]) - float(e[0])) == 0.0 and abs(float(np.ravel(outs[1]["surface_energy"])[0]) - 35.993) < 5e-3
    # without the offset switch the stoichiometric offset is not applied
    raw = EnsembleNFFSurface(golden.blobs, device="cuda:0")
    e_raw = float(raw.get_potential_energy(s)[0])
    table, const = golden.offset_table()
    assert abs(e_raw + table[s.numbers].sum() + const - float(e[0])) < 5e-4

    tcalc = TersoffSurfCalc(golden.tersoff_params, ["Ga", "N"], device="cuda:0")
    g = golden.structure("GaN_3x3_pristine")
    assert abs(tcalc.get_potential_energy(g) - (-144.059)) < 1e-3
    assert tcalc.get_property("per_atom_energies", g).shape == (36,)
    assert abs(tcalc.get_property("surface_energy", g) - tcalc.results["energy"]) == 0.0


def test_error_paths(golden, engine):
    from surface_sampling_amd import backend

    s = golden.structure("SrTiO3_2x2_pristine")
    with pytest.raises(backend.BackendError):
        engine.evaluate([(np.full(len(s), 150, np.int32), s.positions, s.cell, s.pbc)])  # species out of range
    bad = s.positions.copy()
    bad[0, 0] = np.nan
    with pytest.raises(backend.BackendError):
        engine.evaluate([(s.numbers, bad, s.cell, s.pbc)])
    with pytest.raises(backend.BackendError):
        engine.evaluate([(s.numbers, s.positions, np.zeros((3, 3)), s.pbc)])  # singular periodic cell
    # the engine still works afterwards
    ok = engine.evaluate([_arrays(s)])
    assert abs(float(ok["energy"][0]) - (-467.5219)) < 1e-3
    # a single isolated atom (no neighbors at all) is a valid configuration
    lone = engine.evaluate([(np.array([8], np.int32), np.zeros((1, 3)), np.eye(3) * 30.0, [0, 0, 0])])
    assert np.isfinite(lone["energy"][0]) and np.abs(lone["forces"]).max() == 0.0


def test_other_readout_width_runs_the_unfused_reverse_path(golden, oracle_mod):
    """A readout hidden width other than the compiled 64 takes the unfused path: stand-alone readout (painn.hip), reverse
    update kernel reading sbar from memory (MODE 0), stand-alone reverse message MLP (k_msg_mlp_bwd_mfma).  Models: the
    shipped ones with the readout cut to its first 32 hidden units (blob layout: include/vssr_eval.h)."""
    from surface_sampling_amd import backend, structures

    F, H, H2 = 128, 64, 32
    tail = H * F + H + H + 1
    cut = []
    for b in golden.blobs:
        head, t = b[:-tail], b[-tail:]
        w5, b5, w6, b6 = t[:H * F].reshape(H, F), t[H * F:H * F + H], t[H * F + H:H * F + 2 * H], t[-1:]
        cut.append(np.concatenate([head, w5[:H2].ravel(), b5[:H2], w6[:H2], b6]).astype(np.float32))
    table, const = golden.offset_table()
    base = golden.structure("SrTiO3_2x2_pristine")
    chains = [structures.synth_chain(base, c, grid=(4, 4)) for c in (3, 11)]
    eng = backend.PainnEngine(cut, device=0, offset_per_z=table, offset_const=const, hparams={"readout_hidden": H2})
    res = eng.evaluate([_arrays(s) for s in chains])
    e_only = eng.evaluate([_arrays(s) for s in chains], want=backend.WANT_ENERGY)
    eng.close()
    hp = oracle_mod.default_hparams(readout_hidden=H2)
    for b, s in enumerate(chains):
        ref = oracle_mod.ensemble(cut, s.numbers, s.positions, s.cell, s.pbc, 64, table, const, hp=hp)
        a0, a1 = res["cfg_start"][b], res["cfg_start"][b + 1]
        assert abs(float(res["energy"][b]) - ref["energy"]) <= E_TOL
        assert np.abs(res["forces"][a0:a1] - ref["forces"]).max() <= F_TOL
        assert abs(float(res["energy_std"][b]) - ref["energy_std"]) <= STD_TOL
        assert float(e_only["energy"][b]) == float(res["energy"][b])   # energy-only evaluation: same readout kernel


@pytest.mark.gpu
def test_fallback_paths_large_chain_many_species_and_forced_gather(golden, oracle_mod):
    """Paths that the BASELINE workload does not touch: (1) a chain too large for the LDS slices (960 atoms) runs
    the gather kernels, (2) more than 8 species disables the layer-0 factorisation, (3) forcing the gather kernels
    (VSSR_EDGE_IMPL=gather) or the per-edge layer 0 (VSSR_L0_FACTORISE=0) gives the same physics."""
    import os

    from surface_sampling_amd import backend, structures

    table, const = golden.offset_table()
    base = golden.structure("SrTiO3_2x2_pristine")
    big = base.repeat((4, 4, 1))                       # 960 atoms > LDS-slice capacity
    rng = np.random.default_rng(3)
    big.positions += rng.normal(0, 0.03, big.positions.shape)
    small = structures.synth_chain(base, 5, grid=(4, 4))
    eng = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
    res = eng.evaluate([_arrays(big), _arrays(small)])
    for b, s in enumerate((big, small)):
        ref = _oracle(golden, oracle_mod, s)
        a0, a1 = res["cfg_start"][b], res["cfg_start"][b + 1]
        assert abs(float(res["energy_f64"][b]) - ref["energy"]) <= (E_TOL_LARGE if len(s) > 300 else E_TOL), (b, res["energy_f64"][b], ref["energy"])
        assert float(np.float32(res["energy_f64"][b])) == float(res["energy"][b])      # the float32 word = the same value, narrowed
        assert np.abs(res["forces"][a0:a1] - ref["forces"]).max() <= F_TOL
    # (2) nine species: the embedding rows of the extra species are untrained but perfectly valid inputs
    many = small.copy()
    many.numbers[:9] = np.array([1, 6, 7, 9, 13, 14, 20, 26, 29], np.int32)
    r9 = eng.evaluate([_arrays(many)])
    ref9 = _oracle(golden, oracle_mod, many)
    assert abs(float(r9["energy"][0]) - ref9["energy"]) <= 3e-4 and np.abs(r9["forces"] - ref9["forces"]).max() <= 5e-4
    # (2b) six species: still factorised (<= 8), the matrix-pipe layer-0 kernels run with K = 32 x 6 per chunk group
    six = small.copy()
    six.numbers[:3] = np.array([1, 6, 7], np.int32)
    r6 = eng.evaluate([_arrays(six), _arrays(small)])
    ref6 = _oracle(golden, oracle_mod, six)
    a0, a1 = r6["cfg_start"][0], r6["cfg_start"][1]
    assert abs(float(r6["energy"][0]) - ref6["energy"]) <= 3e-4 and np.abs(r6["forces"][a0:a1] - ref6["forces"]).max() <= 5e-4
    # (2c) four and two species: the layer-0 T blocks are accumulated inside k_edge_geom<NZ> (nbr.hip; NZ = 1 .. 4), one
    # batch with a single species count and one that mixes a 2-species chain with the 3-species slab (batch species = 3)
    four = small.copy()
    four.numbers[:2] = np.array([1, 1], np.int32)
    two = small.copy()
    two.numbers[two.numbers == 22] = 38
    for batch in ([four], [two], [two, small]):
        rb = eng.evaluate([_arrays(x) for x in batch])
        for b, x in enumerate(batch):
            refx = _oracle(golden, oracle_mod, x)
            a0, a1 = rb["cfg_start"][b], rb["cfg_start"][b + 1]
            assert abs(float(rb["energy"][b]) - refx["energy"]) <= 3e-4, (len(batch), b)
            assert np.abs(rb["forces"][a0:a1] - refx["forces"]).max() <= 5e-4, (len(batch), b)
    assert eng.profile_read is not None
    ref_small = eng.evaluate([_arrays(small)])
    assert float(ref_small["energy"][0]) == float(r6["energy"][1])   # neighbors in the batch do not change a chain
    eng.close()
    # (3) debug knobs select the other code paths; results agree to fp32 re-association noise
    for var, val in (("VSSR_EDGE_IMPL", "gather"), ("VSSR_L0_FACTORISE", "0")):
        os.environ[var] = val
        try:
            alt = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
            r = alt.evaluate([_arrays(small)])
            alt.close()
        finally:
            del os.environ[var]
        assert abs(float(r["energy"][0]) - float(ref_small["energy"][0])) <= 1e-4, var
        assert np.abs(r["forces"] - ref_small["forces"]).max() <= 1e-4, var


@pytest.mark.gpu
def test_repeatability_stress_many_chains(golden):
    """64 independent chains evaluated repeatedly on fresh engines give bit-identical energies and forces, on the default
    path and on the per-edge layer-0 path.  This is the test that exposed the MFMA source-register hazard (isolated wrong
    messages once in a few hundred chain evaluations; the small parity cases never hit it): a repeat that differs in one
    bit fails."""
    import os

    from surface_sampling_amd import backend, structures

    table, const = golden.offset_table()
    base = golden.structure("SrTiO3_2x2_pristine")
    chains = [_arrays(structures.synth_chain(base, c, grid=(4, 4))) for c in range(64)]
    for env in ({}, {"VSSR_L0_FACTORISE": "0"}):
        os.environ.update(env)
        try:
            ref = None
            for _ in range(2):
                eng = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
                for _ in range(12):
                    r = eng.evaluate(chains)
                    if ref is None:
                        ref = (r["energy"].copy(), r["forces"].copy())
                    assert np.array_equal(r["energy"], ref[0]), env
                    assert np.array_equal(r["forces"], ref[1]), env
                eng.close()
        finally:
            for k in env:
                del os.environ[k]


@pytest.mark.gpu
def test_colliding_atoms_stay_finite_and_match_until_absurd(golden, oracle_mod, engine):
    """Two atoms pushed towards each other: energies of 1e3 .. 1e8 eV still agree with the fp64 oracle to 1e-5 relative
    (the matrix path splits fp32 into fp16 pieces; activations are clamped to the fp16 range before the split), and even
    absurd overlaps return finite numbers, which the relaxation driver flags with the reference's +-1000 guard."""
    from surface_sampling_amd.calculators import EnsembleNFFSurface

    base = golden.structure("SrTiO3_2x2_pristine")
    table, const = golden.offset_table()
    for dmin, check in ((0.6, True), (0.4, True), (0.15, False)):
        s = base.copy()
        v = s.positions[8] - s.positions[7]
        s.positions[8] = s.positions[7] + v / np.linalg.norm(v) * dmin
        r = engine.evaluate([_arrays(s)])
        e, f = float(r["energy"][0]), r["forces"]
        assert np.isfinite(e) and np.isfinite(f).all()
        if check:
            ref = _oracle(golden, oracle_mod, s)
            assert abs(e - ref["energy"]) <= 1e-5 * abs(ref["energy"])
            assert np.abs(f - ref["forces"]).max() <= 1e-5 * np.abs(ref["forces"]).max()
    calc = EnsembleNFFSurface(golden.blobs, device="cuda:0")
    calc.set(offset=True, offset_data=golden.offset_data)
    out = calc.relax_batch([s, base], relax_steps=1)
    assert out[0][3] is True and out[0][2] == calc.ENERGY_THRESHOLD      # energy_oob, clamped energy
    assert out[1][3] is False


@pytest.mark.gpu
def test_degenerate_chains_in_one_batch(golden, oracle_mod, engine):
    """A batch mixing a normal slab with chains that have no edges at all (atoms further apart than the cutoff), a
    single-atom chain, and a non-periodic 4-atom cluster: every kernel must cope with empty CSR rows, tiles of one atom
    and zero-slot chains; results equal the oracle's and do not disturb the neighbours in the batch."""
    from surface_sampling_amd import structures

    Z = structures.ATOMIC_NUMBERS
    box = np.diag([30.0, 30.0, 30.0])
    far = structures.Structure(np.array([Z["Sr"], Z["O"], Z["Ti"]], np.int32),
                               np.array([[2.0, 2.0, 2.0], [12.0, 12.0, 12.0], [22.0, 22.0, 22.0]]), box, np.array([True] * 3))
    one = structures.Structure(np.array([Z["O"]], np.int32), np.array([[1.0, 1.0, 1.0]]), box, np.array([True] * 3))
    cluster = structures.Structure(np.array([Z["Ti"], Z["O"], Z["O"], Z["Sr"]], np.int32),
                                   np.array([[0.0, 0.0, 0.0], [1.9, 0.0, 0.0], [0.0, 1.9, 0.0], [2.2, 2.2, 1.0]]), box,
                                   np.array([False] * 3))
    slab = golden.structure("SrTiO3_2x2_pristine")
    chains = [far, slab, one, cluster]
    res = engine.evaluate([_arrays(s) for s in chains])
    assert np.isfinite(res["energy"]).all() and np.isfinite(res["forces"]).all()
    for b, s in enumerate(chains):
        a0, a1 = res["cfg_start"][b], res["cfg_start"][b + 1]
        ref = _oracle(golden, oracle_mod, s)
        assert abs(float(res["energy"][b]) - ref["energy"]) <= E_TOL, (b, float(res["energy"][b]), ref["energy"])
        assert np.abs(res["forces"][a0:a1] - ref["forces"]).max() <= F_TOL
    assert np.abs(res["forces"][: len(far)]).max() == 0.0           # isolated atoms feel nothing
    alone = engine.evaluate([_arrays(slab)])
    assert float(alone["energy"][0]) == float(res["energy"][1])


@pytest.mark.gpu
def test_odd_chain_sizes_through_the_bundle_walk(golden, oracle_mod, engine):
    """The MFMA edge kernels walk bundles of 4 centres sorted by slot count, dealt to the waves in snake order
    (painn_edge_mfma.hip BundleWalk).  Chain sizes that are not multiples of 4, fewer bundles than waves, very uneven
    coordination numbers (slab fragments in a big periodic box) and all of them mixed in one ragged batch must give the
    oracle's energies and forces and must not depend on what else is in the batch."""
    from surface_sampling_amd import structures

    slab = golden.structure("SrTiO3_2x2_pristine")
    rng = np.random.default_rng(7)
    chains = []
    for n_keep in (57, 41, 13, 6, 5, 3, 2):
        keep = np.sort(rng.choice(len(slab), size=n_keep, replace=False))
        chains.append(structures.Structure(slab.numbers[keep], slab.positions[keep], slab.cell, slab.pbc))
    # a fragment in a large box: surface-like atoms with few neighbours next to fully coordinated ones
    box = slab.cell.copy()
    box[0, 0] *= 3.0
    box[1, 1] *= 3.0
    chains.append(structures.Structure(slab.numbers, slab.positions, box, slab.pbc))
    res = engine.evaluate([_arrays(s) for s in chains])
    assert np.isfinite(res["energy"]).all() and np.isfinite(res["forces"]).all()
    for b, s in enumerate(chains):
        a0, a1 = res["cfg_start"][b], res["cfg_start"][b + 1]
        ref = _oracle(golden, oracle_mod, s)
        assert abs(float(res["energy"][b]) - ref["energy"]) <= E_TOL, (b, len(s), float(res["energy"][b]), ref["energy"])
        assert np.abs(res["forces"][a0:a1] - ref["forces"]).max() <= F_TOL, (b, len(s))
        alone = engine.evaluate([_arrays(s)])
        assert float(alone["energy"][0]) == float(res["energy"][b])
        assert np.array_equal(alone["forces"], res["forces"][a0:a1])


def _bench_chain(golden, c):
    from surface_sampling_amd import structures

    return structures.synth_chain(golden.structure("SrTiO3_2x2_pristine").repeat((2, 2, 1)), c)


def test_bench_batch_subsample_vs_oracle_including_shard_chains(golden, oracle_mod):
    """32 chains of the BASELINE workload -- 16 from the single-GPU batch (configs[3], chains 0..255) and 16 from the blocks the
    other seven GPUs own in configs[4] (chains 256..2047) -- evaluated inside full 256-chain lock-step batches and compared
    chain by chain with the fp64 oracle (the builder-run tools/gpu_full_parity.py covers all 256; stated tolerances)."""
    from surface_sampling_amd import backend

    table, const = golden.offset_table()
    eng = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
    picks = {0: [0, 17, 34, 51, 68, 85, 102, 119, 136, 153, 170, 187, 204, 221, 238, 255],
             1792: [1792 + k for k in (0, 33, 66, 99, 132, 165, 198, 231, 255)],          # rank 7's block
             768: [768 + k for k in (5, 41, 77, 113, 149, 185, 221)]}                     # rank 3's block
    worst_e = worst_f = 0.0
    for first, which in picks.items():
        chains = [_bench_chain(golden, c) for c in range(first, first + 256)]
        res = eng.evaluate([_arrays(s) for s in chains])
        cs = res["cfg_start"]
        assert np.isfinite(res["energy"]).all() and np.isfinite(res["forces"]).all()
        assert res["saturated"].shape == (256,) and not res["saturated"].any()   # the bench batch stays inside the fp16-split range
        for c in which:
            b = c - first
            ref = _oracle(golden, oracle_mod, chains[b])
            de = abs(float(res["energy"][b]) - ref["energy"])
            df = float(np.abs(res["forces"][cs[b]:cs[b + 1]] - ref["forces"]).max())
            assert de <= E_TOL and df <= F_TOL, (c, de, df)
            assert abs(float(res["energy_std"][b]) - ref["energy_std"]) <= STD_TOL
            worst_e, worst_f = max(worst_e, de), max(worst_f, df)
    print(f"32-chain subsample: max |dE| {worst_e:.2e} eV, max |dF| {worst_f:.2e} eV/A")
    eng.close()


def test_whole_bench_batch_against_the_committed_fp64_vectors(golden):
    """ALL 256 chains of the benchmark batch (BASELINE configs[3]) against the fp64 oracle's committed answers
    (tests/golden/bench_batch_fp64.npz, tools/make_bench_golden.py) at the stated tolerances, and the device's deviation printed
    next to that of the oracle's own fp32 mode: what the fp16-split arithmetic costs relative to plain fp32 of the same algorithm."""
    import hashlib

    from surface_sampling_amd import backend

    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "bench_batch_fp64.npz"))
    chains = [_bench_chain(golden, c) for c in range(256)]
    h = hashlib.sha256()
    for s in chains:
        h.update(np.ascontiguousarray(s.numbers, dtype=np.int32).tobytes())
        h.update(np.ascontiguousarray(s.positions, dtype=np.float64).tobytes())
        h.update(np.ascontiguousarray(s.cell, dtype=np.float64).tobytes())
    assert h.hexdigest() == str(G["inputs_sha256"]), "the benchmark batch is not the one the vectors were generated from"
    table, const = golden.offset_table()
    eng = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
    res = eng.evaluate([_arrays(s) for s in chains])
    eng.close()
    cs = res["cfg_start"]
    assert np.array_equal(np.asarray(cs, dtype=np.int64), G["cfg_start"])
    assert not res["saturated"].any()
    dE = np.abs(res["energy"].astype(np.float64) - G["energy"])
    dE64 = np.abs(res["energy_f64"] - G["energy"])
    dS = np.abs(res["energy_std_f64"] - G["energy_std"])
    dF = np.abs(res["forces"].astype(np.float64) - G["forces"])
    dF_chain = np.maximum.reduceat(dF.max(axis=1), cs[:-1])
    # the oracle's fp32 mode against its fp64 mode on the same chains (committed with the vectors)
    oE = np.abs(G["energy_fp32mode"] - G["energy"])
    oS = np.abs(G["energy_std_fp32mode"] - G["energy_std"])
    oF = np.abs(G["forces_fp32mode"].astype(np.float64) - G["forces"])
    print(f"256 chains, {int(cs[-1])} atoms, deviation from the fp64 oracle (max | mean of per-chain max):")
    print(f"   device        |dE| f32 word {dE.max():.2e} | {dE.mean():.2e}   f64 word {dE64.max():.2e} | {dE64.mean():.2e} eV   "
          f"|dF| {dF.max():.2e} | {dF_chain.mean():.2e} eV/A   |dEstd| {dS.max():.2e}")
    print(f"   oracle fp32   |dE| {oE.max():.2e} | {oE.mean():.2e} eV   |dF| {oF.max():.2e} | "
          f"{np.maximum.reduceat(oF.max(axis=1), cs[:-1]).mean():.2e} eV/A   |dEstd| {oS.max():.2e}")
    worst = int(np.argmax(dF_chain))
    assert dE.max() <= E_TOL, (int(np.argmax(dE)), float(dE.max()))
    assert dE64.max() <= E_TOL
    assert dF.max() <= F_TOL, (worst, float(dF_chain[worst]))
    assert dS.max() <= STD_TOL


def test_repeatability_on_the_bench_batch(golden):
    """The MFMA source-register hazard (profiles/r01/NOTES_mfma_hazards.md) showed up as isolated wrong values once in
    ~1e3 chain-layer instances: the B = 256 bench batch is evaluated repeatedly on two engines, every repeat bit-identical."""
    from surface_sampling_amd import backend

    table, const = golden.offset_table()
    chains = [_arrays(_bench_chain(golden, c)) for c in range(256)]
    ref = None
    for _ in range(2):
        eng = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
        eng.upload(chains)
        for _ in range(6):
            eng.run()
            r = eng.download()
            if ref is None:
                ref = {k: r[k].copy() for k in ("energy", "forces", "energy_std", "forces_std")}
            for k, v in ref.items():
                assert np.array_equal(r[k], v), k
        eng.close()


@pytest.mark.parametrize("what,factor", [("embed", 1e-4), ("embed", 1e-2), ("embed", 1.5), ("filter", 1e-4), ("filter", 1e-2),
                                         ("filter", 1.5), ("embed", 2.0)])
def test_precision_envelope_of_the_fp16_split(golden, oracle_mod, what, factor):
    """Every contraction runs as an fp16 2-way split (x = h + l, 22 mantissa bits, fp16 exponent range, activations clamped
    to +-65504).  The envelope, measured (tools/gpu_envelope.py): with the embedding table or the radial-filter weights of all
    layers scaled by 1e-4 .. 1.5 the device stays at fp32-level RELATIVE accuracy against the fp64 oracle (energy <= 2.1e-7 of
    the largest model energy, forces <= 3.1e-5 of the largest force component).  Scaled by 2 the models themselves leave
    their physical regime (model energies ~ -2.7e4, forces ~ 5e5 kcal/mol/A; by 3 the fp64 energies are ~1e15) and activations
    reach the fp16 range limit: results stay finite but are 4-8 % off.  The shipped weights peak at |activation| ~ 160
    (tools/gpu_ranges.py), a factor ~400 below the clamp."""
    from surface_sampling_amd import backend, checkpoint, structures

    s = structures.synth_chain(golden.structure("SrTiO3_2x2_pristine"), 3, grid=(4, 4))
    blobs = []
    for b in golden.blobs:
        b = b.copy()
        f = checkpoint.blob_to_fields(b)
        if what == "embed":
            f["embed"] *= factor
        else:
            for l in range(3):
                f[f"msg{l}.Wd"] *= factor
                f[f"msg{l}.bd"] *= factor
        blobs.append(b)
    eng = backend.PainnEngine(blobs, device=0, model_units_per_ev=1.0)
    r = eng.evaluate([_arrays(s)])
    eng.close()
    assert np.isfinite(r["energy"]).all() and np.isfinite(r["forces"]).all()
    if factor >= 2.0:
        # outside the envelope: finite, no accuracy claim -- and the caller is TOLD (vssr_batch_saturated): a clamp fired
        assert r["saturated"].all(), "values beyond +-65504 were clamped without raising the saturation flag"
        return
    assert not r["saturated"].any(), "saturation flag raised inside the envelope"
    o = oracle_mod.ensemble(blobs, s.numbers, s.positions, s.cell, s.pbc, 64, None, 0.0, 1.0)
    rel_e = abs(float(r["energy"][0]) - o["energy"]) / np.abs(o["energy_models"]).max()
    rel_f = np.abs(r["forces"] - o["forces"]).max() / np.abs(o["forces"]).max()
    assert rel_e <= 1e-6 and rel_f <= 1e-4, (what, factor, rel_e, rel_f)


# ---- slice-width classes of the neighbor-sum kernels (no batch-wide cliff at 351 atoms) -------------------
def _sized_chains(golden):
    """~260 (16-feature slices, residual in LDS), ~370 (16-feature slices, residual from memory), ~495 and ~735 (8-feature
    slices) and ~975 atoms (gather kernels)."""
    from surface_sampling_amd import structures

    s60, s80 = golden.structure("SrTiO3_2x2_pristine"), golden.structure("SrTiO3_2x2x4_pristine")
    return [structures.synth_chain(s60.repeat((2, 2, 1)), 4), structures.synth_chain(s60.repeat((3, 2, 1)), 2, grid=(12, 8)),
            structures.synth_chain(s80.repeat((3, 2, 1)), 7, grid=(12, 8)), structures.synth_chain(s80.repeat((3, 3, 1)), 5, grid=(12, 12)),
            structures.synth_chain(s80.repeat((4, 3, 1)), 6, grid=(16, 12))]


def test_mixed_chain_sizes_take_their_own_path(golden, oracle_mod, engine):
    """One batch with chains of 260 / 370 / 495 / 735 / 975 atoms: every chain is served by the kernels its own size selects
    (forward: 16-feature slices up to 350 atoms, the same with the scalar residual read from memory up to 405, beyond that the
    16-feature kernel in 2 .. 4 passes over sub-ranges of the chain's neighbors; reverse: 16-feature slices up to 557 atoms, then in
    2 .. 3 passes; gather kernels beyond 1 462 atoms), within the stated tolerances of the fp64 oracle, and BIT-IDENTICAL to the same
    chain evaluated alone or in another order -- one large chain does not send its whole batch to a slower path."""
    chains = _sized_chains(golden)
    sizes = [len(c) for c in chains]
    assert sizes[0] <= 350 < sizes[1] <= 405 < sizes[2] < sizes[3] <= 787 < sizes[4] <= 1127, sizes
    res = engine.evaluate([_arrays(c) for c in chains])
    assert not res["saturated"].any()
    cs = res["cfg_start"]
    for b, c in enumerate(chains):
        ref = _oracle(golden, oracle_mod, c)
        # on the fp64 output word (vssr_batch_energy_f64): the float32 word alone is good for 1.2e-4 (-1.9 keV) .. 4.9e-4 eV (-7.5 keV)
        tol_e = E_TOL if len(c) <= 300 else E_TOL_LARGE
        assert abs(float(res["energy_f64"][b]) - ref["energy"]) <= tol_e, (b, len(c), float(res["energy_f64"][b]), ref["energy"])
        assert float(np.float32(res["energy_f64"][b])) == float(res["energy"][b])
        assert np.abs(res["forces"][cs[b]:cs[b + 1]] - ref["forces"]).max() <= F_TOL, (b, len(c))
        assert abs(float(res["energy_std"][b]) - ref["energy_std"]) <= STD_TOL
        alone = engine.evaluate([_arrays(c)])
        assert float(alone["energy"][0]) == float(res["energy"][b]), (b, len(c))
        assert np.array_equal(alone["forces"], res["forces"][cs[b]:cs[b + 1]])
        assert np.array_equal(alone["forces_std"], res["forces_std"][cs[b]:cs[b + 1]])
    rev = engine.evaluate([_arrays(c) for c in chains[::-1]])
    assert np.array_equal(rev["energy"][::-1], res["energy"])


@pytest.mark.parametrize("knobs", [{"VSSR_EDGE_FS16_MAX": "0"}, {"VSSR_EDGE_FS16_MAX": "0", "VSSR_EDGE_FS8_MAX": "0"},
                                   {"VSSR_EDGE_FS16_MAX": "0", "VSSR_EDGE_FS8_MAX": "0", "VSSR_EDGE_FWD_2PASS": "8"},
                                   {"VSSR_EDGE_FS16_MAX": "0", "VSSR_EDGE_FS8_MAX": "0", "VSSR_EDGE_FWD_2PASS": "0"},
                                   {"VSSR_EDGE_FS16_MAX": "0", "VSSR_EDGE_FS8_MAX": "0", "VSSR_EDGE_BWD_MPASS": "2", "VSSR_EDGE_SUB_CHUNK": "100"},
                                   {"VSSR_EDGE_FS16_MAX": "0", "VSSR_EDGE_FS8_MAX": "0", "VSSR_EDGE_BWD_MPASS": "2", "VSSR_EDGE_SUB_CHUNK": "70",
                                    "VSSR_EDGE_FWD_2PASS": "8"}],
                         ids=["8-feature", "4-feature-class-forward-16-feature-multi-pass", "4-feature-class-forward-8-feature-multi-pass",
                              "4-feature-both-ways", "multi-pass-both-ways-ranges-of-100", "multi-pass-both-ways-ranges-of-70-forward-8-feature"])
def test_narrow_feature_slices_match_the_oracle_on_the_small_structures(golden, oracle_mod, monkeypatch, knobs):
    """The 8- and the 4-feature-slice kernels on inputs every other test sends through the 16-feature ones (VSSR_EDGE_FS16_MAX=0
    moves every chain to the next class, VSSR_EDGE_FS8_MAX=0 one further: the class of 788 .. 1 462-atom chains -- reverse pass on
    4-feature slices, forward pass as several launches of the 16- (or, VSSR_EDGE_FWD_2PASS=8, 8-) feature kernel over sub-ranges of
    the chain's neighbors (round 5), or with VSSR_EDGE_FWD_2PASS=0 on 4-feature slices as in round 4; VSSR_EDGE_BWD_MPASS=2 +
    VSSR_EDGE_SUB_CHUNK=n: forward AND reverse pass in several passes over neighbor ranges of n atoms -- what chains of more than
    557 atoms take in the reverse pass -- with up to four ranges on these 76 .. 270-atom structures): reference KAT structure, per-layer intermediates, a ragged batch; results agree with the
    default path to fp32 rounding (different summation tree inside a slot quad is NOT involved: same order, other slices)."""
    from surface_sampling_amd import backend, structures

    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    table, const = golden.offset_table()
    eng8 = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
    for k in knobs:
        monkeypatch.delenv(k)
    eng16 = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
    big = golden.structure("SrTiO3_2x2_pristine").repeat((2, 2, 1))
    chains = [golden.structure("O44Sr12Ti16"), structures.synth_chain(big, 3), structures.synth_chain(big, 17),
              structures.synth_chain(golden.structure("SrTiO3_2x2_pristine"), 2, grid=(4, 4))]
    r8 = eng8.evaluate([_arrays(c) for c in chains])
    r16 = eng16.evaluate([_arrays(c) for c in chains])
    cs = r8["cfg_start"]
    for b, c in enumerate(chains):
        ref = _oracle(golden, oracle_mod, c)
        assert abs(float(r8["energy"][b]) - ref["energy"]) <= E_TOL
        assert np.abs(r8["forces"][cs[b]:cs[b + 1]] - ref["forces"]).max() <= F_TOL
    assert np.abs(r8["energy"] - r16["energy"]).max() <= 1e-4 and np.abs(r8["forces"] - r16["forces"]).max() <= 2e-5
    # per-layer activations of the last chain, model 1
    s = chains[-1]
    eng8.evaluate([_arrays(s)])
    _, _, d = oracle_mod.painn(golden.blobs[1], s.numbers, s.positions, s.cell, s.pbc, 64, dump=True)
    n = len(s)
    for l in (1, 2):
        for name, shape in (("s_msg", (n, 128)), ("v_msg", (n, 3, 128))):
            got = eng8.debug_read(f"{name}{l}", 1).reshape(shape).astype(np.float64)
            want = d[name][l]
            assert np.abs(got - want).max() / max(1.0, np.abs(want).max()) < 2e-5, (name, l)
    # determinism of the new instantiation
    again = eng8.evaluate([_arrays(c) for c in chains])
    assert np.array_equal(again["energy"], r8["energy"]) and np.array_equal(again["forces"], r8["forces"])
    eng8.close()
    eng16.close()


def test_stored_forward_intermediates_give_identical_results(golden, monkeypatch):
    """VSSR_UPD_SAVE=1: update_fwd stores U v, V v, the gate pre-activation and the gates; the reverse update kernel loads
    them instead of recomputing three GEMMs.  Energies are bit-identical (the forward pass is untouched); forces agree to fp32
    rounding -- the recomputing reverse kernel sums the gate MLP's first layer in one K = 256 pass, the forward kernel in two
    K = 128 halves, so the two paths differentiate forward values that differ in the last bit."""
    from surface_sampling_amd import backend, structures

    table, const = golden.offset_table()
    big = golden.structure("SrTiO3_2x2_pristine").repeat((2, 2, 1))
    chains = [_arrays(structures.synth_chain(big, c)) for c in (1, 9, 20)] + [_arrays(golden.structure("O40Sr16Ti12"))]
    monkeypatch.setenv("VSSR_UPD_SAVE", "1")
    e1 = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
    monkeypatch.setenv("VSSR_UPD_SAVE", "0")
    e0 = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
    r1, r0 = e1.evaluate(chains), e0.evaluate(chains)
    for k in ("energy", "energy_std", "energy_models"):
        assert np.array_equal(r1[k], r0[k]), k
    assert np.abs(r1["forces"] - r0["forces"]).max() <= 2e-5 and np.abs(r1["forces_std"] - r0["forces_std"]).max() <= 2e-5   # (a few ulp of 30 eV/A forces)
    e_only = e1.evaluate(chains, want=backend.WANT_ENERGY)          # energy-only runs store nothing
    assert np.array_equal(e_only["energy"], r0["energy"])
    e1.close()
    e0.close()


def test_repeatability_of_every_neighbor_sum_path():
    """Short form of tools/gpu_stress_classes.py (the builder-run soak: 9 configurations incl. the 4-feature slices of round 4 and the
    two-pass forward form of round 5,
    profiles/r03/NOTES_soak.md): every slice width / workgroup width of the edge kernels, three fresh engines each, every
    evaluation bit-identical to the first (MFMA operand hazards show up as isolated run-to-run differences)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NCHAIN="16", REPS="6")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_stress_classes.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("mismatches 0") == 10, r.stdout


def test_large_chain_takes_four_feature_slices_in_both_directions(golden, oracle_mod, engine):
    """A 1 2xx-atom chain (5 x 3 tiling of the 80-atom slab + adsorbates): forward pass in four, reverse pass in three passes of the
    16-feature kernels over neighbor sub-ranges by its own size (round 4: 4-feature slices in both directions -- the class is still
    called that).  Forces within the stated 2e-4 eV/A of the fp64 oracle;
    the energy (-3.8 keV) within 1e-4 eV on the fp64 output word (the float32 word's spacing there is 2.4e-4 eV: it is the
    same value narrowed, at most half a spacing away); bit-identical when evaluated again and next to a small chain."""
    from surface_sampling_amd import structures

    s80 = golden.structure("SrTiO3_2x2x4_pristine")
    big = structures.synth_chain(s80.repeat((5, 3, 1)), 3, grid=(20, 12))
    small = structures.synth_chain(golden.structure("SrTiO3_2x2_pristine").repeat((2, 2, 1)), 8)
    assert 1127 < len(big) <= 1462
    res = engine.evaluate([_arrays(big)])
    ref = _oracle(golden, oracle_mod, big)
    assert abs(float(res["energy_f64"][0]) - ref["energy"]) <= E_TOL_LARGE, (float(res["energy_f64"][0]), ref["energy"])
    assert float(np.float32(res["energy_f64"][0])) == float(res["energy"][0]) and abs(float(res["energy"][0]) - ref["energy"]) <= 1.3e-4 + E_TOL_LARGE
    assert np.abs(res["forces"] - ref["forces"]).max() <= F_TOL
    assert np.abs(res["forces_std"] - ref["forces_std"]).max() <= STD_TOL and not res["saturated"].any()
    both = engine.evaluate([_arrays(small), _arrays(big)])
    cs = both["cfg_start"]
    assert float(both["energy"][1]) == float(res["energy"][0]) and np.array_equal(both["forces"][cs[1]:cs[2]], res["forces"])
    st, _ = engine.stress()
    assert np.isfinite(st).all()


def test_lammps_surf_calc_runs_the_gan_tutorial_flow(tmp_path, golden, oracle_mod):
    """BASELINE configs[1] through the reference's own front end (tutorials/GaN_0001.ipynb): ``LAMMPSSurfCalc()`` configured
    with ``set(**calc_settings)`` (run directory with lammps_config.json + templates), static energy -144.059 eV
    (``tutorials/GaN_0001.ipynb:228``), ``run_lammps_energy`` / ``run_lammps_opt`` tuples, the template's bulk group held."""
    from test_host_logic import _gan_run_dir
    from surface_sampling_amd import calculators as calcs

    rd = _gan_run_dir(tmp_path, golden)
    calc = calcs.LAMMPSSurfCalc(device="cuda:0")
    calc.set(calc_name="LAMMPS", optimizer="LAMMPS", chem_pots={"Ga": 5}, relax_atoms=True, relax_steps=100, run_dir=rd)
    s = golden.structure("GaN_3x3_pristine")
    e = calc.get_potential_energy(s)
    assert abs(e - golden.kat["tersoff"]["energy"]) <= golden.kat["tolerance"]["tersoff_energy_abs"]
    same, e2, pe = calc.run_lammps_energy(s, run_dir=rd)
    assert same is s and e2 == e and pe.shape == (36,) and abs(pe.sum() - e) <= 1e-9
    assert calc.get_surface_energy(s) == e and calc.get_property("per_atom_energies", s).shape == (36,)
    # 12 Ga adatoms on top (the tutorial's canonical composition), relaxed with the template's minimiser: ids <= bulk_index held
    rng = np.random.default_rng(3)
    ads = s.copy()
    ads.numbers = np.r_[s.numbers, np.full(12, 31)].astype(np.int32)
    ads.positions = np.r_[s.positions, s.positions[18:30] + [0.3, 0.2, 1.9] + rng.normal(0, 0.03, (12, 3))]
    e_static = calc.get_potential_energy(ads)
    relaxed, e_rel, pe_rel = calc.run_lammps_opt(ads, run_dir=calc.run_dir)
    assert np.array_equal(relaxed.positions[:36], ads.positions[:36])           # group bulk id <= 36, setforce 0
    assert np.abs(relaxed.positions[36:] - ads.positions[36:]).max() > 1e-3 and e_rel < e_static - 1e-3
    assert pe_rel.shape == (48,) and calc.last_opt["optimizer"] == "CG"
    types = np.array([0 if z == 31 else 1 for z in relaxed.numbers], np.int32)
    E0, _, _ = oracle_mod.tersoff(golden.tersoff_params, types, relaxed.positions, relaxed.cell, [1, 1, 1])
    assert abs(e_rel - E0) <= 1e-9 * abs(E0)
    assert calc.get_property("relaxed_energy", ads) == pytest.approx(e_rel, abs=1e-9)


def test_row_scan_across_tile_boundaries(golden):
    """The CSR row offsets come from a two-launch scan over tiles of 4 096 atoms (``k_scan_tiles`` + ``k_scan_rows``): batches whose
    atom count sits on, just below and just above one and two tile boundaries give every chain exactly the energy, per-atom
    energies and forces it has when evaluated alone, and the batch's edge count is the sum of the chains'."""
    from surface_sampling_amd import backend

    eng = backend.TersoffEngine(golden.tersoff_params, device=0)
    s = golden.structure("GaN_3x3_pristine")
    types = np.array([0 if z == 31 else 1 for z in s.numbers], np.int32)
    rng = np.random.default_rng(5)
    slab = (types, s.positions + rng.normal(0, 0.02, s.positions.shape), s.cell, [1, 1, 1])
    dimer = (np.array([0, 1], np.int32), np.array([[0.0, 0.0, 0.0], [1.9, 0.1, 0.0]]), np.eye(3) * 15.0, [0, 0, 0])
    lone = (np.zeros(1, np.int32), np.zeros((1, 3)), np.eye(3) * 15.0, [0, 0, 0])
    alone = {}
    for name, st in (("slab", slab), ("dimer", dimer), ("lone", lone)):
        e, ea, f = eng.evaluate_f64([st])
        alone[name] = (e[0], ea.copy(), f.copy(), eng.stats()["edges"])
    for total in (4095, 4096, 4097, 8191, 8192, 8193, 12289):
        n_slab = total // 36 - 1
        rest = total - 36 * n_slab                    # 36 .. 71 atoms as dimers + single atoms, spread through the batch
        names = ["slab"] * n_slab + ["dimer"] * (rest // 2) + ["lone"] * (rest % 2)
        order = rng.permutation(len(names))
        names = [names[k] for k in order]
        batch = [{"slab": slab, "dimer": dimer, "lone": lone}[k] for k in names]
        assert sum(len(b[0]) for b in batch) == total
        e, ea, f = eng.evaluate_f64(batch)
        assert eng.stats()["atoms"] == total and eng.stats()["edges"] == sum(alone[k][3] for k in names)
        o = 0
        for b, k in enumerate(names):
            n = len(batch[b][0])
            assert e[b] == alone[k][0], (total, b, k)
            assert np.array_equal(ea[o:o + n], alone[k][1]) and np.array_equal(f[o:o + n], alone[k][2])
            o += n
    eng.close()


def test_lammps_surf_calc_packed_path_holds_the_bulk_group(tmp_path, golden):
    """``LAMMPSSurfCalc.evaluate_packed`` without a mask holds the template's bulk group (ids <= bulk_index) like ``relax_batch``
    without ``fixed_indices``: batched MC through the packed path and through the per-slab path end in identical states, and the
    first 36 atoms of every relaxed slab are where the pristine slab has them."""
    from test_host_logic import _gan_run_dir
    from surface_sampling_amd import calculators as calcs
    from surface_sampling_amd import mc

    rd = _gan_run_dir(tmp_path, golden)
    g = golden.structure("GaN_3x3_pristine")
    ztop = g.positions[:, 2].max()
    coords = np.array([(i + 0.5) / 3 * g.cell[0] + (j + 0.5) / 3 * g.cell[1] for i in range(3) for j in range(3)], float)
    coords[:, 2] = ztop + 1.8
    runs = []
    for fast in (True, False):
        calc = calcs.LAMMPSSurfCalc(device="cuda:0")
        calc.set(calc_name="LAMMPS", optimizer="LAMMPS", chem_pots={"Ga": 5}, relax_atoms=True, relax_steps=20, run_dir=rd)
        ens = mc.ChainEnsemble(g, coords, ("Ga", "N"), 5, calc, seed=8, relax=True, relax_steps=20, temperature=0.4, optimizer="LAMMPS")
        ens.fast_path = fast
        ens.initialize()
        acc = np.stack([ens.step_semigrand() for _ in range(4)])
        runs.append((acc, ens))
    (a, ea), (b, eb) = runs
    assert np.array_equal(a, b) and np.array_equal(ea.state.species, eb.state.species) and np.array_equal(ea.state.energy, eb.state.energy)
    assert a.any() and (ea.num_adsorbates() > 0).any()
    for k in range(5):
        assert np.array_equal(ea.relaxed[k].positions, eb.relaxed[k].positions)
        assert np.array_equal(ea.relaxed[k].positions[:36], g.positions)               # group bulk id <= 36: setforce 0
        if len(ea.relaxed[k]) > 36:
            assert np.abs(ea.relaxed[k].positions[36:] - ea.structure(k).positions[36:]).max() > 1e-4   # the adsorbates did move
