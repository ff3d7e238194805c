"""The one-shot C-ABI entry points, driven through raw ctypes exactly as INTEGRATION.md shows a maintainer of the
reference would bind them (no helper classes of this package on the call path except the struct definitions)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def test_vssr_eval_and_eval_batch_through_raw_ctypes(golden):
    from surface_sampling_amd import backend

    lib = backend.load_library()
    table, const = golden.offset_table()
    engine = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
    blobs = [np.ascontiguousarray(b, np.float32) for b in golden.blobs]
    ptrs = (C.POINTER(C.c_float) * len(blobs))(*[_f32(b) for b in blobs])
    tab = np.ascontiguousarray(table, np.float64)
    cfg = backend.PainnConfig(C.sizeof(backend.PainnConfig), 0, len(blobs), ptrs, blobs[0].size, 128, 20, 3, 100, 64, 5.0, 1,
                              12, 1.5, 23.0605, tab.ctypes.data_as(C.POINTER(C.c_double)), float(const))
    h = C.c_void_p()
    assert lib.vssr_create(C.byref(cfg), C.byref(h)) == 0, lib.vssr_last_error(None)
    try:
        structs = [golden.structure("O36Sr12Ti12"), golden.structure("O44Sr12Ti16")]
        want = backend.WANT_ENERGY | backend.WANT_FORCES | backend.WANT_STD
        ref = engine.evaluate([(s.numbers, s.positions, s.cell, s.pbc) for s in structs])
        # ---- vssr_eval: one configuration ----------------------------------------------------------------------------------
        s = structs[0]
        Z = np.ascontiguousarray(s.numbers, np.int32)
        pos = np.ascontiguousarray(s.positions, np.float64)
        cell = np.ascontiguousarray(s.cell, np.float64)
        pbc = np.ascontiguousarray(s.pbc, np.uint8)
        E, Es = np.zeros(1, np.float32), np.zeros(1, np.float32)
        F, Fs = np.zeros((len(Z), 3), np.float32), np.zeros((len(Z), 3), np.float32)
        out = backend.Out(_f32(E), _f32(Es), _f32(F), _f32(Fs), None, None)
        rc = lib.vssr_eval(h, len(Z), Z.ctypes.data_as(C.POINTER(C.c_int32)), pos.ctypes.data_as(C.POINTER(C.c_double)),
                           cell.ctypes.data_as(C.POINTER(C.c_double)), pbc.ctypes.data_as(C.POINTER(C.c_uint8)), want,
                           C.byref(out))
        assert rc == 0, lib.vssr_last_error(h)
        assert abs(float(E[0]) - (-467.525604)) <= 2e-4                  # reference notebook value
        assert float(E[0]) == float(ref["energy"][0]) and np.array_equal(F, ref["forces"][: len(Z)])
        assert float(Es[0]) == float(ref["energy_std"][0])
        # ---- vssr_eval_batch: two configurations, concatenated arrays ------------------------------------------------------------
        n_atoms = np.array([len(x) for x in structs], np.int32)
        Zb = np.ascontiguousarray(np.concatenate([x.numbers for x in structs]), np.int32)
        posb = np.ascontiguousarray(np.concatenate([x.positions for x in structs]), np.float64)
        cellb = np.ascontiguousarray(np.stack([x.cell for x in structs]), np.float64)
        pbcb = np.ascontiguousarray(np.stack([x.pbc for x in structs]), np.uint8)
        Eb, Esb = np.zeros(2, np.float32), np.zeros(2, np.float32)
        Fb, Fsb = np.zeros((len(Zb), 3), np.float32), np.zeros((len(Zb), 3), np.float32)
        Em = np.zeros((2, len(blobs)), np.float32)
        outb = backend.Out(_f32(Eb), _f32(Esb), _f32(Fb), _f32(Fsb), _f32(Em), None)
        rc = lib.vssr_eval_batch(h, 2, n_atoms.ctypes.data_as(C.POINTER(C.c_int32)), Zb.ctypes.data_as(C.POINTER(C.c_int32)),
                                 posb.ctypes.data_as(C.POINTER(C.c_double)), cellb.ctypes.data_as(C.POINTER(C.c_double)),
                                 pbcb.ctypes.data_as(C.POINTER(C.c_uint8)), want | backend.WANT_PER_MODEL, C.byref(outb))
        assert rc == 0, lib.vssr_last_error(h)
        assert np.array_equal(Eb, ref["energy"]) and np.array_equal(Fb, ref["forces"]) and np.array_equal(Fsb, ref["forces_std"])
        assert np.allclose(Em.mean(axis=1), Eb, atol=2e-4) and Em.std() > 0
        # ---- vssr_batch_energy_f64: the same energies without the float32 result word, raw pointers -------------------------------
        e64, s64, m64 = np.zeros(2), np.zeros(2), np.zeros((2, len(blobs)))
        dp = C.POINTER(C.c_double)
        rc = lib.vssr_batch_energy_f64(h, e64.ctypes.data_as(dp), s64.ctypes.data_as(dp), m64.ctypes.data_as(dp))
        assert rc == 0, lib.vssr_last_error(h)
        assert np.array_equal(e64.astype(np.float32), Eb) and np.array_equal(s64.astype(np.float32), Esb)
        assert np.array_equal(m64.astype(np.float32), Em) and (e64 != Eb.astype(np.float64)).any()     # more digits than float32 holds
        assert np.allclose(m64.mean(axis=1), e64, rtol=0, atol=1e-9) and np.allclose(m64.std(axis=1), s64, rtol=0, atol=1e-9)
        assert lib.vssr_batch_energy_f64(h, None, None, None) == 0        # every output optional
        d_e, d_s = C.c_void_p(), C.c_void_p()
        assert lib.vssr_batch_device_results_f64(h, C.byref(d_e), C.byref(d_s)) == 0 and d_e.value and d_s.value == d_e.value + 16
        # ---- vssr_batch_stress: the virial of the evaluation that has just run (two chains), raw pointers -------------------------
        st, sd = np.zeros((2, 6)), np.zeros((2, 6))
        rc = lib.vssr_batch_stress(h, st.ctypes.data_as(C.POINTER(C.c_double)), sd.ctypes.data_as(C.POINTER(C.c_double)))
        assert rc == 0, lib.vssr_last_error(h)
        want_st, want_sd = engine.stress()                      # the helper class on the same resident batch
        assert np.array_equal(st, want_st) and np.array_equal(sd, want_sd) and np.isfinite(st).all() and np.abs(st).max() > 0
        assert lib.vssr_batch_stress(h, None, None) == 0        # both outputs optional
        # ---- error reporting: bad argument -> negative status + message ---------------------------------------------------------
        rc = lib.vssr_eval(h, -1, Z.ctypes.data_as(C.POINTER(C.c_int32)), pos.ctypes.data_as(C.POINTER(C.c_double)),
                           cell.ctypes.data_as(C.POINTER(C.c_double)), pbc.ctypes.data_as(C.POINTER(C.c_uint8)), want,
                           C.byref(out))
        assert rc < 0 and lib.vssr_last_error(h)
    finally:
        lib.vssr_destroy(h)
        engine.close()


def test_resident_batch_set_positions_and_profiler(golden):
    """The split API a relaxation loop uses: upload once, then set_positions / run / download per step — results equal a
    fresh upload of the moved structure; the per-class profiler counts the launches of each step."""
    from surface_sampling_amd import backend

    table, const = golden.offset_table()
    eng = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
    try:
        s0 = golden.structure("SrTiO3_2x2_pristine")
        s1 = golden.structure("O40Sr16Ti12")
        eng.upload([(s.numbers, s.positions, s.cell, s.pbc) for s in (s0, s1)])
        eng.profile_enable(True)
        eng.profile_reset()
        rng = np.random.default_rng(0)
        moved = [s.positions + rng.normal(0, 0.02, s.positions.shape) for s in (s0, s1)]
        eng.set_positions(np.concatenate(moved))
        eng.run()
        got = eng.download()
        prof = eng.profile_read()
        assert prof["neighbor_list"]["launches"] == 1 and prof["edge_message_bwd"]["launches"] == 2
        assert prof["update_bwd"]["launches"] == 3 and all(v["total_ms"] >= 0 for v in prof.values())
        fresh = eng.evaluate([(s.numbers, p, s.cell, s.pbc) for s, p in zip((s0, s1), moved)])
        assert np.array_equal(got["energy"], fresh["energy"]) and np.array_equal(got["forces"], fresh["forces"])
        assert eng.stats()["atoms"] == len(s0) + len(s1)
    finally:
        eng.close()


def test_tersoff_create_from_text_and_state_errors(golden, oracle_mod):
    """vssr_tersoff_create_from_text (potential text + species, parsed by the library) gives the same
    energies as the array form; call-sequence errors of the resident-batch API are reported, not served from stale buffers."""
    from surface_sampling_amd import backend

    sp = golden.tersoff["species"]
    P = golden.tersoff_params
    lines = ["# GaN test potential, entries rebuilt from the golden parameter table", ""]
    for i, a in enumerate(sp):
        for j, b in enumerate(sp):
            for k, c in enumerate(sp):
                v = ["%.17g" % x for x in P[i, j, k]]
                lines += [f"{a} {b} {c}  " + " ".join(v[:7]) + "   # three-body part", "        " + " ".join(v[7:]), ""]
    lines.append("Si Si Si 3.0 1.0 0.0 100390 16.217 -0.59825 0.78734 1.1e-6 1.7322 471.18 2.85 0.15 2.4799 1830.8")
    text = "\n".join(lines)
    g = golden.structure("GaN_3x3_pristine")
    types = np.array([0 if z == 31 else 1 for z in g.numbers], np.int32)
    pack = [(types, g.positions, g.cell, np.ones(3, np.uint8))]
    e_text = backend.TersoffEngine(text, device=0, species=sp)
    e_arr = backend.TersoffEngine(P, device=0)
    a, b = e_text.evaluate_f64(pack), e_arr.evaluate_f64(pack)
    assert a[0][0] == b[0][0] and np.array_equal(a[2], b[2]) and abs(a[0][0] - (-144.059)) < 1e-3
    with pytest.raises(backend.BackendError):
        backend.TersoffEngine(text, device=0, species=["Ga", "As"])        # triplets with As are missing
    with pytest.raises(backend.BackendError):
        backend.TersoffEngine("Ga Ga Ga 1.0 2.0", device=0, species=["Ga"])
    e_text.close(); e_arr.close()
    table, const = golden.offset_table()
    eng = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
    s = golden.structure("SrTiO3_2x2_pristine")
    eng.upload([(s.numbers, s.positions, s.cell, s.pbc)])
    eng.run(backend.WANT_ENERGY)
    with pytest.raises(backend.BackendError):
        eng.download(backend.WANT_ENERGY | backend.WANT_FORCES)             # forces were not produced by that run
    assert np.isfinite(eng.download(backend.WANT_ENERGY)["energy"]).all()
    eng.set_positions(s.positions + 0.01)
    with pytest.raises(backend.BackendError):
        eng.download(backend.WANT_ENERGY)                                   # results belong to the old positions
    eng.close()


def test_distinct_handles_are_independent_interleaved_and_from_two_threads(golden):
    """SURVEY §8(b) threading contract: a handle is not re-entrant, DISTINCT handles are independent.  Two engines in one process,
    (a) calls interleaved on one thread, (b) each driven by its own Python thread (ctypes releases the GIL during the calls, the
    engines own separate HIP streams): every result is bit-identical to the same batch evaluated alone."""
    import threading
    from surface_sampling_amd import backend, structures

    table, const = golden.offset_table()
    base = golden.structure("SrTiO3_2x2_pristine")
    batches = [[structures.as_arrays(structures.synth_chain(base, c, grid=(4, 4))) for c in range(k, k + 6)] for k in (0, 40)]

    def alone(packs):
        eng = backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const)
        r = eng.evaluate(packs)
        out = (r["energy"].copy(), r["forces"].copy(), r["energy_std"].copy())
        eng.close()
        return out

    ref = [alone(b) for b in batches]
    engines = [backend.PainnEngine(golden.blobs, device=0, offset_per_z=table, offset_const=const) for _ in batches]
    for eng, b in zip(engines, batches):        # (a) interleaved: upload A, upload B, run A, run B, download A, download B
        eng.upload(b)
    for eng in engines:
        eng.run()
    for eng, want in zip(engines, ref):
        r = eng.download()
        assert np.array_equal(r["energy"], want[0]) and np.array_equal(r["forces"], want[1])
    errors = []

    def drive(eng, packs, want):                # (b) two threads, 25 evaluations each
        try:
            for _ in range(25):
                r = eng.evaluate(packs)
                if not (np.array_equal(r["energy"], want[0]) and np.array_equal(r["forces"], want[1])
                        and np.array_equal(r["energy_std"], want[2])):
                    errors.append("mismatch")
                    return
        except Exception as exc:   # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=drive, args=(e, b, w)) for e, b, w in zip(engines, batches, ref)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for eng in engines:
        eng.close()
    assert not errors, errors
