"""ctypes binding of the CPU ORACLE (oracle/libvssr_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (surface-sampling_amd/) never imports this module.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VSSR_ORACLE_LIB: another build of the same library (the AddressSanitizer build of `make -C oracle asan`, tests/test_oracle_asan.py)
_LIB_PATH = os.environ.get("VSSR_ORACLE_LIB") or os.path.join(_HERE, "libvssr_oracle.so")

EV_TO_KCAL_MOL = 23.0605   # nff/utils/constants.py (confirmed by the KATs)
HARTREE_TO_EV = 27.2114


class HParams(C.Structure):
    _fields_ = [
        ("feat_dim", C.c_int32), ("n_rbf", C.c_int32), ("num_conv", C.c_int32),
        ("n_embed", C.c_int32), ("readout_hidden", C.c_int32), ("excl_vol", C.c_int32),
        ("excl_power", C.c_int32), ("cutoff", C.c_float), ("excl_sigma", C.c_float),
    ]


_DPTR = C.POINTER(C.c_double)
_DUMP_LAYER_FIELDS = ("phi", "s_msg", "v_msg", "s_upd", "v_upd", "sbar_msg", "vbar_msg", "sbar_in",
                      "vbar_in")


class Dump(C.Structure):
    _fields_ = ([(name, _DPTR * 8) for name in _DUMP_LAYER_FIELDS]
                + [("e_atom", _DPTR), ("edge_gbar", _DPTR)])


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("vssr_oracle.c", "painn_impl.inc", "vssr_oracle.h")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src if os.path.exists(s))
    if (force or stale) and not os.environ.get("VSSR_ORACLE_LIB"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libvssr_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        dp, ip, u8p, fp = (C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8),
                           C.POINTER(C.c_float))
        L.orc_neighbors.restype = C.c_int64
        L.orc_neighbors.argtypes = [C.c_int32, dp, dp, u8p, C.c_double, C.c_int64, ip, ip, ip, dp]
        L.orc_painn_eval.restype = C.c_int
        L.orc_painn_eval.argtypes = [C.c_int, fp, C.c_int64, C.POINTER(HParams), C.c_int32, ip, dp, dp,
                                     u8p, dp, dp, C.POINTER(Dump)]
        L.orc_ensemble_eval.restype = C.c_int
        L.orc_ensemble_eval.argtypes = [C.c_int, C.c_int32, C.POINTER(fp), C.c_int64, C.POINTER(HParams),
                                        C.c_double, dp, C.c_double, C.c_int32, ip, dp, dp, u8p,
                                        dp, dp, dp, dp, dp]
        L.orc_set_threads.restype = C.c_int
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_tersoff_eval.restype = C.c_int
        L.orc_tersoff_eval.argtypes = [C.c_int32, dp, C.c_int32, ip, dp, dp, u8p, dp, dp, dp]
        _lib = L
    return _lib


def set_threads(n: int) -> int:
    """OpenMP threads used by the oracle from now on."""
    return int(lib().orc_set_threads(int(n)))


def default_hparams(**over) -> HParams:
    hp = HParams(128, 20, 3, 100, 64, 1, 12, 5.0, 1.5)
    for k, v in over.items():
        setattr(hp, k, v)
    return hp


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _prep(Z, pos, cell, pbc):
    Z = np.ascontiguousarray(Z, dtype=np.int32)
    pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
    cell = np.ascontiguousarray(cell, dtype=np.float64).reshape(9)
    pbc = np.ascontiguousarray(pbc, dtype=np.uint8).reshape(3)
    return Z, pos, cell, pbc


def neighbors(pos, cell, pbc, cutoff):
    """Sorted directed edge multigraph: (i, j, S[E,3], r[E,3])."""
    _, pos, cell, pbc = _prep(np.zeros(len(pos), np.int32), pos, cell, pbc)
    n = len(pos)
    cap = max(64, n * 96)
    while True:
        ei = np.empty(cap, np.int32); ej = np.empty(cap, np.int32)
        eS = np.empty((cap, 3), np.int32); er = np.empty((cap, 3), np.float64)
        E = lib().orc_neighbors(n, _p(pos, C.c_double), _p(cell, C.c_double), _p(pbc, C.c_uint8),
                                float(cutoff), cap, _p(ei, C.c_int32), _p(ej, C.c_int32),
                                _p(eS, C.c_int32), _p(er, C.c_double))
        if E < 0:
            raise RuntimeError(f"orc_neighbors failed: {E}")
        if E <= cap:
            return ei[:E].copy(), ej[:E].copy(), eS[:E].copy(), er[:E].copy()
        cap = int(E)


def painn(blob, Z, pos, cell, pbc, real_bits=64, want_grad=True, dump=False, hp=None):
    """Single model: energy (kcal/mol), dE/dx [N,3], optional dict of intermediates."""
    hp = hp or default_hparams()
    Z, pos, cell, pbc = _prep(Z, pos, cell, pbc)
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    n, F, L = len(Z), hp.feat_dim, hp.num_conv
    e = C.c_double(0.0)
    grad = np.zeros((n, 3), np.float64)
    d = None
    arrays = {}
    if dump:
        d = Dump()
        shapes = {"phi": (n, 3 * F), "s_msg": (n, F), "v_msg": (n, 3, F), "s_upd": (n, F),
                  "v_upd": (n, 3, F), "sbar_msg": (n, F), "vbar_msg": (n, 3, F), "sbar_in": (n, F),
                  "vbar_in": (n, 3, F)}
        for name, shp in shapes.items():
            arrays[name] = [np.zeros(shp, np.float64) for _ in range(L)]
            arr_t = getattr(d, name)
            for l in range(L):
                arr_t[l] = _p(arrays[name][l], C.c_double)
        arrays["e_atom"] = np.zeros(n, np.float64)
        d.e_atom = _p(arrays["e_atom"], C.c_double)
        ei, ej, eS, er = neighbors(pos, cell, pbc, hp.cutoff)
        arrays["edge_gbar"] = np.zeros((len(ei), 3), np.float64)
        d.edge_gbar = _p(arrays["edge_gbar"], C.c_double)
        arrays["edges"] = (ei, ej, eS, er)
    rc = lib().orc_painn_eval(int(real_bits), _p(blob, C.c_float), blob.size, C.byref(hp), n,
                              _p(Z, C.c_int32), _p(pos, C.c_double), _p(cell, C.c_double),
                              _p(pbc, C.c_uint8), C.byref(e),
                              _p(grad, C.c_double) if (want_grad or dump) else None,
                              C.byref(d) if d is not None else None)
    if rc:
        raise RuntimeError(f"orc_painn_eval failed: {rc}")
    return (e.value, grad, arrays) if dump else (e.value, grad)


def ensemble(blobs, Z, pos, cell, pbc, real_bits=64, offset_per_z=None, offset_const=0.0,
             model_units_per_ev=EV_TO_KCAL_MOL, hp=None):
    """EnsembleNFF-equivalent: dict(energy, energy_std, forces, forces_std, energy_models) in eV."""
    hp = hp or default_hparams()
    Z, pos, cell, pbc = _prep(Z, pos, cell, pbc)
    blobs = [np.ascontiguousarray(b, dtype=np.float32) for b in blobs]
    M, n = len(blobs), len(Z)
    ptrs = (C.POINTER(C.c_float) * M)(*[_p(b, C.c_float) for b in blobs])
    em, es = C.c_double(0), C.c_double(0)
    fm = np.zeros((n, 3)); fs = np.zeros((n, 3)); emod = np.zeros(M)
    off = None
    if offset_per_z is not None:
        off = np.ascontiguousarray(offset_per_z, dtype=np.float64)
    rc = lib().orc_ensemble_eval(int(real_bits), M, ptrs, blobs[0].size, C.byref(hp),
                                 float(model_units_per_ev),
                                 _p(off, C.c_double) if off is not None else None, float(offset_const),
                                 n, _p(Z, C.c_int32), _p(pos, C.c_double), _p(cell, C.c_double),
                                 _p(pbc, C.c_uint8), C.byref(em), C.byref(es), _p(fm, C.c_double),
                                 _p(fs, C.c_double), _p(emod, C.c_double))
    if rc:
        raise RuntimeError(f"orc_ensemble_eval failed: {rc}")
    return {"energy": em.value, "energy_std": es.value, "forces": fm, "forces_std": fs,
            "energy_models": emod}


def tersoff(params, types, pos, cell, pbc, want_forces=True):
    """params [nt,nt,nt,14] (LAMMPS column order); returns energy, e_atom [N], forces [N,3]."""
    params = np.ascontiguousarray(params, dtype=np.float64)
    nt = params.shape[0]
    types, pos, cell, pbc = _prep(types, pos, cell, pbc)
    n = len(types)
    e = C.c_double(0)
    ea = np.zeros(n); F = np.zeros((n, 3))
    rc = lib().orc_tersoff_eval(nt, _p(params, C.c_double), n, _p(types, C.c_int32), _p(pos, C.c_double),
                                _p(cell, C.c_double), _p(pbc, C.c_uint8), C.byref(e),
                                _p(ea, C.c_double), _p(F, C.c_double) if want_forces else None)
    if rc:
        raise RuntimeError(f"orc_tersoff_eval failed: {rc}")
    return e.value, ea, F
