"""ctypes binding of ``libvssr_eval.so`` (the C ABI in ``include/vssr_eval.h``).

There is no CPU fallback: if the HIP library is missing or no GPU is visible, every entry
point raises.  The oracle under ``oracle/`` is test infrastructure and is never imported here.
"""

from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VSSR_EVAL_LIB selects another build of the same library (A/B measurements, tools/gpu_ab.sh); there is still no fallback
LIB_PATH = os.environ.get("VSSR_EVAL_LIB") or os.path.join(_HERE, "libvssr_eval.so")

WANT_ENERGY, WANT_FORCES, WANT_STD, WANT_PER_MODEL, WANT_PER_ATOM = 1, 2, 4, 8, 16
WANT_ALL = WANT_ENERGY | WANT_FORCES | WANT_STD | WANT_PER_MODEL | WANT_PER_ATOM

EXPORTS = (
    "vssr_abi_version", "vssr_create", "vssr_destroy", "vssr_last_error", "vssr_eval", "vssr_eval_batch",
    "vssr_batch_upload", "vssr_batch_set_positions", "vssr_batch_run", "vssr_batch_download",
    "vssr_synchronize", "vssr_profile_enable", "vssr_profile_reset", "vssr_profile_read",
    "vssr_batch_stats", "vssr_batch_neighbors", "vssr_debug_read", "vssr_tersoff_create",
    "vssr_tersoff_eval_batch", "vssr_batch_relax_fire", "vssr_batch_relax_bfgs", "vssr_debug_capacity",
    "vssr_batch_device_results", "vssr_eam_create", "vssr_eam_eval_batch",
    "vssr_tersoff_create_from_text", "vssr_batch_relax_cg", "vssr_batch_saturated",
    "vssr_batch_embedding", "vssr_batch_traj_configure", "vssr_batch_traj_read",
    "vssr_device_context", "vssr_batch_stress", "vssr_batch_energy_f64", "vssr_batch_device_results_f64",
    "vssr_batch_relax_counts",
)


class BackendError(RuntimeError):
    pass


class PainnConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32), ("n_models", C.c_int32),
        ("weights", C.POINTER(C.POINTER(C.c_float))), ("weights_len", C.c_uint64),
        ("feat_dim", C.c_int32), ("n_rbf", C.c_int32), ("num_conv", C.c_int32), ("n_embed", C.c_int32),
        ("readout_hidden", C.c_int32), ("cutoff", C.c_float), ("excl_vol", C.c_int32),
        ("excl_power", C.c_int32), ("excl_sigma", C.c_float), ("model_units_per_ev", C.c_double),
        ("offset_per_z", C.POINTER(C.c_double)), ("offset_const", C.c_double),
    ]


class FireParams(C.Structure):
    """vssr_fire_params; defaults = ASE FIRE defaults and the reference's relax_steps / fmax."""
    _fields_ = [("max_steps", C.c_int32), ("fmax", C.c_float), ("dt", C.c_float), ("maxstep", C.c_float),
                ("dtmax", C.c_float), ("finc", C.c_float), ("fdec", C.c_float), ("astart", C.c_float),
                ("fa", C.c_float), ("nmin", C.c_int32)]

    @classmethod
    def default(cls, max_steps=20, fmax=0.01):
        return cls(int(max_steps), float(fmax), 0.1, 0.2, 1.0, 1.1, 0.5, 0.1, 0.99, 5)


class EamGrid(C.Structure):
    _fields_ = [("nrho", C.c_int32), ("nr", C.c_int32), ("drho", C.c_double), ("dr", C.c_double), ("cutoff", C.c_double)]


class CgParams(C.Structure):
    """vssr_cg_params; defaults = the reference's LAMMPS template (minimize 1e-5 1e-5 {relax_steps} 10000, dmax 0.1)."""
    _fields_ = [("max_iter", C.c_int32), ("max_eval", C.c_int32), ("etol", C.c_double), ("ftol", C.c_double), ("dmax", C.c_double)]

    @classmethod
    def default(cls, max_iter=100, max_eval=10000, etol=1e-5, ftol=1e-5):
        return cls(int(max_iter), int(max_eval), float(etol), float(ftol), 0.1)


CG_STOP_REASONS = {1: "energy tolerance", 2: "force tolerance", 3: "max iterations", 4: "max force evaluations",
                   5: "search direction is not downhill", 6: "forces are zero", 7: "linesearch: zero quadratic step",
                   8: "linesearch alpha is zero"}


class BfgsParams(C.Structure):
    """vssr_bfgs_params; defaults = ASE BFGS defaults and the reference's relax_steps / fmax."""
    _fields_ = [("max_steps", C.c_int32), ("fmax", C.c_float), ("alpha", C.c_float), ("maxstep", C.c_float)]

    @classmethod
    def default(cls, max_steps=20, fmax=0.01):
        return cls(int(max_steps), float(fmax), 70.0, 0.2)


class Out(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_float)) for n in
                ("energy", "energy_std", "forces", "forces_std", "energy_models", "energy_atoms")]


_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  The PyTorch wheel carries its own ``libamdhip64.so`` / ``libhsa-runtime64.so``; if
    ``libvssr_eval.so`` has already brought up the system copy (``/opt/rocm``), a later ``import torch`` loads the second
    copy and finds no devices ("No HIP GPUs are available": the sharding path, which hands the engine's buffers to
    ``torch.distributed``, would fail whenever torch is imported after the first engine).  Loading torch's copy first --
    without importing torch -- makes both sides resolve the same runtime whatever the import order
    (``VSSR_SYSTEM_HIP=1`` skips this)."""
    if os.environ.get("VSSR_SYSTEM_HIP") == "1" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for loc in (spec.submodule_search_locations or []) if spec else []:
        p = os.path.join(loc, "lib", "libamdhip64.so")
        if os.path.exists(p):
            try:
                C.CDLL(p, mode=C.RTLD_GLOBAL)
            except OSError:
                pass   # (an unusable bundled copy: the system runtime serves the library)
            return


def load_library():
    """Load libvssr_eval.so; raise BackendError (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). This backend has no CPU fallback.")
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, ip, dp, fp, u8p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_float), \
        C.POINTER(C.c_uint8)
    i64p = C.POINTER(C.c_int64)
    L.vssr_abi_version.restype = C.c_int
    L.vssr_create.restype = C.c_int
    L.vssr_create.argtypes = [C.POINTER(PainnConfig), C.POINTER(vp)]
    L.vssr_destroy.restype = None
    L.vssr_destroy.argtypes = [vp]
    L.vssr_last_error.restype = C.c_char_p
    L.vssr_last_error.argtypes = [vp]
    L.vssr_eval.restype = C.c_int
    L.vssr_eval.argtypes = [vp, C.c_int32, ip, dp, dp, u8p, C.c_uint32, C.POINTER(Out)]
    L.vssr_eval_batch.restype = C.c_int
    L.vssr_eval_batch.argtypes = [vp, C.c_int32, ip, ip, dp, dp, u8p, C.c_uint32, C.POINTER(Out)]
    L.vssr_batch_upload.restype = C.c_int
    L.vssr_batch_upload.argtypes = [vp, C.c_int32, ip, ip, dp, dp, u8p]
    L.vssr_batch_set_positions.restype = C.c_int
    L.vssr_batch_set_positions.argtypes = [vp, dp]
    L.vssr_batch_run.restype = C.c_int
    L.vssr_batch_run.argtypes = [vp, C.c_uint32]
    L.vssr_batch_download.restype = C.c_int
    L.vssr_batch_download.argtypes = [vp, C.c_uint32, C.POINTER(Out)]
    L.vssr_synchronize.restype = C.c_int
    L.vssr_synchronize.argtypes = [vp]
    L.vssr_profile_enable.restype = C.c_int
    L.vssr_profile_enable.argtypes = [vp, C.c_int]
    L.vssr_profile_reset.restype = C.c_int
    L.vssr_profile_reset.argtypes = [vp]
    L.vssr_profile_read.restype = C.c_int
    L.vssr_profile_read.argtypes = [vp, C.c_int32, C.POINTER(C.c_char_p), i64p, dp, ip]
    L.vssr_batch_stats.restype = C.c_int
    L.vssr_batch_stats.argtypes = [vp, i64p, i64p, i64p]
    L.vssr_batch_neighbors.restype = C.c_int
    L.vssr_batch_neighbors.argtypes = [vp, C.c_int64, ip, ip, ip, fp, i64p]
    L.vssr_debug_read.restype = C.c_int
    L.vssr_debug_read.argtypes = [vp, C.c_char_p, C.c_int32, fp, C.c_int64, i64p]
    L.vssr_tersoff_create.restype = C.c_int
    L.vssr_tersoff_create.argtypes = [C.c_int32, C.c_int32, dp, C.POINTER(vp)]
    L.vssr_tersoff_eval_batch.restype = C.c_int
    L.vssr_tersoff_eval_batch.argtypes = [vp, C.c_int32, ip, ip, dp, dp, u8p, C.c_uint32, C.POINTER(Out), dp, dp, dp]
    L.vssr_tersoff_create_from_text.restype = C.c_int
    L.vssr_tersoff_create_from_text.argtypes = [C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(vp)]
    L.vssr_batch_relax_cg.restype = C.c_int
    L.vssr_batch_relax_cg.argtypes = [vp, C.POINTER(CgParams), u8p, C.c_uint32, dp, ip, ip, ip]
    L.vssr_eam_create.restype = C.c_int
    L.vssr_eam_create.argtypes = [C.c_int32, C.POINTER(EamGrid), dp, dp, dp, C.POINTER(vp)]
    L.vssr_eam_eval_batch.restype = C.c_int
    L.vssr_eam_eval_batch.argtypes = L.vssr_tersoff_eval_batch.argtypes
    L.vssr_batch_relax_fire.restype = C.c_int
    L.vssr_batch_relax_fire.argtypes = [vp, C.POINTER(FireParams), u8p, C.c_uint32, dp, ip, u8p]
    L.vssr_batch_relax_bfgs.restype = C.c_int
    L.vssr_batch_relax_bfgs.argtypes = [vp, C.POINTER(BfgsParams), u8p, C.c_uint32, dp, ip, u8p]
    L.vssr_batch_device_results.restype = C.c_int
    L.vssr_batch_device_results.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.vssr_batch_device_results_f64.restype = C.c_int
    L.vssr_batch_device_results_f64.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.vssr_batch_relax_counts.restype = C.c_int
    L.vssr_batch_relax_counts.argtypes = [vp, i64p, i64p]
    L.vssr_batch_energy_f64.restype = C.c_int
    L.vssr_batch_energy_f64.argtypes = [vp, dp, dp, dp]
    L.vssr_batch_traj_configure.restype = C.c_int
    L.vssr_batch_traj_configure.argtypes = [vp, C.c_int32]
    L.vssr_batch_traj_read.restype = C.c_int
    L.vssr_batch_traj_read.argtypes = [vp, C.c_int32, ip, dp, fp, dp, ip]
    L.vssr_batch_embedding.restype = C.c_int
    L.vssr_batch_embedding.argtypes = [vp, C.c_int32, fp, C.c_int64, i64p]
    L.vssr_batch_saturated.restype = C.c_int
    L.vssr_batch_saturated.argtypes = [vp, u8p, ip]
    L.vssr_batch_stress.restype = C.c_int
    L.vssr_batch_stress.argtypes = [vp, dp, dp]
    L.vssr_device_context.restype = C.c_int
    L.vssr_device_context.argtypes = [vp, ip, C.POINTER(vp), C.POINTER(vp)]
    L.vssr_debug_capacity.restype = C.c_int
    L.vssr_debug_capacity.argtypes = [vp, C.c_int32, C.c_int32, ip]
    if L.vssr_abi_version() != 1:
        raise BackendError("libvssr_eval.so ABI version mismatch")
    _lib = L
    return L


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def pack_batch(structs):
    """List of (numbers, positions, cell, pbc) -> concatenated ABI arrays."""
    n_atoms = np.array([len(s[0]) for s in structs], dtype=np.int32)
    Z = np.ascontiguousarray(np.concatenate([np.asarray(s[0]) for s in structs]), dtype=np.int32)
    pos = np.ascontiguousarray(np.concatenate([np.asarray(s[1], dtype=np.float64).reshape(-1, 3) for s in structs]))
    cell = np.ascontiguousarray(np.stack([np.asarray(s[2], dtype=np.float64).reshape(9) for s in structs]))
    pbc = np.ascontiguousarray(np.stack([np.asarray(s[3]).astype(np.uint8).reshape(3) for s in structs]))
    return n_atoms, Z, pos, cell, pbc


class _DeviceArray:
    """A float32 (or int32) vector in device memory owned by an engine (``__cuda_array_interface__`` v2)."""

    def __init__(self, ptr, n, typestr="<f4"):
        # (read-only flag False: torch refuses read-only device arrays; consumers only read these buffers)
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2,
                                         "strides": None}


class _Handle:
    """Owns a vssr_handle*; shared plumbing for the PaiNN and Tersoff engines."""

    def __init__(self):
        self._lib = load_library()
        self._h = C.c_void_p(None)
        self.n_models = 1
        self._n_cfg = 0
        self._n_atoms = 0

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.vssr_last_error(self._h)
            raise BackendError(f"vssr error {rc}: {msg.decode() if msg else '?'}")

    def close(self):
        if self._h:
            self._lib.vssr_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- resident-batch API ------------------------------------------------------------------
    def upload(self, structs):
        n_atoms, Z, pos, cell, pbc = pack_batch(structs)
        self.upload_arrays(n_atoms, Z, pos, cell, pbc)

    def upload_arrays(self, n_atoms, Z, pos, cell, pbc):
        n_atoms = np.ascontiguousarray(n_atoms, dtype=np.int32)
        Z = np.ascontiguousarray(Z, dtype=np.int32)
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        cell = np.ascontiguousarray(cell, dtype=np.float64)
        pbc = np.ascontiguousarray(pbc, dtype=np.uint8)
        if Z.shape[0] != int(n_atoms.sum()) or pos.size != 3 * Z.shape[0] or cell.size != 9 * len(n_atoms) \
                or pbc.size != 3 * len(n_atoms):
            raise ValueError("inconsistent batch arrays")
        self._check(self._lib.vssr_batch_upload(self._h, len(n_atoms), _ptr(n_atoms, C.c_int32),
                                                _ptr(Z, C.c_int32), _ptr(pos, C.c_double),
                                                _ptr(cell, C.c_double), _ptr(pbc, C.c_uint8)))
        self._n_cfg, self._n_atoms = len(n_atoms), int(Z.shape[0])
        self._cfg_start = np.concatenate([[0], np.cumsum(n_atoms)]).astype(np.int64)

    def set_positions(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        if pos.size != 3 * self._n_atoms:
            raise ValueError("positions do not match the resident batch")
        self._check(self._lib.vssr_batch_set_positions(self._h, _ptr(pos, C.c_double)))

    def run(self, want=WANT_ALL):
        self._check(self._lib.vssr_batch_run(self._h, int(want)))

    def synchronize(self):
        self._check(self._lib.vssr_synchronize(self._h))

    def download(self, want=WANT_ALL):
        B, N, M = self._n_cfg, self._n_atoms, self.n_models
        res = {
            "energy": np.zeros(B, np.float32), "energy_std": np.zeros(B, np.float32),
            "forces": np.zeros((N, 3), np.float32), "forces_std": np.zeros((N, 3), np.float32),
            "energy_models": np.zeros((B, M), np.float32), "energy_atoms": np.zeros(N, np.float32),
        }
        out = Out(*[_ptr(res[k], C.c_float) for k in
                    ("energy", "energy_std", "forces", "forces_std", "energy_models", "energy_atoms")])
        self._check(self._lib.vssr_batch_download(self._h, int(want), C.byref(out)))
        res["cfg_start"] = self._cfg_start
        res["saturated"] = self.saturated()
        # the same energies without the float32 output word (vssr_batch_energy_f64): what acceptance tests and relaxation
        # drivers compare; "energy" keeps the reference's float32 type.  Only when energies were asked for (a second call with
        # three blocking copies otherwise bought nothing).
        if int(want) & WANT_ENERGY:
            e64, s64, m64 = np.zeros(B), np.zeros(B), np.zeros((B, M))
            self._check(self._lib.vssr_batch_energy_f64(self._h, _ptr(e64, C.c_double), _ptr(s64, C.c_double), _ptr(m64, C.c_double)))
            res["energy_f64"], res["energy_std_f64"], res["energy_models_f64"] = e64, s64, m64
        return res

    def embedding(self, model=None):
        """Per-atom latent features (final scalar state, the readout's input) of the resident batch after a run:
        ``[M, sum N, F]`` for ``model=None``, ``[sum N, F]`` for one ensemble member (vssr_batch_embedding)."""
        n = C.c_int64(0)
        m = -1 if model is None else int(model)
        self._check(self._lib.vssr_batch_embedding(self._h, m, None, 0, C.byref(n)))
        buf = np.zeros(n.value, np.float32)
        self._check(self._lib.vssr_batch_embedding(self._h, m, _ptr(buf, C.c_float), n.value, C.byref(n)))
        return buf.reshape((self.n_models, self._n_atoms, -1) if model is None else (self._n_atoms, -1))

    def saturated(self):
        """bool [B]: chains whose last evaluation left the range of the fp16-split arithmetic (a value beyond +-65504 was
        clamped) or produced a non-finite energy -- their results are finite but not the model's (vssr_batch_saturated)."""
        flags = np.zeros(self._n_cfg, np.uint8)
        n = C.c_int32(0)
        self._check(self._lib.vssr_batch_saturated(self._h, _ptr(flags, C.c_uint8), C.byref(n)))
        return flags.astype(bool)

    def stress(self):
        """``(stress [B, 6], stress_std [B, 6])`` float64, Voigt order xx yy zz yz xz xy in eV / A^3 (ASE's convention): the
        virial of the LAST evaluation of every chain, from the edge gradients its reverse pass left on the device
        (vssr_batch_stress; the run must have produced forces)."""
        st, sd = np.zeros((self._n_cfg, 6)), np.zeros((self._n_cfg, 6))
        self._check(self._lib.vssr_batch_stress(self._h, _ptr(st, C.c_double), _ptr(sd, C.c_double)))
        return st, sd

    def device_results(self):
        """``(energy, energy_std)`` of the resident batch as zero-copy device arrays (objects with
        ``__cuda_array_interface__``: ``torch.as_tensor(x, device="cuda")`` wraps them).  Valid after a synchronised run."""
        e, s = C.c_void_p(None), C.c_void_p(None)
        self._check(self._lib.vssr_batch_device_results(self._h, C.byref(e), C.byref(s)))
        return _DeviceArray(e.value, self._n_cfg), _DeviceArray(s.value, self._n_cfg)

    def device_results_f64(self):
        """``(energy, energy_std)`` as zero-copy float64 device arrays (vssr_batch_device_results_f64)."""
        e, s = C.c_void_p(None), C.c_void_p(None)
        self._check(self._lib.vssr_batch_device_results_f64(self._h, C.byref(e), C.byref(s)))
        return _DeviceArray(e.value, self._n_cfg, "<f8"), _DeviceArray(s.value, self._n_cfg, "<f8")

    def device_context(self):
        """``(device ordinal, hipStream_t of the engine as int, device address of the overflow flag or None)``
        (vssr_device_context)."""
        d, st, fl = C.c_int32(0), C.c_void_p(None), C.c_void_p(None)
        self._check(self._lib.vssr_device_context(self._h, C.byref(d), C.byref(st), C.byref(fl)))
        return d.value, st.value or 0, fl.value

    def evaluate(self, structs, want=WANT_ALL):
        self.upload(structs)
        self.run(want)
        return self.download(want)

    # -- lock-step relaxation ------------------------------------------------------------------------
    def relax(self, optimizer="FIRE", **kw):
        """Dispatch on the reference's optimizer names (``mcmc/dynamics.py:119-127``: a name containing "BFGS" selects
        BFGS, everything else FIRE; BFGSLineSearch / CG / LAMMPS are not provided by this backend)."""
        name = str(optimizer)
        if "BFGSLineSearch" in name or "CG" in name or "LAMMPS" in name:
            raise BackendError(f"optimizer {optimizer!r} is not available on the device (FIRE and BFGS are)")
        return self.relax_bfgs(**kw) if "BFGS" in name else self.relax_fire(**kw)

    def relax_bfgs(self, fixed=None, max_steps=20, fmax=0.01, want=WANT_ALL, params=None, record_interval=0):
        """BFGS-relax every chain of the resident batch on the device (ASE BFGS, the reference's SrTiO3 optimizer).
        Same arguments and return value as :meth:`relax_fire`."""
        p = params or BfgsParams.default(max_steps, fmax)
        return self._relax_call(self._lib.vssr_batch_relax_bfgs, p, fixed, want, record_interval)

    def relax_fire(self, fixed=None, max_steps=20, fmax=0.01, want=WANT_ALL, params=None, record_interval=0):
        """FIRE-relax every chain of the resident batch on the device (reference optimize_slab with FIRE).
        ``fixed``: bool/uint8 [sum N], True = held fixed.  Returns dict(positions [sum N,3] float64,
        n_steps [B], converged [B]) — fetch energies/forces of the relaxed batch with download().
        ``record_interval`` k > 0 also records every chain after 0, k, 2k, ... optimizer steps (the reference's
        TrajectoryObserver, ``mcmc/dynamics.py:131-151``): key ``"traj"`` = dict(n_records [B], positions [R, sum N, 3],
        forces [R, sum N, 3] with FixAtoms applied, energies [R, B]); entries of chain b beyond n_records[b] are unused."""
        p = params or FireParams.default(max_steps, fmax)
        return self._relax_call(self._lib.vssr_batch_relax_fire, p, fixed, want, record_interval)

    def relax_counts(self):
        """``(lock-step evaluations, dispatched chain-evaluations)`` of the last relaxation (vssr_batch_relax_counts)."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.vssr_batch_relax_counts(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def debug_capacity(self, slots_per_atom=0, tight=-1):
        """Test hook (vssr_debug_capacity); returns the regrow count of the last relaxation."""
        n = C.c_int32(0)
        self._check(self._lib.vssr_debug_capacity(self._h, int(slots_per_atom), int(tight), C.byref(n)))
        return n.value

    def _relax_call(self, fn, p, fixed, want, record_interval=0):
        N, B = self._n_atoms, self._n_cfg
        self._check(self._lib.vssr_batch_traj_configure(self._h, int(record_interval or 0)))
        fx = None
        if fixed is not None:
            fx = np.ascontiguousarray(fixed, dtype=np.uint8)
            if fx.size != N:
                raise ValueError("fixed mask does not match the resident batch")
        pos = np.zeros((N, 3), np.float64)
        steps = np.zeros(B, np.int32)
        conv = np.zeros(B, np.uint8)
        self._check(fn(self._h, C.byref(p), _ptr(fx, C.c_uint8), int(want), _ptr(pos, C.c_double),
                       _ptr(steps, C.c_int32), _ptr(conv, C.c_uint8)))
        out = {"positions": pos, "n_steps": steps, "converged": conv.astype(bool)}
        self.last_relax_counts = self.relax_counts()
        if record_interval:
            R = C.c_int32(0)
            self._check(self._lib.vssr_batch_traj_read(self._h, 0, None, None, None, None, C.byref(R)))
            R = R.value
            n_rec = np.zeros(B, np.int32)
            tpos, tf, te = np.zeros((R, N, 3), np.float64), np.zeros((R, N, 3), np.float32), np.zeros((R, B), np.float64)
            self._check(self._lib.vssr_batch_traj_read(self._h, R, _ptr(n_rec, C.c_int32), _ptr(tpos, C.c_double),
                                                       _ptr(tf, C.c_float), _ptr(te, C.c_double), None))
            self._check(self._lib.vssr_batch_traj_configure(self._h, 0))
            out["traj"] = {"n_records": n_rec, "positions": tpos, "forces": tf, "energies": te,
                           "record_interval": int(record_interval)}
        return out

    # -- introspection -----------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self._lib.vssr_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        self._check(self._lib.vssr_profile_reset(self._h))

    def profile_read(self):
        cap = 32
        names = (C.c_char_p * cap)()
        launches = np.zeros(cap, np.int64)
        ms = np.zeros(cap, np.float64)
        n = C.c_int32(0)
        self._check(self._lib.vssr_profile_read(self._h, cap, names, _ptr(launches, C.c_int64),
                                                _ptr(ms, C.c_double), C.byref(n)))
        return {names[k].decode(): {"launches": int(launches[k]), "total_ms": float(ms[k])}
                for k in range(n.value)}

    def stats(self):
        a, e, s = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._check(self._lib.vssr_batch_stats(self._h, C.byref(a), C.byref(e), C.byref(s)))
        return {"atoms": a.value, "edges": e.value, "slots": s.value}

    def neighbors(self):
        n = C.c_int64(0)
        self._check(self._lib.vssr_batch_neighbors(self._h, 0, None, None, None, None, C.byref(n)))
        E = n.value
        ei = np.zeros(E, np.int32); ej = np.zeros(E, np.int32)
        eS = np.zeros((E, 3), np.int32); er = np.zeros((E, 3), np.float32)
        self._check(self._lib.vssr_batch_neighbors(self._h, E, _ptr(ei, C.c_int32), _ptr(ej, C.c_int32),
                                                   _ptr(eS, C.c_int32), _ptr(er, C.c_float), C.byref(n)))
        return ei, ej, eS, er

    def debug_read(self, name, model=0):
        n = C.c_int64(0)
        self._check(self._lib.vssr_debug_read(self._h, name.encode(), model, None, 0, C.byref(n)))
        buf = np.zeros(n.value, np.float32)
        self._check(self._lib.vssr_debug_read(self._h, name.encode(), model, _ptr(buf, C.c_float), n.value,
                                              C.byref(n)))
        return buf


class PainnEngine(_Handle):
    """PaiNN-ensemble evaluator on one GPU (one handle = one HIP stream)."""

    has_device_results = True     # vssr_batch_device_results serves fp32 PaiNN handles (sharding.ShardedEnsemble's device path)

    def __init__(self, blobs, device=0, cutoff=5.0, model_units_per_ev=23.0605, offset_per_z=None,
                 offset_const=0.0, hparams=None):
        super().__init__()
        hp = {"feat_dim": 128, "n_rbf": 20, "num_conv": 3, "n_embed": 100, "readout_hidden": 64,
              "excl_vol": True, "V_ex_power": 12, "V_ex_sigma": 1.5}
        hp.update(hparams or {})
        self._blobs = [np.ascontiguousarray(b, dtype=np.float32) for b in blobs]
        if not self._blobs:
            raise ValueError("at least one model is required")
        M = len(self._blobs)
        ptrs = (C.POINTER(C.c_float) * M)(*[_ptr(b, C.c_float) for b in self._blobs])
        off = None
        if offset_per_z is not None:
            off = np.ascontiguousarray(offset_per_z, dtype=np.float64)
            if off.size != hp["n_embed"]:
                raise ValueError("offset_per_z must have n_embed entries")
        cfg = PainnConfig(
            C.sizeof(PainnConfig), int(device), M, ptrs, self._blobs[0].size, hp["feat_dim"], hp["n_rbf"],
            hp["num_conv"], hp["n_embed"], hp["readout_hidden"], float(cutoff), int(bool(hp["excl_vol"])),
            int(hp["V_ex_power"]), float(hp["V_ex_sigma"]), float(model_units_per_ev),
            _ptr(off, C.c_double), float(offset_const))
        rc = self._lib.vssr_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            msg = self._lib.vssr_last_error(None)
            raise BackendError(f"vssr_create failed ({rc}): {msg.decode() if msg else '?'}")
        self.n_models = M
        self.cutoff = float(cutoff)


class _AnalyticEngine(_Handle):
    """Shared fp64 interface of the analytic potentials (Tersoff, EAM): types instead of atomic numbers."""

    has_device_results = False    # fp64 results: sharding uses the host result path

    def relax_f64(self, structs, fixed=None, max_steps=100, fmax=0.01, optimizer="FIRE"):
        """Relax (types, positions, cell, pbc) structures with FIRE / BFGS; returns (energy [B], e_atom [N], forces [N,3],
        positions [N,3], n_steps [B], converged [B]) with fp64 energies/forces of the relaxed structures."""
        return self.relax_arrays_f64(*pack_batch(structs), fixed=fixed, max_steps=max_steps, fmax=fmax, optimizer=optimizer)

    def relax_arrays_f64(self, n_atoms, T, pos, cell, pbc, fixed=None, max_steps=100, fmax=0.01, optimizer="FIRE"):
        """``relax_f64`` on the ABI's packed arrays (no per-structure objects)."""
        self.upload_arrays(n_atoms, T, pos, cell, pbc)
        info = self.relax(optimizer, fixed=fixed, max_steps=max_steps, fmax=fmax,
                          want=WANT_ENERGY | WANT_FORCES | WANT_PER_ATOM)
        e, ea, f = self.evaluate_arrays_f64(n_atoms, T, info["positions"], cell, pbc)
        return e, ea, f, info["positions"], info["n_steps"], info["converged"]

    def relax_cg_f64(self, structs, fixed=None, max_iter=100, max_eval=10000, etol=1e-5, ftol=1e-5):
        """LAMMPS ``min_style cg`` / ``minimize etol ftol max_iter max_eval`` on the device (vssr_batch_relax_cg).  Returns
        (energy [B], e_atom [N], forces [N,3], positions [N,3], n_iter [B], n_eval [B], stop_reason [B])."""
        return self.relax_cg_arrays_f64(*pack_batch(structs), fixed=fixed, max_iter=max_iter, max_eval=max_eval, etol=etol, ftol=ftol)

    def relax_cg_arrays_f64(self, n_atoms, T, pos, cell, pbc, fixed=None, max_iter=100, max_eval=10000, etol=1e-5, ftol=1e-5):
        """``relax_cg_f64`` on the ABI's packed arrays: the minimisation, then the static evaluation of the minimised geometries
        (the reference's ``run_lammps_opt`` followed by ``run_lammps_energy``)."""
        self.upload_arrays(n_atoms, T, pos, cell, pbc)
        N, B = self._n_atoms, self._n_cfg
        fx = None
        if fixed is not None:
            fx = np.ascontiguousarray(fixed, dtype=np.uint8)
            if fx.size != N:
                raise ValueError("fixed mask does not match the resident batch")
        p = CgParams.default(max_iter, max_eval, etol, ftol)
        out = np.zeros((N, 3), np.float64)
        it, ev, why = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
        self._check(self._lib.vssr_batch_relax_cg(self._h, C.byref(p), _ptr(fx, C.c_uint8),
                                                  WANT_ENERGY | WANT_FORCES | WANT_PER_ATOM, _ptr(out, C.c_double),
                                                  _ptr(it, C.c_int32), _ptr(ev, C.c_int32), _ptr(why, C.c_int32)))
        self.last_relax_counts = self.relax_counts()
        e, ea, f = self.evaluate_arrays_f64(n_atoms, T, out, cell, pbc)
        return e, ea, f, out, it, ev, why

    def evaluate_f64(self, structs, want=WANT_ENERGY | WANT_FORCES | WANT_PER_ATOM):
        """structs: list of (types, positions, cell, pbc). Returns fp64 energy [B], e_atom [N], forces [N,3]."""
        return self.evaluate_arrays_f64(*pack_batch(structs), want=want)

    def evaluate_arrays_f64(self, n_atoms, T, pos, cell, pbc, want=WANT_ENERGY | WANT_FORCES | WANT_PER_ATOM):
        """``evaluate_f64`` on the ABI's packed arrays."""
        n_atoms = np.ascontiguousarray(n_atoms, dtype=np.int32)
        T = np.ascontiguousarray(T, dtype=np.int32)
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        cell = np.ascontiguousarray(cell, dtype=np.float64)
        pbc = np.ascontiguousarray(pbc, dtype=np.uint8)
        B, N = len(n_atoms), len(T)
        if N != int(n_atoms.sum()) or pos.size != 3 * N or cell.size != 9 * B or pbc.size != 3 * B:
            raise ValueError("inconsistent batch arrays")
        e = np.zeros(B); ea = np.zeros(N); f = np.zeros((N, 3))
        self._check(self._lib.vssr_tersoff_eval_batch(
            self._h, B, _ptr(n_atoms, C.c_int32), _ptr(T, C.c_int32), _ptr(pos, C.c_double),
            _ptr(cell, C.c_double), _ptr(pbc, C.c_uint8), int(want), None, _ptr(e, C.c_double),
            _ptr(ea, C.c_double), _ptr(f, C.c_double)))
        self._n_cfg, self._n_atoms = B, N
        self._cfg_start = np.concatenate([[0], np.cumsum(n_atoms)]).astype(np.int64)
        return e, ea, f


class TersoffEngine(_AnalyticEngine):
    """Tersoff evaluator (fp64 on device)."""

    def __init__(self, params, device=0, species=None):
        """``params``: array [nt, nt, nt, 14], or the TEXT of a LAMMPS tersoff file together with ``species`` (LAMMPS type
        order) -- then the file is parsed by the library (vssr_tersoff_create_from_text)."""
        super().__init__()
        if isinstance(params, (str, bytes)):
            if not species:
                raise ValueError("species (LAMMPS type order) are required with a potential text")
            text = params if isinstance(params, bytes) else params.encode()
            arr = (C.c_char_p * len(species))(*[s.encode() for s in species])
            self.n_types = len(species)
            rc = self._lib.vssr_tersoff_create_from_text(int(device), text, len(species), arr, C.byref(self._h))
            if rc != 0:
                msg = self._lib.vssr_last_error(None)
                raise BackendError(f"vssr_tersoff_create_from_text failed ({rc}): {msg.decode() if msg else '?'}")
            return
        params = np.ascontiguousarray(params, dtype=np.float64)
        if params.ndim != 4 or params.shape[3] != 14 or not (params.shape[0] == params.shape[1] == params.shape[2]):
            raise ValueError("params must be [nt, nt, nt, 14]")
        self.n_types = params.shape[0]
        rc = self._lib.vssr_tersoff_create(int(device), self.n_types, _ptr(params, C.c_double), C.byref(self._h))
        if rc != 0:
            msg = self._lib.vssr_last_error(None)
            raise BackendError(f"vssr_tersoff_create failed ({rc}): {msg.decode() if msg else '?'}")


class EAMEngine(_AnalyticEngine):
    """One-element EAM (LAMMPS funcfl tables) evaluator, fp64 on device; every atom has type 0."""

    def __init__(self, funcfl, device=0):
        super().__init__()
        grid = EamGrid(int(funcfl.nrho), int(funcfl.nr), float(funcfl.drho), float(funcfl.dr), float(funcfl.cutoff))
        frho, zr, rhor = (np.ascontiguousarray(a, dtype=np.float64) for a in (funcfl.frho, funcfl.zr, funcfl.rhor))
        if frho.size != funcfl.nrho or zr.size != funcfl.nr or rhor.size != funcfl.nr:
            raise ValueError("EAM tables do not match their grid")
        self.n_types = 1
        rc = self._lib.vssr_eam_create(int(device), C.byref(grid), _ptr(frho, C.c_double), _ptr(zr, C.c_double),
                                       _ptr(rhor, C.c_double), C.byref(self._h))
        if rc != 0:
            msg = self._lib.vssr_last_error(None)
            raise BackendError(f"vssr_eam_create failed ({rc}): {msg.decode() if msg else '?'}")
