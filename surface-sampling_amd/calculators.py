"""ASE-Calculator-shaped front ends of the MI355X backend, mirroring the reference's classes.

Reference interfaces mirrored (``mcmc/calculators/calculators.py``):
  * ``EnsembleNFFSurface`` (:366-489)  -> :class:`EnsembleNFFSurface` here (same name, ctor kwargs
    ``device, model_units, prediction_units, offset_units``, ``set(**kw)``, ``parameters``,
    ``implemented_properties``, ``calculate``, ``get_surface_energy``, ``results`` keys/shapes,
    ``atoms.results.update`` side effect, ``models`` list).
  * ``LAMMPSSurfCalc`` / ``LAMMMPSCalc`` (:492-752) with ``pair_style tersoff``
                                       -> :class:`TersoffSurfCalc` (``energy``, ``per_atom_energies``,
    ``forces``, ``surface_energy``; no files, no LAMMPS process).
Callers on the reference side: ``SurfaceSystem.get_surface_energy`` (``mcmc/system.py:450-470``),
``optimize_slab`` (``mcmc/dynamics.py:83-170``), ``get_results_single`` (``calculators.py:34-47``).

ASE is optional: with ASE installed the classes derive from ``ase.calculators.calculator.Calculator``;
without it they derive from a small stand-in implementing the same caching protocol
(``get_property`` / ``check_state`` / ``get_potential_energy`` / ``get_forces``).
All numerical work happens in ``libvssr_eval.so``; there is no CPU fallback.
"""

from __future__ import annotations

import copy
import dataclasses
import json
import logging
import os
import re
from collections import Counter

import numpy as np

from . import backend, checkpoint, structures, tersoff as tersoff_io

EV_TO_KCAL_MOL = 23.0605   # nff/utils/constants.py
HARTREE_TO_EV = 27.2114    # nff/utils/constants.py (reference import: calculators.py:21)

all_changes = ["positions", "numbers", "cell", "pbc", "initial_charges", "initial_magmoms"]

try:  # pragma: no cover - ASE is not installed in the build container
    from ase.calculators.calculator import Calculator as _AseCalculator
    from ase.calculators.calculator import all_changes as _ase_all_changes

    all_changes = list(_ase_all_changes)
    HAVE_ASE = True
except Exception:  # ImportError or a broken ASE install
    _AseCalculator = None
    HAVE_ASE = False


class _MiniCalculator:
    """The part of ``ase.calculators.calculator.Calculator`` the reference's callers rely on."""

    implemented_properties: tuple = ()
    default_parameters: dict = {}
    name = "calculator"

    def __init__(self, restart=None, label=None, atoms=None, **kwargs):
        self.atoms = None
        self.results: dict = {}
        self.parameters: dict = dict(self.default_parameters)
        self.label = label
        if atoms is not None:
            atoms.calc = self
        self.set(**kwargs)

    def set(self, **kwargs) -> dict:
        changed = {}
        for key, value in kwargs.items():
            old = self.parameters.get(key, None)
            same = False
            try:
                same = key in self.parameters and bool(np.all(old == value))
            except Exception:
                same = False
            if not same:
                changed[key] = value
                self.parameters[key] = value
        if changed:
            self.reset()
        return changed

    def reset(self):
        self.atoms = None
        self.results = {}

    @staticmethod
    def _snapshot(atoms):
        Z, pos, cell, pbc = structures.as_arrays(atoms)
        return structures.Structure(Z.copy(), pos.copy(), cell.copy(), pbc.astype(bool))

    def check_state(self, atoms, tol=1e-15):
        if self.atoms is None:
            return list(all_changes)
        old, new = structures.as_arrays(self.atoms), structures.as_arrays(atoms)
        changes = []
        for key, a, b in zip(("numbers", "positions", "cell", "pbc"), old, new):
            if a.shape != b.shape or not np.allclose(a, b, rtol=0, atol=tol):
                changes.append(key)
        return changes

    def calculate(self, atoms=None, properties=("energy",), system_changes=all_changes):
        if atoms is not None:
            self.atoms = self._snapshot(atoms)

    def get_property(self, name, atoms=None, allow_calculation=True):
        if name not in self.implemented_properties:
            raise NotImplementedError(f"{name} property not implemented")
        if atoms is None:
            raise ValueError("atoms is required")
        changes = self.check_state(atoms)
        if changes:
            self.results = {}
        if name not in self.results:
            if not allow_calculation:
                return None
            self.calculate(atoms, [name], changes)
        result = self.results[name]
        if isinstance(result, np.ndarray):
            result = result.copy()
        return result

    def get_potential_energy(self, atoms=None, force_consistent=False):
        return self.get_property("energy", atoms)

    def get_forces(self, atoms=None):
        return self.get_property("forces", atoms)

    def calculation_required(self, atoms, properties):
        if self.check_state(atoms):
            return True
        return any(p not in self.results for p in properties)


_Base = _AseCalculator if HAVE_ASE else _MiniCalculator


def _device_index(device) -> int:
    """'cuda', 'cuda:1', 1 -> ordinal; 'cpu' is rejected (the product has no CPU path)."""
    if isinstance(device, (int, np.integer)):
        return int(device)
    s = str(device).lower()
    if s.startswith("cpu"):
        raise backend.BackendError("device='cpu' requested: this backend only runs on an MI355X (HIP) device")
    if ":" in s:
        return int(s.split(":", 1)[1])
    return 0


def _units_per_ev(model_units: str, prediction_units: str) -> float:
    mu, pu = model_units.lower(), prediction_units.lower()
    if pu != "ev":
        raise ValueError("prediction_units must be 'eV'")
    if mu in ("kcal/mol", "kcal"):
        return EV_TO_KCAL_MOL
    if mu == "ev":
        return 1.0
    if mu in ("atomic", "hartree", "ha"):
        return 1.0 / HARTREE_TO_EV
    raise ValueError(f"unknown model_units {model_units!r}")


def stoich_offset_table(offset_data: dict, n_embed: int = 100, offset_units: str = "atomic"):
    """``stoidict`` -> per-species eV table + constant, as EnsembleNFF adds it to every prediction when ``offset_data`` is
    configured.  ``offset_units="atomic"`` (the PaiNN configuration,
    ``scripts/sample_surface.py:173``): the entries are Hartree; any other unit string ("eV"): taken as eV."""
    stoidict = offset_data["stoidict"]
    factor = HARTREE_TO_EV if offset_units == "atomic" else 1.0
    table = np.zeros(n_embed)
    for sym, val in stoidict.items():
        if sym == "offset":
            continue
        table[structures.ATOMIC_NUMBERS[sym]] = float(val) * factor
    return table, float(stoidict.get("offset", 0.0)) * factor


def surface_energy_from_energy(energy: float, symbols, chem_pots: dict, offset_data: dict,
                               offset_units: str = "atomic") -> float:
    """Bulk-reference and chemical-potential bookkeeping of
    ``EnsembleNFFSurface.get_surface_energy`` (reference ``calculators.py:379-446``)."""
    ads_count = Counter(symbols)
    bulk_energies = offset_data["bulk_energies"]
    stoics = offset_data["stoics"]
    ref_formula = offset_data["ref_formula"]
    ref_element = offset_data["ref_element"]
    bulk_ref_en = ads_count[ref_element] * bulk_energies[ref_formula]
    for ele in ads_count:
        if ele != ref_element:
            bulk_ref_en += (ads_count[ele] - stoics[ele] / stoics[ref_element] * ads_count[ref_element]) \
                * bulk_energies[ele]
    surface_energy = energy - (bulk_ref_en * HARTREE_TO_EV if offset_units == "atomic" else bulk_ref_en)
    pot = 0.0
    for ele in ads_count:
        if ele != ref_element:
            pot += (ads_count[ele] - stoics[ele] / stoics[ref_element] * ads_count[ref_element]) * chem_pots[ele]
    return surface_energy - pot


class _LockstepProxy(_Base):
    """Calculator of ONE chain inside a host-driven optimizer (host_opt.optimizer_class_batch): energy and forces come from the
    lock-step evaluation of the whole resident batch."""

    implemented_properties = ("energy", "forces")
    name = "vssr_lockstep_proxy"

    def __init__(self, request, **kwargs):
        super().__init__(**kwargs)
        self._request = request

    def calculate(self, atoms=None, properties=("energy",), system_changes=all_changes):
        super().calculate(atoms, properties, system_changes)
        e, f = self._request(atoms)
        self.results = {"energy": float(e), "forces": np.asarray(f, dtype=np.float64)}


class EnsembleNFFSurface(_Base):
    """PaiNN-ensemble surface calculator on MI355X (drop-in for the reference class of the same name).

    Args:
        models: list of checkpoints — paths to nff ``best_model`` files / canonical ``.f32`` blobs, or
            float32 arrays already in the canonical layout (``include/vssr_eval.h``).
        device: ``"cuda"``, ``"cuda:N"`` or an int ordinal.
        model_units / prediction_units / offset_units: as in the reference (``"kcal/mol"``, ``"eV"``,
            ``"atomic"``).
        cutoff: neighbor cutoff in Å (the reference passes it through ``get_atoms_batch``).
        position_dtype: ``"float64"`` (default: the caller's positions as they are) or ``"float32"``: positions are rounded to
            float32 before an evaluation, the way nff's ``AtomsBatch`` holds them (``nxyz`` is a float32 tensor, reference
            ``mcmc/utils/misc.py:34-42``) -- single-point results then reproduce the reference's prints to the last printed
            digit of fmax (1e-5 eV/A).  Applies to ``calculate``, ``calculate_batch`` and the
            single-point form of ``evaluate_packed``; relaxations move fp64 positions on the device either way.
        properties: like nff's ``NeuralFF(properties=[...])``; with ``"embedding"`` in it ``results["embedding"]`` holds the
            per-atom latent features ``[N, 128]`` (final scalar state) of the first model -- the reference computes them
            with ``NeuralFF(models[0], properties=["energy", "forces", "embedding"])`` -- and ``results["embedding_models"]``
            those of every ensemble member ``[M, N, 128]``.
    """

    # "stress" / "stress_std" like nff's EnsembleNFF (the reference's class attribute, calculators.py:369, inherits them): the
    # virial of the same evaluation, Voigt 6-vector in eV / A^3 (what ase.Atoms.get_stress asks a calculator for)
    implemented_properties = ("energy", "forces", "stress", "energy_std", "forces_std", "stress_std", "surface_energy", "embedding")
    name = "ensemble_nff_surface_mi355x"

    def __init__(self, models, device="cuda", model_units="kcal/mol", prediction_units="eV",
                 offset_units="atomic", cutoff=5.0, hparams=None, logger=None, properties=("energy", "forces"),
                 position_dtype="float64", **kwargs):
        # a single model is accepted as well: the reference's NFFPourbaix is a NeuralFF and is constructed with ONE module,
        # NFFPourbaix(models[0], device=..., model_units=..., prediction_units="eV") (scripts/sample_pourbaix_surface.py:253-258)
        if isinstance(models, (str, bytes)) or hasattr(models, "__fspath__") or hasattr(models, "state_dict") \
                or (isinstance(models, np.ndarray) and models.ndim == 1):
            models = [models]
        self.models = [self._load_model(m, hparams) for m in models]
        # like NeuralFF(properties=[...]) in the reference's clustering script (scripts/clustering.py:150-158): "embedding" in
        # this list makes every calculate() also return the latent features
        self.properties = tuple(properties)
        self.device = device
        self.model_units = model_units
        self.prediction_units = prediction_units
        self.offset_units = offset_units
        self.cutoff = float(cutoff)
        if str(np.dtype(position_dtype)) not in ("float64", "float32"):
            raise ValueError(f"position_dtype must be float64 or float32, not {position_dtype!r}")
        self.position_dtype = str(np.dtype(position_dtype))
        self.hparams = dict(hparams or {})
        self.chem_pots: dict = {}
        self.offset_data: dict = {}
        self.logger = logger or logging.getLogger(__name__)
        self._engine = None
        self._engine_key = None
        super().__init__(**kwargs)

    @staticmethod
    def _load_model(m, hparams):
        """A model may be a checkpoint path (nff ``best_model`` or canonical ``.f32``), a float32 blob in the canonical
        layout, or -- what ``scripts/sample_surface.py:164-174`` passes -- an nff ``Painn`` module (anything with
        ``state_dict()``): its tensors are taken over, its excluded-volume / cutoff attributes are checked against
        ``hparams`` like those of a checkpoint file."""
        if isinstance(m, (str, bytes)) or hasattr(m, "__fspath__"):
            return checkpoint.load_painn_blob(str(m), hparams)
        if hasattr(m, "state_dict") and callable(m.state_dict):
            sd = {}
            for k, v in m.state_dict().items():
                for attr in ("detach", "cpu", "numpy"):
                    if hasattr(v, attr):
                        v = getattr(v, attr)()
                sd[k] = np.asarray(v, dtype=np.float32)
            attrs = {a: getattr(m, a) for a in ("excl_vol", "power", "sigma", "cutoff") if hasattr(m, a)}
            checkpoint.check_model_against_hparams(type(m).__name__, sd, hparams, attrs=attrs)
            return checkpoint.state_dict_to_blob(sd, hparams)
        blob = np.ascontiguousarray(m, dtype=np.float32).reshape(-1)
        checkpoint.blob_to_fields(blob, hparams)
        return blob

    # -- engine lifetime (lazy: created on first use, re-created when the offset config changes) ----
    def _offset_config(self):
        if self.parameters.get("offset", False) and self.offset_data and "stoidict" in self.offset_data:
            table, const = stoich_offset_table(self.offset_data, offset_units=self.offset_units)
            return table, const
        return None, 0.0

    def _get_engine(self):
        table, const = self._offset_config()
        key = (None if table is None else table.tobytes(), const, _device_index(self.device), self.cutoff)
        if self._engine is None or key != self._engine_key:
            if self._engine is not None:
                self._engine.close()
            self._engine = backend.PainnEngine(
                self.models, device=_device_index(self.device), cutoff=self.cutoff,
                model_units_per_ev=_units_per_ev(self.model_units, self.prediction_units),
                offset_per_z=table, offset_const=const, hparams=self.hparams)
            self._engine_key = key
        return self._engine

    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_engine", "_engine_key"):
                setattr(new, k, None)
            elif k in ("_extra_engines", "_extra_key"):
                continue
            elif k in ("models", "logger"):
                setattr(new, k, v)  # weights are immutable: share
            else:
                setattr(new, k, copy.deepcopy(v, memo))
        return new

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_engine"] = None
        state["_engine_key"] = None
        state.pop("_extra_engines", None)
        state.pop("_extra_key", None)
        state["logger"] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        if self.logger is None:
            self.logger = logging.getLogger(__name__)

    # -- reference surface -----------------------------------------------------------------------------
    def set(self, **kwargs) -> dict:
        """Set parameters; ``chem_pots`` / ``offset_data`` are mirrored onto attributes
        (reference ``calculators.py:448-466``)."""
        changed = _Base.set(self, **kwargs)
        if "chem_pots" in self.parameters:
            self.chem_pots = self.parameters["chem_pots"]
            self.logger.info("chemical potentials: %s are set from parameters", self.chem_pots)
        if "offset_data" in self.parameters:
            self.offset_data = self.parameters["offset_data"]
            self.logger.info("offset data: %s is set from parameters", self.offset_data)
        return changed

    def get_surface_energy(self, atoms=None, chem_pots: dict | None = None, offset_data: dict | None = None) -> float:
        """Surface energy = E_slab - bulk reference - chemical-potential deviation
        (reference ``calculators.py:379-446``).  Raises ValueError when the chemical potentials or the
        offset data are not set."""
        if atoms is None:
            atoms = self.atoms
        if chem_pots is None:
            chem_pots = self.chem_pots
        if not chem_pots:
            raise ValueError("chemical potentials are not set")
        if offset_data is None:
            offset_data = self.offset_data
        if not offset_data:
            raise ValueError("offset data is not set")
        energy = self.get_potential_energy(atoms=atoms)
        return surface_energy_from_energy(energy, atoms.get_chemical_symbols(), chem_pots, offset_data,
                                          self.offset_units)

    def surface_energy_of(self, energy, atoms):
        """Surface energy of a configuration whose potential energy is already known (used by the batched paths and by
        ``mc.ChainEnsemble``); subclasses override the arithmetic (``NFFPourbaix``)."""
        return surface_energy_from_energy(energy, atoms.get_chemical_symbols(),
                                          self._require(self.chem_pots, "chemical potentials"),
                                          self._require(self.offset_data, "offset data"), self.offset_units)

    def _arrays(self, atoms):
        """(numbers, positions, cell, pbc) of an Atoms-like for the ABI, positions rounded as ``position_dtype`` says."""
        Z, pos, cell, pbc = structures.as_arrays(atoms)
        if self.__dict__.get("position_dtype", "float64") == "float32":
            pos = np.asarray(pos, dtype=np.float64).astype(np.float32).astype(np.float64)
        return Z, pos, cell, pbc

    def _fill_results(self, res, b=0):
        a0, a1 = int(res["cfg_start"][b]), int(res["cfg_start"][b + 1])
        return {
            "energy": res["energy"][b:b + 1].copy(),            # shape (1,), like the reference
            "energy_std": res["energy_std"][b:b + 1].copy(),
            "forces": res["forces"][a0:a1].copy(),
            "forces_std": res["forces_std"][a0:a1].copy(),
            "energy_models": res["energy_models"][b].copy(),
            # ensemble-mean per-atom energies in eV (the stoichiometric offset is not distributed over atoms); what the
            # reference's Boltzmann-weighted switch proposal reads from results["per_atom_energies"] (mcmc/slab.py:92)
            "per_atom_energies": res["energy_atoms"][a0:a1].copy(),
            # the evaluation left the range of the fp16-split arithmetic (backend.saturated): finite, but not the model's
            "saturated": bool(res["saturated"][b]) if "saturated" in res else False,
            # the same ensemble mean / spread without the float32 result word (vssr_batch_energy_f64): the reference's
            # results["energy"] is a float32 tensor (calculators.py:484) and stays one; acceptance tests and relaxation drivers
            # of this package compare these
            **({"energy_f64": float(res["energy_f64"][b]), "energy_std_f64": float(res["energy_std_f64"][b])}
               if "energy_f64" in res else {}),
        }

    def calculate(self, atoms=None, properties=implemented_properties, system_changes=all_changes):
        """One ensemble energy+force evaluation (reference ``calculators.py:468-489``)."""
        if atoms is None:
            atoms = self.atoms
        _Base.calculate(self, atoms, properties, system_changes)
        eng = self._get_engine()
        res = eng.evaluate([self._arrays(atoms)])
        # a fresh dict per calculation, like nff's NeuralFF / EnsembleNFF.calculate: entries of an earlier structure that this
        # call does not produce (surface_energy, stress, embedding) must not survive a direct calculate() on another one
        self.results = dict(self._fill_results(res, 0))
        if "stress" in properties or "stress_std" in properties:
            st, sd = eng.stress()
            self.results["stress"], self.results["stress_std"] = st[0], sd[0]
        if "embedding" in self.properties or "embedding" in tuple(properties) and tuple(properties) != tuple(self.implemented_properties):
            emb = eng.embedding()
            self.results["embedding"] = emb[0]
            self.results["embedding_models"] = emb
        if self.results["saturated"]:
            self.logger.warning("activations left the fp16-split range (|x| > 65504): energy %.6g eV is not reliable",
                                float(self.results["energy"][0]))
        if "surface_energy" in properties:
            self.results["surface_energy"] = self.surface_energy_of(self.results["energy"], atoms)
        if hasattr(atoms, "results") and isinstance(atoms.results, dict):
            atoms.results.update(self.results)

    @staticmethod
    def _require(value, what):
        if not value:
            raise ValueError(f"{what} are not set" if what.endswith("s") else f"{what} is not set")
        return value

    # -- new capability: many independent chains in one lock-step evaluation ----------------------------
    def calculate_batch(self, atoms_list, want_surface_energy: bool = False, want_embedding: bool | None = None,
                        want_stress: bool = False) -> list[dict]:
        """Evaluate B independent configurations at once; returns one results dict per configuration
        (``want_embedding``: default = ``"embedding" in self.properties``; ``want_stress``: also ``stress`` / ``stress_std``)."""
        eng = self._get_engine()
        res = eng.evaluate([self._arrays(a) for a in atoms_list])
        stress = eng.stress() if want_stress else None
        if want_embedding is None:
            want_embedding = "embedding" in self.properties
        emb = eng.embedding() if want_embedding else None
        out = []
        for b, atoms in enumerate(atoms_list):
            r = self._fill_results(res, b)
            if want_surface_energy:
                r["surface_energy"] = self.surface_energy_of(r["energy"], atoms)
            if stress is not None:
                r["stress"], r["stress_std"] = stress[0][b].copy(), stress[1][b].copy()
            if emb is not None:
                a0, a1 = int(res["cfg_start"][b]), int(res["cfg_start"][b + 1])
                r["embedding"] = emb[0, a0:a1].copy()
                r["embedding_models"] = emb[:, a0:a1].copy()
            out.append(r)
        return out


    # thresholds of the reference's out-of-bounds guard (mcmc/dynamics.py:17-18,159-168)
    ENERGY_THRESHOLD = 1000.0
    MAX_FORCE_THRESHOLD = 1000.0
    # evaluate_packed: engines (HIP streams) the chains of a batch are split over, and the smallest part worth a stream
    streams = 2
    MIN_CHAINS_PER_STREAM = 32

    def _get_engines(self, n: int):
        """``n`` engines with this calculator's configuration: the first is the calculator's own, the others are created on
        first use (weights are shared on the host, every engine holds its own device copy and workspaces)."""
        first = self._get_engine()
        extra = self.__dict__.setdefault("_extra_engines", [])
        if self.__dict__.get("_extra_key") != self._engine_key:
            for e in extra:
                e.close()
            extra.clear()
            self.__dict__["_extra_key"] = self._engine_key
        table, const = self._offset_config()
        while len(extra) < n - 1:
            extra.append(backend.PainnEngine(
                self.models, device=_device_index(self.device), cutoff=self.cutoff,
                model_units_per_ev=_units_per_ev(self.model_units, self.prediction_units),
                offset_per_z=table, offset_const=const, hparams=self.hparams))
        return [first] + extra[: n - 1]

    @staticmethod
    def packed_supported(relax: bool, optimizer) -> bool:
        """Whether ``evaluate_packed`` serves this request: everything but relaxations with a host-driven optimizer (an optimizer
        class, "BFGSLineSearch", the LAMMPS-style "CG")."""
        return not relax or optimizer is None or (isinstance(optimizer, str)
                                                  and not any(k in optimizer for k in ("CG", "LAMMPS", "BFGSLineSearch")))

    def evaluate_packed(self, n_atoms, Z, pos, cell, pbc, relax: bool = False, fixed_mask=None, relax_steps: int = 20,
                        fmax: float = 0.01, optimizer=None) -> dict:
        """Arrays in, arrays out: the batched evaluation (``relax=False``: ``calculate_batch``) or lock-step relaxation
        (``relax=True``: ``relax_batch`` with a device optimizer, "BFGS" / "FIRE") of B slabs given as the ABI's packed arrays
        (``n_atoms [B]``, ``Z [sum N]``, ``pos [sum N, 3]``, ``cell [B, 9]``, ``pbc [B, 3]``; ``fixed_mask [sum N]`` = FixAtoms).
        No per-slab Python objects on either side -- what ``mc.ChainEnsemble`` calls every MC step.  Returns ``energy [B]``
        (float32, the TRUE energies of the final geometries), ``energy_std``, ``forces``, ``energy_atoms``, ``positions``
        (relaxed, or the input), ``cfg_start``, ``saturated [B]``, ``oob [B]`` (the reference's +-1000 guard,
        ``mcmc/dynamics.py:159-168``; a saturated evaluation counts), and for relaxations ``n_steps`` / ``converged``."""
        if relax and optimizer is None:
            optimizer = self.parameters.get("optimizer", "FIRE")
        n_atoms = np.ascontiguousarray(n_atoms, dtype=np.int64)
        Z = np.asarray(Z)
        B = len(n_atoms)

        def one(eng, c0, c1, a0, a1):
            eng.upload_arrays(n_atoms[c0:c1], Z[a0:a1], pos[a0:a1], cell[c0:c1], pbc[c0:c1])
            inf = None
            if relax:
                inf = eng.relax(optimizer, fixed=None if fixed_mask is None else fixed_mask[a0:a1], max_steps=relax_steps,
                                fmax=fmax)   # (raises for host-driven optimizers)
            else:
                eng.run()
            return eng.download(), inf

        pos = np.asarray(pos, dtype=np.float64).reshape(-1, 3)
        if not relax and self.__dict__.get("position_dtype", "float64") == "float32":
            pos = pos.astype(np.float32).astype(np.float64)   # single points on nff's float32 nxyz (see the class docstring)
        cell, pbc = np.asarray(cell, dtype=np.float64).reshape(B, 9), np.asarray(pbc).reshape(B, 3)
        # (a single lock-step evaluation is too short to pay for the second host thread: measured 15.4 vs 15.0 ms per MC step)
        n_str = self.streams if (relax and B >= 2 * self.MIN_CHAINS_PER_STREAM) else 1
        if n_str <= 1:
            res, info = one(self._get_engine(), 0, B, 0, int(n_atoms.sum()))
        else:
            # the chains split over n_str engines (own HIP streams) that run concurrently: the latency-bound node kernels of one
            # part fill issue slots under the neighbor-sum kernels of the other (+3 % on 256 chains).  ctypes
            # releases the GIL inside the C calls, one host thread per engine drives its relaxation; a chain's results do not
            # depend on the split (bit-identical, tests/test_mc_gpu.py).
            from concurrent.futures import ThreadPoolExecutor

            cb = [(k * B) // n_str for k in range(n_str + 1)]
            ab = np.concatenate([[0], np.cumsum(n_atoms)]).astype(np.int64)
            engines = self._get_engines(n_str)
            with ThreadPoolExecutor(max_workers=n_str) as pool:
                parts = list(pool.map(lambda k: one(engines[k], cb[k], cb[k + 1], int(ab[cb[k]]), int(ab[cb[k + 1]])), range(n_str)))
            res = {k: np.concatenate([p[0][k] for p in parts]) for k in parts[0][0] if k != "cfg_start"}
            res["cfg_start"] = ab
            info = None
            if relax:
                info = {k: np.concatenate([p[1][k] for p in parts]) for k in ("positions", "n_steps", "converged")}
        start = np.asarray(res["cfg_start"], dtype=np.int64)
        fabs = np.abs(res["forces"]).max(axis=1) if len(res["forces"]) else np.zeros(0, np.float32)
        nonempty = start[1:] > start[:-1]
        max_force = np.zeros(len(start) - 1, np.float32)
        if nonempty.any():
            max_force[nonempty] = np.maximum.reduceat(fabs, start[:-1][nonempty])
        energy = res["energy"]
        e64 = np.asarray(res["energy_f64"] if "energy_f64" in res else energy, dtype=np.float64)
        with np.errstate(invalid="ignore"):
            oob = (~np.isfinite(energy)) | (~np.isfinite(max_force)) | (np.abs(energy) > self.ENERGY_THRESHOLD) \
                | (max_force > self.MAX_FORCE_THRESHOLD) | np.asarray(res["saturated"], dtype=bool)
        out = {"energy": energy, "energy_std": res["energy_std"], "forces": res["forces"], "energy_atoms": res["energy_atoms"],
               "positions": info["positions"] if info is not None else np.asarray(pos, dtype=np.float64).reshape(-1, 3),
               "cfg_start": start, "saturated": np.asarray(res["saturated"], dtype=bool), "oob": oob, "energy_f64": e64,
               "energy_std_f64": np.asarray(res["energy_std_f64"] if "energy_std_f64" in res else res["energy_std"], dtype=np.float64)}
        if info is not None:
            out["n_steps"], out["converged"] = info["n_steps"], info["converged"]
        return out

    def relax_batch(self, atoms_list, fixed_indices=None, relax_steps: int = 20, fmax: float = 0.01, optimizer=None,
                    save_traj: bool = False, record_interval: int = 5):
        """Relax B independent slabs at once on the device — the batched counterpart of
        ``optimize_slab(slab, optimizer=..., relax_steps=..., save_traj=..., record_interval=...)`` (reference
        ``mcmc/dynamics.py:83-170``).  ``optimizer``: "BFGS" (ASE BFGS, the reference's SrTiO3 setting,
        ``scripts/configs/sample_config_painn.json:26``), "FIRE" (the reference's default) or "CG" (ASE SciPyFminCG: host-driven scipy
        optimizers over lock-step device evaluations, ``host_opt.py``); None takes
        ``parameters["optimizer"]`` (how ``calc_settings`` reach ``optimize_slab``), else "FIRE".
        ``fixed_indices``: per slab, the atom indices held by FixAtoms (or None).
        Returns, per slab, the reference's tuple ``(relaxed_slab, traj, energy, energy_oob)`` where ``energy_oob`` follows
        the same +-1000 guard (a saturated evaluation counts as out of bounds), plus the results dict of the final
        evaluation.  ``save_traj=True``: ``traj`` is the reference's dict ``{"atoms", "energies", "forces"}`` recorded like
        its TrajectoryObserver attached with ``interval=record_interval`` (after 0, k, 2k, ... optimizer steps; forces with
        FixAtoms applied); otherwise None."""
        eng = self._get_engine()
        packs = [structures.as_arrays(a) for a in atoms_list]
        eng.upload(packs)
        fixed = None
        if fixed_indices is not None:
            fixed = np.zeros(sum(len(p[0]) for p in packs), np.uint8)
            o = 0
            for p, idx in zip(packs, fixed_indices):
                if idx is not None and len(idx):
                    fixed[o + np.asarray(idx, dtype=np.int64)] = 1
                o += len(p[0])
        if optimizer is None:
            optimizer = self.parameters.get("optimizer", "FIRE")
        host_traj = None
        opt_cls = optimizer if callable(optimizer) else None
        if opt_cls is None and "BFGSLineSearch" in str(optimizer):
            try:   # the reference's import (mcmc/dynamics.py:7): only where ASE is installed
                from ase.optimize import BFGSLineSearch as opt_cls
            except Exception as exc:
                raise backend.BackendError('optimizer "BFGSLineSearch" is ASE\'s class and ASE is not importable here; pass an '
                                           "optimizer class, or use BFGS / FIRE / CG") from exc
        if opt_cls is not None:
            # any optimizer of the ASE protocol, one per chain on the host, served by lock-step evaluations (host_opt.py)
            from . import host_opt

            def evaluate(pos_all):
                eng.set_positions(pos_all)
                eng.run()
                r = eng.download()
                return np.asarray(r.get("energy_f64", r["energy"]), dtype=np.float64), np.asarray(r["forces"], dtype=np.float64)

            cfg = np.concatenate([[0], np.cumsum([len(p[0]) for p in packs])])
            kw = {"logfile": None} if HAVE_ASE and not callable(optimizer) else {}
            info = host_opt.optimizer_class_batch(opt_cls, atoms_list, _LockstepProxy, evaluate, cfg, fixed_indices=fixed_indices,
                                                  steps=relax_steps, fmax=fmax, record_interval=int(record_interval) if save_traj else 0,
                                                  optimizer_kwargs=kw)
            info["n_steps"] = np.array([int(getattr(d, "nsteps", -1)) for d in info["optimizers"]], np.int32)   # ASE optimizers count
            info["converged"] = np.zeros(len(packs), bool)   # judged below from the final forces
            class_traj = info["traj"]
            info["traj"] = None
            eng.set_positions(info["positions"])
            eng.run()
        elif "CG" in str(optimizer) and "LAMMPS" not in str(optimizer):
            # the reference maps "CG" to ASE's SciPyFminCG (mcmc/dynamics.py:123-124): scipy owns the control flow, so every chain's
            # optimizer runs on the host and their energy / force requests are served by lock-step evaluations (host_opt.py)
            from . import host_opt

            def evaluate(pos_all):
                eng.set_positions(pos_all)
                eng.run()
                r = eng.download()
                return np.asarray(r.get("energy_f64", r["energy"]), dtype=np.float64), np.asarray(r["forces"], dtype=np.float64)

            cfg = np.concatenate([[0], np.cumsum([len(p[0]) for p in packs])])
            pos0 = np.concatenate([np.asarray(p[1], dtype=np.float64).reshape(-1, 3) for p in packs])
            info = host_opt.scipy_cg_batch(evaluate, cfg, pos0, fixed=fixed, steps=relax_steps, fmax=fmax,
                                           record_interval=int(record_interval) if save_traj else 0)
            host_traj = info["traj"]
            eng.set_positions(info["positions"])
            eng.run()
        else:
            info = eng.relax(optimizer, fixed=fixed, max_steps=relax_steps, fmax=fmax,
                             record_interval=int(record_interval) if save_traj else 0)
        res = eng.download()
        out = []
        for b, atoms in enumerate(atoms_list):
            a0, a1 = int(res["cfg_start"][b]), int(res["cfg_start"][b + 1])
            relaxed = atoms.copy()
            relaxed.set_positions(info["positions"][a0:a1])
            if opt_cls is not None:
                traj = class_traj[b] if (save_traj and class_traj is not None) else None
            elif host_traj is not None:
                frames = []
                for fpos, _, _ in host_traj[b]:
                    frame = atoms.copy()
                    frame.set_positions(fpos)
                    if hasattr(frame, "calc"):
                        frame.calc = None
                    frames.append(frame)
                traj = {"atoms": frames, "energies": [float(e) for _, e, _ in host_traj[b]],
                        "forces": [np.asarray(f, dtype=np.float32) for _, _, f in host_traj[b]]}
            else:
                traj = _traj_of_chain(info.get("traj"), b, a0, a1, atoms) if save_traj else None
            r = self._fill_results(res, b)
            energy = float(r["energy_f64"]) if "energy_f64" in r else float(r["energy"][0])
            max_force = float(np.abs(r["forces"]).max()) if a1 > a0 else 0.0
            oob = bool(not np.isfinite(energy) or not np.isfinite(max_force) or abs(energy) > self.ENERGY_THRESHOLD
                       or max_force > self.MAX_FORCE_THRESHOLD or r["saturated"])
            if oob:
                energy = self.ENERGY_THRESHOLD
            r["n_steps"] = int(info["n_steps"][b])
            r["converged"] = bool(info["converged"][b])
            if opt_cls is not None:   # ASE's criterion on the final forces (constraints applied)
                ff = np.array(r["forces"], dtype=np.float64, copy=True)
                if fixed_indices is not None and fixed_indices[b] is not None and len(fixed_indices[b]):
                    ff[np.asarray(fixed_indices[b], dtype=np.int64)] = 0.0
                r["converged"] = bool((ff ** 2).sum(axis=1).max() < fmax ** 2) if len(ff) else True
            out.append((relaxed, traj, energy, oob, r))
        return out


def surface_energy_from_counts(energy, counts: dict, chem_pots: dict, offset_data: dict, offset_units: str = "atomic"):
    """:func:`surface_energy_from_energy` for MANY slabs at once: ``energy`` ``[B]`` and ``counts`` = element symbol ->
    ``[B]`` atom counts, the dict's key order being the order in which the elements first appear in the slabs' atom lists
    (the scalar function walks a ``Counter`` in that order).  Same operations in the same order per slab, so every entry
    equals the scalar result bit for bit (``tests/test_mc.py``)."""
    energy = np.asarray(energy, dtype=np.float64)
    bulk_energies, stoics = offset_data["bulk_energies"], offset_data["stoics"]
    ref_formula, ref_element = offset_data["ref_formula"], offset_data["ref_element"]
    n = {k: np.asarray(v, dtype=np.float64) for k, v in counts.items()}
    n_ref = n.get(ref_element, np.zeros_like(energy))
    bulk_ref_en = n_ref * bulk_energies[ref_formula]
    for ele, n_e in n.items():
        if ele != ref_element:
            bulk_ref_en = bulk_ref_en + (n_e - stoics[ele] / stoics[ref_element] * n_ref) * bulk_energies[ele]
    surface_energy = energy - (bulk_ref_en * HARTREE_TO_EV if offset_units == "atomic" else bulk_ref_en)
    pot = np.zeros_like(energy)
    for ele, n_e in n.items():
        if ele != ref_element:
            pot = pot + (n_e - stoics[ele] / stoics[ref_element] * n_ref) * chem_pots[ele]
    return surface_energy - pot


def _traj_of_chain(traj, b, a0, a1, atoms):
    """Records of chain b in the layout of the reference's ``optimize_slab`` (``mcmc/dynamics.py:145-151``)."""
    if traj is None:
        return None
    frames = []
    n = int(traj["n_records"][b])
    for r in range(n):
        frame = atoms.copy()
        frame.set_positions(traj["positions"][r, a0:a1])
        if hasattr(frame, "calc"):
            frame.calc = None   # the observer stores copies without the calculator
        frames.append(frame)
    return {"atoms": frames, "energies": [float(traj["energies"][r, b]) for r in range(n)],
            "forces": [traj["forces"][r, a0:a1].copy() for r in range(n)]}


# ---- helpers of the reference's clustering / uncertainty scripts (mcmc/calculators/calculators.py:34-135) ------------------
def get_results_single(atoms_batch, calc) -> dict:
    """One calculation of ``atoms_batch`` with ``calc``; returns ``calc.results`` (reference ``:34-47``)."""
    atoms_batch.calc = calc
    calc.calculate(atoms_batch)
    return calc.results


def get_embeddings_single(atoms_batch, calc, results_cache: dict | None = None, flatten: bool = True,
                          flatten_axis: int = 0) -> np.ndarray:
    """Latent-space embedding of one structure (reference ``:67-93``): the per-atom features ``results["embedding"]``
    averaged over ``flatten_axis`` (atoms) when ``flatten``; ``results_cache`` avoids a second evaluation.  The calculator
    must provide the embedding (``EnsembleNFFSurface(..., properties=[..., "embedding"])``)."""
    results = results_cache if results_cache is not None and "embedding" in results_cache \
        else get_results_single(atoms_batch, calc)
    if "embedding" not in results:
        raise KeyError('the calculator returns no "embedding": construct it with properties=(..., "embedding")')
    emb = np.asarray(results["embedding"])
    return emb.mean(axis=flatten_axis).squeeze() if flatten else emb.squeeze()


def get_embeddings(atoms_batches, calc) -> np.ndarray:
    """One embedding row per structure (reference ``:50-64``)."""
    return np.stack([get_embeddings_single(a, calc) for a in atoms_batches])


def get_std_devs_single(atoms_batch, calc):
    """Mean ensemble standard deviation of the force components of one structure, 0.0 for a single model (reference
    ``:117-135``)."""
    if len(calc.models) > 1:
        atoms_batch.calc = calc
        calc.calculate(atoms_batch)
        return calc.results.get("forces_std", np.array([0.0])).mean()
    return 0.0


def get_std_devs(atoms_batches, calc) -> np.ndarray:
    """Force standard deviation of every structure (reference ``:96-114``)."""
    return np.stack([get_std_devs_single(a, calc) for a in atoms_batches])


@dataclasses.dataclass
class PourbaixAtom:
    """Per-element data of the Pourbaix dissolution reaction (reference ``mcmc/pourbaix/atoms.py:25-66``)."""

    symbol: str
    dominant_species: str = ""
    species_conc: float = 1e-6
    num_e: int = 0
    num_H: int = 0
    atom_std_state_energy: float = 0.0
    delta_G2_std: float = 0.0


def _formula_counts(formula: str) -> Counter:
    """Element counts of a plain formula string such as ``"OH"`` or ``"H2O"``."""
    out = Counter()
    for sym, num in re.findall(r"([A-Z][a-z]?)(\d*)", formula):
        out[sym] += int(num) if num else 1
    return out


def pourbaix_delta_G1(energy: float, symbols, pourbaix_atoms: dict, adsorbate_corrections: dict | None = None) -> float:
    """``NFFPourbaix.get_delta_G1`` (reference ``calculators.py:235-272``): ``sum_atoms E_std(atom) - (E_slab + adsorbate
    corrections)``.  Adsorbate corrections count how many times the adsorbate formula fits into the slab formula; for O-H
    adsorbates the hydrogens in excess of the oxygens are first attributed to water and removed (``:254-268``)."""
    counts = Counter(symbols)
    sum_std = sum(n * pourbaix_atoms[a].atom_std_state_energy for a, n in counts.items())
    slab_energy = float(np.ravel(energy)[0])
    formula = Counter(counts)
    for adsorbate, correction in (adsorbate_corrections or {}).items():
        if "O" in adsorbate and "H" in adsorbate:
            ho_diff = max(formula["H"] - formula["O"], 0)
            if ho_diff > 0:
                formula = Counter({k: v - {"H": 2, "O": 1}.get(k, 0) * ho_diff for k, v in formula.items()})
        ads = _formula_counts(adsorbate)
        div = min(formula[k] // n for k, n in ads.items()) if ads else 0
        slab_energy += div * correction
    return sum_std - slab_energy


def pourbaix_delta_G2(symbols, pourbaix_atoms: dict, temp: float = 0.0257, phi: float = 0.0, pH: float = 7.0) -> float:
    """``NFFPourbaix.get_delta_G2`` (reference ``calculators.py:197-233``): ``sum_atoms [dG2_std - n_e phi - ln(10) n_H kT pH +
    kT ln(conc)]``."""
    dg2 = 0.0
    for a, n in Counter(symbols).items():
        pa = pourbaix_atoms[a]
        dg2 += n * (pa.delta_G2_std - pa.num_e * phi - np.log(10) * pa.num_H * temp * pH + temp * np.log(pa.species_conc))
    return dg2


def pourbaix_potential_from_energy(energy: float, symbols, pourbaix_atoms: dict, temp: float = 0.0257, phi: float = 0.0,
                                   pH: float = 7.0, adsorbate_corrections: dict | None = None) -> float:
    """Pourbaix ("grand") potential of a slab with known potential energy — the arithmetic of the reference's
    ``NFFPourbaix`` (``calculators.py:197-305``): ``-(dG1 + dG2)`` with :func:`pourbaix_delta_G1` and :func:`pourbaix_delta_G2`."""
    return -(pourbaix_delta_G1(energy, symbols, pourbaix_atoms, adsorbate_corrections)
             + pourbaix_delta_G2(symbols, pourbaix_atoms, temp, phi, pH))


class NFFPourbaix(EnsembleNFFSurface):
    """Surface calculator whose ``surface_energy`` is the Pourbaix potential (reference ``NFFPourbaix``,
    ``calculators.py:121-357``; there a single-model ``NeuralFF``, here any ensemble size on the same engine).
    Parameters through ``set``: ``temperature`` (kT in eV), ``phi``, ``pH``, ``pourbaix_atoms`` (symbol ->
    ``PourbaixAtom``), ``adsorbate_corrections`` (formula -> eV)."""

    implemented_properties = (*EnsembleNFFSurface.implemented_properties, "pourbaix_potential")
    name = "nff_pourbaix_mi355x"

    def __init__(self, *args, temp: float = 0.0257, phi: float = 0.0, pH: float = 7.0, **kwargs):
        super().__init__(*args, **kwargs)
        self.temp, self.phi, self.pH = temp, phi, pH
        self.pourbaix_atoms: dict = {}
        self.adsorbate_corrections: dict = {}

    def set(self, **kwargs) -> dict:
        changed = super().set(**kwargs)
        for key, attr in (("temperature", "temp"), ("phi", "phi"), ("pH", "pH"), ("pourbaix_atoms", "pourbaix_atoms"),
                          ("adsorbate_corrections", "adsorbate_corrections")):
            if key in self.parameters:
                setattr(self, attr, self.parameters[key])
        return changed

    def get_delta_G2_individual(self, atom) -> float:
        pa = self.pourbaix_atoms[atom] if isinstance(atom, str) else atom
        return pa.delta_G2_std - pa.num_e * self.phi - np.log(10) * pa.num_H * self.temp * self.pH \
            + self.temp * np.log(pa.species_conc)

    def get_delta_G2(self, atoms=None) -> float:
        atoms = self.atoms if atoms is None else atoms
        return sum(self.get_delta_G2_individual(a) for a in atoms.get_chemical_symbols())

    def get_delta_G1(self, atoms=None) -> float:
        """Dissociation energy of all atoms incl. the adsorbate corrections (reference ``calculators.py:235-272``)."""
        atoms = self.atoms if atoms is None else atoms
        return pourbaix_delta_G1(self.get_potential_energy(atoms=atoms), atoms.get_chemical_symbols(),
                                 self._require(self.pourbaix_atoms, "Pourbaix atoms"), self.adsorbate_corrections)

    def surface_energy_of(self, energy, atoms):
        return pourbaix_potential_from_energy(energy, atoms.get_chemical_symbols(),
                                              self._require(self.pourbaix_atoms, "Pourbaix atoms"), self.temp, self.phi,
                                              self.pH, self.adsorbate_corrections)

    def get_pourbaix_potential(self, atoms=None) -> float:
        atoms = self.atoms if atoms is None else atoms
        return self.surface_energy_of(self.get_potential_energy(atoms=atoms), atoms)

    def get_surface_energy(self, atoms=None, **kwargs) -> float:
        return self.get_pourbaix_potential(atoms)

    def calculate(self, atoms=None, properties=implemented_properties, system_changes=all_changes):
        super().calculate(atoms, properties, system_changes)
        if "pourbaix_potential" in properties or "surface_energy" in properties:
            self.results["pourbaix_potential"] = self.results.get(
                "surface_energy", self.surface_energy_of(self.results["energy"], self.atoms if atoms is None else atoms))


class _AnalyticSurfCalc(_Base):
    """Shared front end of the analytic potentials that the reference evaluates through LAMMPS (Tersoff, EAM): fp64
    ``energy`` / ``per_atom_energies`` / ``forces`` from the device, ``surface_energy`` = potential energy
    (reference ``calculators.py:707-719,766-777``), batched evaluation and lock-step relaxation for ``mc.ChainEnsemble``."""

    implemented_properties = ("energy", "relaxed_energy", "forces", "per_atom_energies", "surface_energy")
    species: list = []

    def _init_common(self, device, all_periodic, logger):
        self.device = device
        self.all_periodic = bool(all_periodic)
        self.run_dir = None
        self.relax_steps = 100
        self.logger = logger or logging.getLogger(__name__)
        self._engine = None

    def _make_engine(self):
        raise NotImplementedError

    def _get_engine(self):
        if self._engine is None:
            self._engine = self._make_engine()
        return self._engine

    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, None if k == "_engine" else (v if k in ("params", "funcfl", "logger") else copy.deepcopy(v, memo)))
        return new

    def set(self, **kwargs) -> dict:
        changed = _Base.set(self, **kwargs)
        if "run_dir" in self.parameters:
            self.run_dir = self.parameters["run_dir"]
        if "relax_steps" in self.parameters:
            self.relax_steps = self.parameters["relax_steps"]
        return changed

    def _pack(self, atoms):
        Z, pos, cell, pbc = structures.as_arrays(atoms)
        idx = {structures.ATOMIC_NUMBERS[s]: t for t, s in enumerate(self.species)}
        try:
            types = np.array([idx[int(z)] for z in Z], dtype=np.int32)
        except KeyError as e:
            raise ValueError(f"element Z={e.args[0]} is not covered by the potential {self.species}") from None
        if self.all_periodic:
            pbc = np.ones(3, np.uint8)
        return types, pos, cell, pbc

    def get_surface_energy(self, atoms=None) -> float:
        """Currently the same as the potential energy (reference ``calculators.py:707-719``)."""
        if atoms is None:
            atoms = self.atoms
        return self.get_potential_energy(atoms=atoms)

    def calculate(self, atoms=None, properties=implemented_properties, system_changes=all_changes):
        if atoms is None:
            atoms = self.atoms
        _Base.calculate(self, atoms, properties, system_changes)
        e, ea, f = self._get_engine().evaluate_f64([self._pack(atoms)])
        self.results["energy"] = float(e[0])
        self.results["per_atom_energies"] = ea
        self.results["forces"] = f
        if "relaxed_energy" in properties:   # reference calculators.py:688-691
            _, e_rel, ea_rel = self.run_lammps_opt(atoms)
            self.results["relaxed_energy"] = e_rel
            self.results["per_atom_energies"] = ea_rel
        if "surface_energy" in properties:
            self.results["surface_energy"] = self.results["energy"]

    def run_lammps_energy(self, slab, run_dir=None, **kwargs):
        """``LAMMMPSCalc.run_lammps_energy`` (reference ``calculators.py:621-640``: the static energy through the run
        directory's ``lammps_energy_template.txt``): one device evaluation; returns the reference's tuple
        ``(slab, energy, per_atom_energies)``."""
        e, ea, _ = self._get_engine().evaluate_f64([self._pack(slab)])
        return slab, float(e[0]), ea

    def run_lammps_opt(self, slab, run_dir=None, fixed_indices=None, optimizer="CG", **kwargs):
        """Relaxation counterpart of ``LAMMMPSCalc.run_lammps_opt`` (reference ``calculators.py:600-619``: LAMMPS
        ``min_style cg``, ``minimize 1e-5 1e-5 {relax_steps} 10000``, bulk atoms with ``setforce 0``): the same minimiser on
        the device (``vssr_batch_relax_cg``; ``etol`` / ``ftol`` keywords override 1e-5).  ``optimizer="FIRE"`` / ``"BFGS"``
        select the ASE-style optimizers with ``fmax`` (default 0.01) instead.  Returns the reference's tuple
        ``(relaxed_slab, energy, per_atom_energies)``; ``self.last_opt`` holds iteration / evaluation counts and the stop
        reason."""
        types, pos, cell, pbc = self._pack(slab)
        fixed = None
        if fixed_indices is not None and len(fixed_indices):
            fixed = np.zeros(len(types), np.uint8)
            fixed[np.asarray(fixed_indices, dtype=np.int64)] = 1
        eng = self._get_engine()
        if str(optimizer).upper() in ("CG", "LAMMPS"):
            e, ea, f, new_pos, it, ev, why = eng.relax_cg_f64(
                [(types, pos, cell, pbc)], fixed=fixed, max_iter=int(self.relax_steps), etol=kwargs.get("etol", 1e-5),
                ftol=kwargs.get("ftol", 1e-5))
            self.last_opt = {"optimizer": "CG", "iterations": int(it[0]), "evaluations": int(ev[0]),
                             "stop": backend.CG_STOP_REASONS.get(int(why[0]), str(int(why[0])))}
        else:
            e, ea, f, new_pos, steps, conv = eng.relax_f64(
                [(types, pos, cell, pbc)], fixed=fixed, max_steps=int(self.relax_steps), fmax=kwargs.get("fmax", 0.01),
                optimizer=optimizer)
            self.last_opt = {"optimizer": str(optimizer), "iterations": int(steps[0]), "converged": bool(conv[0])}
        relaxed = slab.copy()
        relaxed.set_positions(new_pos)
        relaxed.calc = getattr(slab, "calc", None)
        return relaxed, float(e[0]), ea

    ENERGY_THRESHOLD = 1000.0
    MAX_FORCE_THRESHOLD = 1000.0

    def relax_batch(self, atoms_list, fixed_indices=None, relax_steps: int | None = None, fmax: float = 0.01,
                    optimizer=None, **kwargs):
        """Relax B slabs in ONE lock-step call -- what ``mc.ChainEnsemble(relax=True)`` needs from a calculator.  Default
        ``optimizer`` "LAMMPS" / "CG": the reference's GaN minimiser (``optimize_slab(optimizer="LAMMPS")`` ->
        ``run_lammps_opt``, ``mcmc/dynamics.py:107-116``) for all slabs at once (``vssr_batch_relax_cg``); "FIRE" / "BFGS"
        use the ASE-style optimizers with ``fmax``.  ``relax_steps`` defaults to ``self.relax_steps`` (the reference takes it
        from ``calc.relax_steps``).  Returns per slab ``(relaxed, None, energy, energy_oob, results)`` like
        ``EnsembleNFFSurface.relax_batch``; results carry ``per_atom_energies`` of the relaxed slab."""
        if optimizer is None:
            optimizer = self.parameters.get("optimizer", "LAMMPS")
        steps = int(self.relax_steps if relax_steps is None else relax_steps)
        packs = [self._pack(a) for a in atoms_list]
        fixed = None
        if fixed_indices is not None:
            fixed = np.zeros(sum(len(p[0]) for p in packs), np.uint8)
            o = 0
            for p, idx in zip(packs, fixed_indices):
                if idx is not None and len(idx):
                    fixed[o + np.asarray(idx, dtype=np.int64)] = 1
                o += len(p[0])
        eng = self._get_engine()
        if str(optimizer).upper() in ("CG", "LAMMPS"):
            e, ea, f, pos, it, ev, why = eng.relax_cg_f64(packs, fixed=fixed, max_iter=steps, etol=kwargs.get("etol", 1e-5),
                                                          ftol=kwargs.get("ftol", 1e-5))
            extra = [{"iterations": int(it[b]), "evaluations": int(ev[b]),
                      "stop": backend.CG_STOP_REASONS.get(int(why[b]), str(int(why[b])))} for b in range(len(packs))]
        else:
            e, ea, f, pos, nst, conv = eng.relax_f64(packs, fixed=fixed, max_steps=steps, fmax=fmax, optimizer=optimizer)
            extra = [{"n_steps": int(nst[b]), "converged": bool(conv[b])} for b in range(len(packs))]
        out, o = [], 0
        for b, (atoms, p) in enumerate(zip(atoms_list, packs)):
            n = len(p[0])
            relaxed = atoms.copy()
            relaxed.set_positions(pos[o:o + n])
            energy = float(e[b])
            max_force = float(np.abs(f[o:o + n]).max()) if n else 0.0
            oob = bool(not np.isfinite(energy) or not np.isfinite(max_force) or abs(energy) > self.ENERGY_THRESHOLD
                       or max_force > self.MAX_FORCE_THRESHOLD)
            r = {"energy": energy, "per_atom_energies": ea[o:o + n].copy(), "forces": f[o:o + n].copy(), **extra[b]}
            out.append((relaxed, None, self.ENERGY_THRESHOLD if oob else energy, oob, r))
            o += n
        return out

    def calculate_batch(self, atoms_list) -> list[dict]:
        packs = [self._pack(a) for a in atoms_list]
        e, ea, f = self._get_engine().evaluate_f64(packs)
        out, o = [], 0
        for b, p in enumerate(packs):
            n = len(p[0])
            out.append({"energy": float(e[b]), "per_atom_energies": ea[o:o + n].copy(), "forces": f[o:o + n].copy()})
            o += n
        return out


    @staticmethod
    def packed_supported(relax: bool, optimizer) -> bool:
        """``evaluate_packed`` relaxes with the device optimizers only: "LAMMPS" / "CG" (the default), "FIRE", "BFGS"."""
        return not relax or optimizer is None or (isinstance(optimizer, str)
                                                  and optimizer.upper() in ("CG", "LAMMPS", "FIRE", "BFGS"))

    def _types_of(self, Z) -> np.ndarray:
        """LAMMPS types of atomic numbers (the order of ``species``), vectorised."""
        table = np.full(len(structures.SYMBOLS) + 1, -1, np.int32)
        for t, sym in enumerate(self.species):
            table[structures.ATOMIC_NUMBERS[sym]] = t
        Z = np.asarray(Z, dtype=np.int64)
        types = table[np.clip(Z, 0, len(table) - 1)]
        if len(types) and types.min() < 0:
            raise ValueError(f"element Z={int(Z[np.argmin(types)])} is not covered by the potential {self.species}")
        return types

    def evaluate_packed(self, n_atoms, Z, pos, cell, pbc, relax: bool = False, fixed_mask=None, relax_steps: int | None = None,
                        fmax: float = 0.01, optimizer=None, **kwargs) -> dict:
        """``calculate_batch`` (``relax=False``) / ``relax_batch`` (``relax=True``) with the ABI's packed arrays on both sides
        (``n_atoms [B]``, atomic numbers ``Z [sum N]``, ``pos [sum N, 3]``, ``cell [B, 9]``, ``pbc [B, 3]``, ``fixed_mask
        [sum N]``): what ``mc.ChainEnsemble`` calls every MC step -- no per-slab objects (4 096 GaN chains: the per-slab
        bookkeeping was 41 % of an MC step, ``profiles/r04/bench_gan.jsonl``).  Returns fp64 ``energy [B]`` (the static energies of
        the final geometries), ``forces``, ``energy_atoms``, ``positions``, ``cfg_start``, ``oob [B]`` (the +-1000 guard of
        ``mcmc/dynamics.py:159-168``), ``energy_std`` / ``saturated`` (zeros: one deterministic fp64 potential), and for
        relaxations ``iterations`` / ``evaluations`` / ``stop`` (CG) or ``n_steps`` / ``converged`` (FIRE, BFGS)."""
        if optimizer is None:
            optimizer = self.parameters.get("optimizer", "LAMMPS")
        if relax and not self.packed_supported(True, optimizer):
            raise backend.BackendError(f"evaluate_packed relaxes on the device only (CG / LAMMPS, FIRE, BFGS), not with {optimizer!r}")
        n_atoms = np.ascontiguousarray(n_atoms, dtype=np.int32)
        B = len(n_atoms)
        types = self._types_of(Z)
        pos = np.asarray(pos, dtype=np.float64).reshape(-1, 3)
        cell = np.asarray(cell, dtype=np.float64).reshape(B, 9)
        pbc = np.ones((B, 3), np.uint8) if self.all_periodic else np.asarray(pbc).astype(np.uint8).reshape(B, 3)
        eng = self._get_engine()
        extra = {}
        if not relax:
            e, ea, f = eng.evaluate_arrays_f64(n_atoms, types, pos, cell, pbc)
            new_pos = pos
        elif str(optimizer).upper() in ("CG", "LAMMPS"):
            steps = int(self.relax_steps if relax_steps is None else relax_steps)
            e, ea, f, new_pos, it, ev, why = eng.relax_cg_arrays_f64(n_atoms, types, pos, cell, pbc, fixed=fixed_mask, max_iter=steps,
                                                                     etol=kwargs.get("etol", 1e-5), ftol=kwargs.get("ftol", 1e-5))
            ls, ce = getattr(eng, "last_relax_counts", (0, 0))
            extra = {"iterations": it, "evaluations": ev, "stop": why, "lockstep_evaluations": ls, "dispatched_chain_evaluations": ce}
        else:
            steps = int(self.relax_steps if relax_steps is None else relax_steps)
            e, ea, f, new_pos, nst, conv = eng.relax_arrays_f64(n_atoms, types, pos, cell, pbc, fixed=fixed_mask, max_steps=steps,
                                                                fmax=fmax, optimizer=optimizer)
            extra = {"n_steps": nst, "converged": conv}
        start = np.concatenate([[0], np.cumsum(n_atoms)]).astype(np.int64)
        fabs = np.abs(f).max(axis=1) if len(f) else np.zeros(0)
        nonempty = start[1:] > start[:-1]
        max_force = np.zeros(B)
        if nonempty.any():
            max_force[nonempty] = np.maximum.reduceat(fabs, start[:-1][nonempty])
        with np.errstate(invalid="ignore"):
            oob = (~np.isfinite(e)) | (~np.isfinite(max_force)) | (np.abs(e) > self.ENERGY_THRESHOLD) \
                | (max_force > self.MAX_FORCE_THRESHOLD)
        return {"energy": e, "energy_std": np.zeros(B), "forces": f, "energy_atoms": ea, "positions": new_pos, "cfg_start": start,
                "saturated": np.zeros(B, bool), "oob": oob, **extra}


class TersoffSurfCalc(_AnalyticSurfCalc):
    """Tersoff energy / per-atom energies / forces on MI355X (drop-in for ``LAMMPSSurfCalc`` with
    ``pair_style tersoff``, reference ``calculators.py:492-752``): ``energy`` is the static energy,
    ``per_atom_energies`` is LAMMPS' ``pe/atom``; periodic in all directions like the reference's
    ``boundary p p p`` template."""

    name = "tersoff_mi355x"

    def __init__(self, potential, species, device="cuda", all_periodic=True, logger=None, **kwargs):
        """potential: path or text of a LAMMPS tersoff file, or a params array [nt,nt,nt,14];
        species: symbols in LAMMPS type order (e.g. ["Ga", "N"])."""
        if isinstance(potential, np.ndarray):
            self.params = np.ascontiguousarray(potential, dtype=np.float64)
        else:
            text = potential
            if "\n" not in str(potential):
                with open(potential) as fh:
                    text = fh.read()
            self.params = tersoff_io.parse_tersoff(text, list(species))
        self.species = list(species)
        self._init_common(device, all_periodic, logger)
        super().__init__(**kwargs)

    def _make_engine(self):
        return backend.TersoffEngine(self.params, device=_device_index(self.device))


class EAMSurfCalc(_AnalyticSurfCalc):
    """One-element EAM on MI355X: drop-in for ``LAMMPSRunSurfCalc`` (reference ``calculators.py:755-811``, a modified ASE
    ``lammpsrun`` that pipes ``pair_style eam`` / ``pair_coeff * * Cu_u3.eam`` to an ``lmp`` subprocess; used by
    ``tests/test_Cu.py`` and ``tutorials/example.ipynb``).  Constructor keeps the reference's keywords: ``files=[potential
    file]`` (funcfl), ``keep_tmp_files`` / ``keep_alive`` / ``tmp_dir`` are accepted and unused (nothing is written to
    disk); ``set(pair_style="eam", pair_coeff=[...])`` is recorded in ``parameters``.  Boundary conditions follow the
    atoms' ``pbc`` (ASE lammpsrun derives ``boundary`` from it)."""

    name = "eam_mi355x"

    def __init__(self, files=None, potential=None, device="cuda", all_periodic=False, logger=None, keep_tmp_files=False,
                 keep_alive=False, tmp_dir=None, **kwargs):
        from . import eam as eam_io

        src = potential if potential is not None else (list(files)[0] if files else None)
        if src is None:
            raise ValueError("EAMSurfCalc needs files=[<funcfl potential file>]")
        self.funcfl = src if isinstance(src, eam_io.Funcfl) else (
            eam_io.parse_funcfl(src) if "\n" in str(src) else eam_io.read_funcfl(src))
        self.species = [structures.SYMBOLS[self.funcfl.atomic_number]]
        self._init_common(device, all_periodic, logger)
        super().__init__(**kwargs)

    def set(self, **kwargs) -> dict:
        style = kwargs.get("pair_style")
        if style is not None and str(style).split()[0] != "eam":
            raise ValueError(f"pair_style {style!r}: this calculator evaluates `pair_style eam` (funcfl) only")
        return super().set(**kwargs)

    def _make_engine(self):
        return backend.EAMEngine(self.funcfl, device=_device_index(self.device))


LAMMPSRunSurfCalc = EAMSurfCalc   # the reference's class name for this role


class LAMMPSSurfCalc(_AnalyticSurfCalc):
    """Drop-in for the reference's ``LAMMPSSurfCalc`` / ``LAMMMPSCalc`` (``calculators.py:492-752``) the way its GaN tutorial
    uses them: constructed WITHOUT arguments and configured through ``set(run_dir=..., relax_steps=..., optimizer="LAMMPS",
    ...)`` (``tutorials/GaN_0001.ipynb``: ``lammps_surf_calc = LAMMPSSurfCalc(); lammps_surf_calc.set(**calc_settings)``).
    Instead of writing ``lammps.in`` / ``lammps.data`` and calling LAMMPS, the run directory's own files say what to evaluate:

    * ``lammps_config.json``: ``potential_file``, ``atoms`` (species in LAMMPS type order), ``bulk_index`` (atoms with id <=
      bulk_index form the group the relaxation template holds with ``fix ... setforce 0``);
    * ``lammps_energy_template.txt`` / ``lammps_opt_template.txt``: the ``pair_style`` line selects the potential -- ``tersoff``
      (multi-element file) or ``eam`` (one funcfl element) are evaluated on the device, anything else raises.

    The potential file is looked up like LAMMPS does (as given, in the run directory, in the working directory, in
    ``$LAMMPS_POTENTIALS``) and additionally in ``potential_dirs`` and in the reference's ``mcmc/potentials`` when that package
    is importable.  ``boundary p p p`` of the templates = periodic in all directions."""

    name = "lammps_surf_mi355x"

    def __init__(self, device="cuda", potential_dirs=(), logger=None, **kwargs):
        self.potential_dirs = [str(d) for d in potential_dirs]
        self.species = []
        self.bulk_index = 0
        self.pair_style = None
        self._cfg_key = None
        self._init_common(device, True, logger)
        self.run_dir = os.getcwd()        # the reference's default (calculators.py:502)
        super().__init__(**kwargs)

    # -- configuration from the run directory ----------------------------------------------------------------------------------
    def _find_potential(self, name, run_dir):
        cands = [name] if os.path.isabs(name) else [os.path.join(d, name) for d in
                                                    [str(run_dir), os.getcwd(),
                                                     *[x for x in os.environ.get("LAMMPS_POTENTIALS", "").split(os.pathsep) if x],
                                                     *self.potential_dirs]]
        if not os.path.isabs(name):
            try:
                import importlib.util

                spec = importlib.util.find_spec("mcmc")
                for loc in (spec.submodule_search_locations or []) if spec else []:
                    cands.append(os.path.join(loc, "potentials", name))
            except (ImportError, ValueError):
                pass
        for c in cands:
            if os.path.isfile(c):
                return c
        raise FileNotFoundError(f"potential file {name!r} not found; looked in: " + ", ".join(cands))

    def _configure(self):
        run_dir = str(self.run_dir)
        cfg_path = os.path.join(run_dir, "lammps_config.json")
        if not os.path.isfile(cfg_path):
            raise FileNotFoundError(f"{cfg_path}: the run directory needs the reference's lammps_config.json "
                                    "(potential_file, atoms, bulk_index)")
        with open(cfg_path, encoding="utf-8") as fh:
            cfg = json.load(fh)
        style = None
        for tmpl in ("lammps_energy_template.txt", "lammps_opt_template.txt"):
            tp = os.path.join(run_dir, tmpl)
            if os.path.isfile(tp):
                with open(tp, encoding="utf-8") as fh:
                    m = re.search(r"^\s*pair_style\s+(\S+)", fh.read(), re.M)
                if m:
                    style = m.group(1)
                    break
        if style is None:
            raise FileNotFoundError(f"{run_dir}: no lammps_energy_template.txt / lammps_opt_template.txt with a pair_style line")
        if self.parameters.get("kim_potential") or style.startswith("kim"):
            raise backend.BackendError("KIM potentials are not provided by this backend (tersoff and eam/funcfl are)")
        pot = self._find_potential(cfg["potential_file"], run_dir)
        key = (cfg_path, os.path.getmtime(cfg_path), pot, style)
        if key == self._cfg_key:
            return
        self.species = list(cfg["atoms"])
        self.bulk_index = int(cfg.get("bulk_index", 0))
        self.pair_style = style
        with open(pot, encoding="utf-8") as fh:
            text = fh.read()
        if style == "tersoff":
            self.params = tersoff_io.parse_tersoff(text, self.species)
            self.funcfl = None
        elif style == "eam":
            from . import eam as eam_io

            self.funcfl = eam_io.parse_funcfl(text)
            self.params = None
            if self.species != [structures.SYMBOLS[self.funcfl.atomic_number]]:
                raise ValueError(f"pair_style eam (funcfl) holds one element, lammps_config.json lists {self.species}")
        else:
            raise backend.BackendError(f"pair_style {style!r} is not provided by this backend (tersoff and eam are)")
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        self._cfg_key = key

    def _make_engine(self):
        if self.pair_style == "tersoff":
            return backend.TersoffEngine(self.params, device=_device_index(self.device))
        return backend.EAMEngine(self.funcfl, device=_device_index(self.device))

    def _get_engine(self):
        self._configure()
        return super()._get_engine()

    def _pack(self, atoms):
        self._configure()
        return super()._pack(atoms)

    def run_lammps_opt(self, slab, run_dir=None, fixed_indices=None, optimizer="CG", **kwargs):
        """As the base class; without ``fixed_indices`` the template's ``group bulk id <= bulk_index`` + ``setforce 0`` applies:
        the first ``bulk_index`` atoms are held."""
        if run_dir is not None and str(run_dir) != str(self.run_dir):
            self.run_dir = run_dir
        self._configure()
        if fixed_indices is None and self.bulk_index > 0:
            fixed_indices = np.arange(min(self.bulk_index, len(slab)))
        return super().run_lammps_opt(slab, run_dir=run_dir, fixed_indices=fixed_indices, optimizer=optimizer, **kwargs)

    def run_lammps_energy(self, slab, run_dir=None, **kwargs):
        if run_dir is not None and str(run_dir) != str(self.run_dir):
            self.run_dir = run_dir
        return super().run_lammps_energy(slab, run_dir=run_dir, **kwargs)

    def relax_batch(self, atoms_list, fixed_indices=None, **kwargs):
        self._configure()
        if fixed_indices is None and self.bulk_index > 0:
            fixed_indices = [np.arange(min(self.bulk_index, len(a))) for a in atoms_list]
        return super().relax_batch(atoms_list, fixed_indices=fixed_indices, **kwargs)

    def evaluate_packed(self, n_atoms, Z, pos, cell, pbc, relax: bool = False, fixed_mask=None, **kwargs) -> dict:
        """As the base class; relaxations without a mask hold the template's bulk group (the first ``bulk_index`` atoms of every slab)."""
        self._configure()
        if relax and fixed_mask is None and self.bulk_index > 0:
            n_atoms = np.asarray(n_atoms, dtype=np.int64)
            start = np.concatenate([[0], np.cumsum(n_atoms)])
            local = np.arange(int(start[-1])) - np.repeat(start[:-1], n_atoms)
            fixed_mask = (local < self.bulk_index).astype(np.uint8)
        return super().evaluate_packed(n_atoms, Z, pos, cell, pbc, relax=relax, fixed_mask=fixed_mask, **kwargs)


LAMMMPSCalc = LAMMPSSurfCalc      # (the reference's base class name, spelled as there)
