"""Stand-in for ase.calculators.calculator (see ../__init__.py)."""
import numpy as np

all_properties = ["energy", "forces", "stress", "stresses", "dipole", "charges", "magmom", "magmoms", "free_energy", "energies"]
all_changes = ["positions", "numbers", "cell", "pbc", "initial_charges", "initial_magmoms"]


class CalculatorError(RuntimeError):
    pass


class PropertyNotImplementedError(NotImplementedError):
    pass


class CalculatorSetupError(CalculatorError):
    pass


def equal(a, b, tol=None):
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        a, b = np.asarray(a), np.asarray(b)
        if a.shape != b.shape:
            return False
        return bool((a == b).all()) if tol is None else bool(np.allclose(a, b, rtol=tol, atol=tol))
    if isinstance(a, dict) and isinstance(b, dict):
        return a.keys() == b.keys() and all(equal(a[k], b[k], tol) for k in a)
    return a == b


def compare_atoms(atoms1, atoms2, tol=1e-15, excluded_properties=None):
    """Names of the properties that differ (all_changes when there is no previous atoms object)."""
    if atoms1 is None:
        return list(all_changes)
    changes = []
    for name in ("positions", "numbers", "cell", "pbc"):
        a, b = np.asarray(getattr(atoms1, name)), np.asarray(getattr(atoms2, name))
        if a.shape != b.shape or not (np.allclose(a, b, rtol=0, atol=tol) if a.dtype.kind == "f" else (a == b).all()):
            changes.append(name)
    return [c for c in changes if c not in (excluded_properties or ())]


class BaseCalculator:
    implemented_properties = []
    default_parameters = {}
    ignored_changes = set()
    discard_results_on_any_change = False

    def get_property(self, name, atoms=None, allow_calculation=True):
        if name not in self.implemented_properties:
            raise PropertyNotImplementedError(f"{name} property not implemented")
        if atoms is None:
            atoms = self.atoms
            system_changes = []
        else:
            system_changes = self.check_state(atoms)
            if system_changes:
                self.reset()
        if name not in self.results:
            if not allow_calculation:
                return None
            self.calculate(atoms, [name], system_changes)
        if name not in self.results:
            raise PropertyNotImplementedError(f"{name} not present in this calculation")
        result = self.results[name]
        if isinstance(result, np.ndarray):
            result = result.copy()
        return result

    def calculation_required(self, atoms, properties):
        if self.check_state(atoms):
            return True
        return any(name not in self.results for name in properties)

    def get_potential_energy(self, atoms=None, force_consistent=False):
        return self.get_property("energy", atoms)

    def get_forces(self, atoms=None):
        return self.get_property("forces", atoms)


class Calculator(BaseCalculator):
    def __init__(self, restart=None, ignore_bad_restart_file=None, label=None, atoms=None, directory=".", **kwargs):
        self.atoms = None
        self.results = {}
        self.parameters = None
        self.directory = directory
        self.prefix = None
        self.label = label
        if self.parameters is None:
            self.parameters = self.get_default_parameters()
        if atoms is not None:
            atoms.calc = self
        self.set(**kwargs)
        if not hasattr(self, "name"):
            self.name = self.__class__.__name__.lower()

    def get_default_parameters(self):
        return dict(self.default_parameters)

    def reset(self):
        self.atoms = None
        self.results = {}

    def set(self, **kwargs):
        changed_parameters = {}
        for key, value in kwargs.items():
            oldvalue = self.parameters.get(key)
            if key not in self.parameters or not equal(value, oldvalue):
                changed_parameters[key] = value
                self.parameters[key] = value
        if self.discard_results_on_any_change and changed_parameters:
            self.reset()
        return changed_parameters

    def check_state(self, atoms, tol=1e-15):
        return compare_atoms(self.atoms, atoms, tol=tol, excluded_properties=set(self.ignored_changes))

    def calculate(self, atoms=None, properties=("energy",), system_changes=all_changes):
        if atoms is not None:
            self.atoms = atoms.copy()
