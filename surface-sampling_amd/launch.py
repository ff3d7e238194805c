"""Run one of the reference's scripts UNCHANGED on the MI355X backend.

    python -m surface_sampling_amd.launch scripts/sample_surface.py --run_name ... --model_paths ... [script arguments]

``scripts/sample_surface.py`` builds its calculator with ``from mcmc.calculators import EnsembleNFFSurface``
(reference ``scripts/sample_surface.py:19,164-176``).  This launcher imports the reference's ``mcmc.calculators`` package
(it must be importable: the reference's own environment), replaces the calculator classes this backend provides -- in the
package namespace and in ``mcmc.calculators.calculators``, which ``mcmc.system`` / ``mcmc.dynamics`` resolve at call time --
and then executes the script with ``runpy`` under ``__main__``.  Nothing else changes: ``load_model`` still hands over nff
modules (accepted through ``state_dict()``), ``get_atoms_batch`` still builds the ``AtomsBatch`` (an ``ase.Atoms``: the
backend reads numbers / positions / cell / pbc from it and builds its own neighbor list on the GPU), ``SurfaceSystem``,
``optimize_slab`` and ``MCMC.run`` are the reference's.

``--vssr-keep NAME`` (repeatable, before the script path) leaves a class untouched, ``--vssr-list`` prints the mapping.
"""

from __future__ import annotations

import importlib
import runpy
import sys

# reference class (mcmc.calculators) -> attribute of surface_sampling_amd.calculators
REPLACEMENTS = {
    "EnsembleNFFSurface": "EnsembleNFFSurface",        # mcmc/calculators/calculators.py:366-489
    "NFFPourbaix": "NFFPourbaix",                      # :137-357
    "LAMMPSRunSurfCalc": "LAMMPSRunSurfCalc",          # :755-811 (pair_style eam)
    "LAMMPSSurfCalc": "LAMMPSSurfCalc",                # :696-752 (run_dir with lammps_config.json + templates: tersoff / eam)
    "LAMMMPSCalc": "LAMMMPSCalc",                      # :492-693
    "get_results_single": "get_results_single",        # :34-47
    "get_embeddings_single": "get_embeddings_single",  # :67-93
    "get_embeddings": "get_embeddings",
    "get_std_devs_single": "get_std_devs_single",      # :117-135
    "get_std_devs": "get_std_devs",
}


def install(keep=(), package: str = "mcmc.calculators") -> dict:
    """Patch the reference's calculator namespace; returns {name: replacement} of what was replaced."""
    from . import calculators as ours

    pkg = importlib.import_module(package)
    targets = [pkg]
    try:
        targets.append(importlib.import_module(package + ".calculators"))
    except ImportError:
        pass
    done = {}
    for name, attr in REPLACEMENTS.items():
        if name in keep:
            continue
        repl = getattr(ours, attr)
        for mod in targets:
            if hasattr(mod, name):
                setattr(mod, name, repl)
                done[name] = repl
    return done


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    keep, show = [], False
    while argv and argv[0].startswith("--vssr-"):
        flag = argv.pop(0)
        if flag == "--vssr-keep" and argv:
            keep.append(argv.pop(0))
        elif flag == "--vssr-list":
            show = True
        else:
            raise SystemExit(f"unknown launcher option {flag}")
    if show:
        for k, v in REPLACEMENTS.items():
            print(f"mcmc.calculators.{k} -> surface_sampling_amd.calculators.{v}")
        if not argv:
            return 0
    if not argv:
        raise SystemExit(__doc__)
    done = install(keep)
    if not done:
        raise SystemExit("mcmc.calculators exposes none of the classes this backend replaces")
    script = argv[0]
    sys.argv = argv                      # the script parses its own arguments
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
