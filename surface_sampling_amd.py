"""Import alias: ``import surface_sampling_amd`` loads the package in ``surface-sampling_amd/``.

The package directory carries the reference's name (with its hyphen), which Python cannot
import directly; this one-file loader registers it under a valid module name.
"""

import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "surface-sampling_amd")
_spec = importlib.util.spec_from_file_location(
    "surface_sampling_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["surface_sampling_amd"] = _mod
_spec.loader.exec_module(_mod)
