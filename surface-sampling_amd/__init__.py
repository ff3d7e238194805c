"""MI355X-native energy/force evaluation backend for the VSSR-MC inner loop.

Scope: neighbor list + PaiNN-ensemble / Tersoff energy and forces as HIP
kernels behind a C ABI (``include/vssr_eval.h``), exposed through ASE-Calculator-shaped
Python classes mirroring ``mcmc/calculators/calculators.py`` of the reference.
"""

__version__ = "0.1.0"

from . import checkpoint, structures  # noqa: F401
