"""Debug: phase clocks of the chain-resident CG minimiser (build with -DCM_PHASE_TIMING: tools/build_variant.sh cmphase -DCM_PHASE_TIMING).
Usage: VSSR_EVAL_LIB=build/variants/lib_cmphase.so python tools/gpu_cm_phase.py [n_chains]"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from surface_sampling_amd import backend
from conftest import Golden
from test_cg import _gan_mc_like_batch
golden = Golden()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
packs, mask = _gan_mc_like_batch(golden, n, 21)
eng = backend.TersoffEngine(golden.tersoff_params, device=0)
lib = backend.load_library()
names = ["wrap", "nbr count", "row scan", "nbr fill", "reverse slots", "site4", "site (long rows)", "gather", "energy", "cg step"]
for form in ("1",):
    os.environ["VSSR_CG_FUSED"] = form
    eng.relax_cg_f64(packs, fixed=mask, max_iter=100)
    buf = (C.c_ulonglong * 16)()
    lib.vssr_debug_cm_phases(buf, 1)
    out = eng.relax_cg_f64(packs, fixed=mask, max_iter=100)
    lib.vssr_debug_cm_phases(buf, 1)
    ev = int(out[5][0]) + 1
    tot = sum(buf[:10])
    print(f"chain-resident minimiser, {n} chains, chain 0: {ev} evaluations, {tot / 100 / ev:.1f} us per evaluation")
    for k, nm in enumerate(names):
        print(f"   {nm:18s} {buf[k] / 100 / ev:7.2f} us  {100 * buf[k] / tot:5.1f} %")
    for k, nm in enumerate(["site4: neighborhood -> LDS", "site4: pass 1 (zeta, b_ij, pair terms)", "site4: pass 2 (cross three-body gradients)"]):
        print(f"      {nm:44s} {buf[10 + k] / 100 / ev:7.2f} us")
eng.close()
