"""Bitwise repeatability stress of every neighbor-sum path introduced in round 3 (the MFMA operand hazards of round 1 showed
up as isolated run-to-run differences, never in small parity tests): for each configuration NCHAIN chains x REPS repeats on
three fresh engines, energies and forces compared bit for bit with the first evaluation.
  small    74-atom chains      -> forward 4 waves x 4 workgroups per CU, reverse 4 waves
  medium   140-atom chains     -> forward 8 waves x 2 workgroups per CU
  fs16m    370-atom chains     -> forward 16-feature slices with the residual from memory, reverse 8 waves
  fs8      490-atom chains     -> forward 8-feature slices, reverse 16-feature slices / 8 waves
  fs8all   VSSR_EDGE_FS16_MAX=0 on 260-atom chains -> 8-feature slices in both directions (4-wave reverse)
  big8     735-atom chains     -> 8-feature slices both directions, 8-wave reverse
  fs4all   VSSR_EDGE_FS16_MAX=0 VSSR_EDGE_FS8_MAX=0 on 260-atom chains -> the class of 788 .. 1 462-atom chains: reverse 4-feature slices,
           forward two passes of the 8-feature kernel (round 5)
  fs4k     the same with VSSR_EDGE_FWD_2PASS=0 -> 4-feature slices in both directions (round 4)
  mpass    260-atom chains, VSSR_EDGE_BWD_MPASS=2 VSSR_EDGE_SUB_CHUNK=100 -> multi-pass 16-feature kernels in both directions (3 ranges)
  big4     975-atom chains     -> multi-pass 16-feature kernels in both directions by the chains' own size (forward 3, reverse 2 ranges)"""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from conftest import Golden
from surface_sampling_amd import backend, structures
g = Golden()
table, const = g.offset_table()
s60, s80 = g.structure("SrTiO3_2x2_pristine"), g.structure("SrTiO3_2x2x4_pristine")
n = int(os.environ.get("NCHAIN", "48"))
reps = int(os.environ.get("REPS", "20"))
cfgs = {
    "small": ({}, [structures.synth_chain(s60, c, grid=(4, 4)) for c in range(4 * n)]),
    "medium": ({}, [structures.synth_chain(s60.repeat((2, 1, 1)), c, grid=(8, 4)) for c in range(2 * n)]),
    "fs16m": ({}, [structures.synth_chain(s60.repeat((3, 2, 1)), c, grid=(12, 8)) for c in range(n)]),
    "fs8": ({}, [structures.synth_chain(s80.repeat((3, 2, 1)), c, grid=(12, 8)) for c in range(n)]),
    "fs8all": ({"VSSR_EDGE_FS16_MAX": "0"}, [structures.synth_chain(s60.repeat((2, 2, 1)), c) for c in range(n)]),
    "big8": ({}, [structures.synth_chain(s80.repeat((3, 3, 1)), c, grid=(12, 12)) for c in range(max(n // 2, 8))]),
    "fs4all": ({"VSSR_EDGE_FS16_MAX": "0", "VSSR_EDGE_FS8_MAX": "0"}, [structures.synth_chain(s60.repeat((2, 2, 1)), c) for c in range(n)]),
    "fs4k": ({"VSSR_EDGE_FS16_MAX": "0", "VSSR_EDGE_FS8_MAX": "0", "VSSR_EDGE_FWD_2PASS": "0"}, [structures.synth_chain(s60.repeat((2, 2, 1)), c) for c in range(n)]),
    "mpass": ({"VSSR_EDGE_FS16_MAX": "0", "VSSR_EDGE_FS8_MAX": "0", "VSSR_EDGE_BWD_MPASS": "2", "VSSR_EDGE_SUB_CHUNK": "100"},
              [structures.synth_chain(s60.repeat((2, 2, 1)), c) for c in range(n)]),
    "big4": ({}, [structures.synth_chain(s80.repeat((4, 3, 1)), c, grid=(16, 12)) for c in range(max(n // 3, 8))]),
}
only = os.environ.get("ONLY")
total_bad = 0
for name, (env, chains) in cfgs.items():
    if only and name not in only.split(","):
        continue
    os.environ.update(env)
    packs = [structures.as_arrays(c) for c in chains]
    bad = 0
    ref = None
    for eng_i in range(3):
        eng = backend.PainnEngine(g.blobs, device=0, offset_per_z=table, offset_const=const)
        eng.upload(packs)
        for i in range(reps):
            eng.run()
            r = eng.download()
            if ref is None:
                ref = (r["energy"].copy(), r["forces"].copy(), r["forces_std"].copy())
                assert np.isfinite(ref[0]).all() and not r["saturated"].any()
                continue
            if not (np.array_equal(r["energy"], ref[0]) and np.array_equal(r["forces"], ref[1]) and np.array_equal(r["forces_std"], ref[2])):
                bad += 1
                if bad < 4:
                    print("   mismatch", name, "engine", eng_i, "rep", i, "max|dF|", float(np.abs(r["forces"] - ref[1]).max()))
        eng.close()
    for k in env:
        del os.environ[k]
    total_bad += bad
    print(f"{name:8s} chains {len(chains):4d} atoms {sum(len(c) for c in chains):7d} evaluations {3 * reps:4d} mismatches {bad}")
sys.exit(1 if total_bad else 0)
