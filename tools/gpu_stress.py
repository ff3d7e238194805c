"""debug: bitwise repeatability stress of the evaluation paths (64 chains x N repeats, fresh engines)"""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from conftest import Golden
from surface_sampling_amd import backend, structures
g = Golden()
table, const = g.offset_table()
base = g.structure("SrTiO3_2x2_pristine")
chains = [structures.as_arrays(structures.synth_chain(base, c, grid=(4, 4))) for c in range(int(os.environ.get("NCHAIN", "64")))]
reps = int(os.environ.get("REPS", "30"))
for env in ({}, {"VSSR_L0_FACTORISE": "0"}):
    os.environ.update(env)
    bad_e = bad_f = 0
    ref = None
    for eng_i in range(3):
        eng = backend.PainnEngine(g.blobs, device=0, offset_per_z=table, offset_const=const)
        for i in range(reps):
            r = eng.evaluate(chains)
            if ref is None:
                ref = (r["energy"].copy(), r["forces"].copy())
                continue
            de = np.flatnonzero(r["energy"] != ref[0]); df = np.flatnonzero((r["forces"] != ref[1]).any(axis=1))
            if len(de) or len(df):
                if bad_e + bad_f < 5:
                    print("   mismatch eng", eng_i, "rep", i, "chains", de[:6].tolist(), "dE", (r["energy"][de[:3]] - ref[0][de[:3]]).tolist(),
                          "n_force_rows", len(df), "max|dF|", float(np.abs(r["forces"] - ref[1]).max()))
                bad_e += len(de) > 0; bad_f += len(df) > 0
        eng.close()
    for k in env: del os.environ[k]
    print(env, "evaluations", 3 * reps, "energy mismatches", bad_e, "force mismatches", bad_f)
