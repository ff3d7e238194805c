"""Precision envelope of the fp16x2-split arithmetic: GPU vs the fp64 oracle with the embedding (hence every activation
scale) or the radial-filter weights multiplied by a factor.  Prints one JSON line per case."""
import json, os, sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
import oracle  # noqa: E402
from surface_sampling_amd import backend, checkpoint, structures  # noqa: E402

blobs, S, _ = bench.load_golden()
oracle.set_threads(min(os.cpu_count() or 1, 16))
k = "SrTiO3_2x2_pristine"
s = structures.synth_chain(structures.Structure(S[f"{k}.numbers"], S[f"{k}.positions"], S[f"{k}.cell"], S[f"{k}.pbc"]), 3, grid=(4, 4))


def scaled(blob, what, a):
    b = blob.copy()
    f = checkpoint.blob_to_fields(b)
    if what == "embed":
        f["embed"] *= a
    else:   # filter weights of every layer (messages scale by a per layer)
        for l in range(3):
            f[f"msg{l}.Wd"] *= a
            f[f"msg{l}.bd"] *= a
    return b


for what in ("embed", "filter"):
    for a in (float(x) for x in os.environ.get("ENV_FACTORS", "1e-4 1e-3 1e-2 1e-1 1 1e1 1e2 1e3").split()):
        bl = [scaled(b, what, a) for b in blobs]
        eng = backend.PainnEngine(bl, device=0, model_units_per_ev=1.0)
        r = eng.evaluate([(s.numbers, s.positions, s.cell, s.pbc)])
        o = oracle.ensemble(bl, s.numbers, s.positions, s.cell, s.pbc, 64, None, 0.0, 1.0)
        em = np.abs(o["energy_models"]).max()
        fm = np.abs(o["forces"]).max()
        print(json.dumps({"scaled": what, "factor": a, "E_oracle": o["energy"], "dE": float(r["energy"][0]) - o["energy"],
                          "rel_dE": abs(float(r["energy"][0]) - o["energy"]) / max(em, 1e-30),
                          "max_dF": float(np.abs(r["forces"] - o["forces"]).max()), "rel_dF": float(np.abs(r["forces"] - o["forces"]).max() / max(fm, 1e-30)),
                          "finite": bool(np.isfinite(r["energy"]).all() and np.isfinite(r["forces"]).all())}))
        eng.close()
