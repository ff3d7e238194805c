"""debug: colliding atoms / huge features — GPU (fp16-split matrix path) vs fp32/fp64 oracle"""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "oracle"))
from conftest import Golden
import oracle
from surface_sampling_amd import backend, structures
oracle.build(); oracle.set_threads(16)
g = Golden()
table, const = g.offset_table()
base = g.structure("SrTiO3_2x2_pristine")
eng = backend.PainnEngine(g.blobs, device=0, offset_per_z=table, offset_const=const)
for dmin in (1.0, 0.6, 0.4, 0.25, 0.15):
    s = base.copy()
    i, j = 7, 8
    v = s.positions[j] - s.positions[i]
    s.positions[j] = s.positions[i] + v / np.linalg.norm(v) * dmin
    r = eng.evaluate([structures.as_arrays(s)])
    ref = oracle.ensemble(g.blobs, s.numbers, s.positions, s.cell, s.pbc, 64, table, const)
    e, f = float(r["energy"][0]), r["forces"]
    print(f"d={dmin:4.2f}  E_gpu {e:14.6g}  E_ref {ref['energy']:14.6g}  rel dE {abs(e - ref['energy']) / max(1, abs(ref['energy'])):.2e}  "
          f"max|F| gpu {np.abs(f).max():.4g} ref {np.abs(ref['forces']).max():.4g}  rel dF {np.abs(f - ref['forces']).max() / max(1, np.abs(ref['forces']).max()):.2e}  finite {np.isfinite(e) and np.isfinite(f).all()}")
eng.close()
