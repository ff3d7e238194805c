"""Energy of chains of 260 .. 1 2xx atoms against the fp64 oracle, as returned in the float32 result word (vssr_out.energy) and in the
fp64 word (vssr_batch_energy_f64): what part of the deviation is arithmetic and what was the output word."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from conftest import Golden
from surface_sampling_amd import backend, structures
oracle.build(); oracle.set_threads(min(os.cpu_count() or 1, 64))
g = Golden()
table, const = g.offset_table()
s60, s80 = g.structure("SrTiO3_2x2_pristine"), g.structure("SrTiO3_2x2x4_pristine")
chains = [structures.synth_chain(s60.repeat((2, 2, 1)), 4), structures.synth_chain(s60.repeat((3, 2, 1)), 2, grid=(12, 8)),
          structures.synth_chain(s80.repeat((3, 2, 1)), 7, grid=(12, 8)), structures.synth_chain(s80.repeat((3, 3, 1)), 5, grid=(12, 12)),
          structures.synth_chain(s80.repeat((4, 3, 1)), 6, grid=(16, 12)), structures.synth_chain(s80.repeat((5, 3, 1)), 3, grid=(20, 12))]
eng = backend.PainnEngine(g.blobs, device=0, offset_per_z=table, offset_const=const)
res = eng.evaluate([(c.numbers, c.positions, c.cell, c.pbc) for c in chains])
rows = []
for b, c in enumerate(chains):
    ref = oracle.ensemble(g.blobs, c.numbers, c.positions, c.cell, c.pbc, 64, table, const)
    a0, a1 = res["cfg_start"][b], res["cfg_start"][b + 1]
    e = ref["energy"]
    rows.append({"atoms": len(c), "energy_eV": e, "f32_spacing_eV": float(abs(np.spacing(np.float32(e)))),
                 "dE_f32_word": abs(float(res["energy"][b]) - e), "dE_f64_word": abs(float(res["energy_f64"][b]) - e),
                 "max_dF": float(np.abs(res["forces"][a0:a1] - ref["forces"]).max())})
    print(json.dumps(rows[-1]), flush=True)
eng.close()
