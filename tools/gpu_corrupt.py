"""debug: which elements of the layer message outputs differ between repeated evaluations"""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from conftest import Golden
from surface_sampling_amd import backend, structures
g = Golden()
table, const = g.offset_table()
base = g.structure("SrTiO3_2x2_pristine")
chains = [structures.as_arrays(structures.synth_chain(base, c, grid=(4, 4))) for c in range(64)]
os.environ["VSSR_L0_FACTORISE"] = "0"
eng = backend.PainnEngine(g.blobs, device=0, offset_per_z=table, offset_const=const)
names = ["phi0", "s_msg0", "v_msg0", "s_upd0", "v_upd0", "phi1", "s_msg1", "v_msg1", "s_upd1", "v_upd1", "phi2", "s_msg2", "v_msg2"]
ref = None
shown = 0
for rep in range(60):
    r = eng.evaluate(chains)
    cur = {f"{n}/m{m}": eng.debug_read(n, m).copy() for m in range(3) for n in names}
    cs = r["cfg_start"]
    if ref is None:
        ref = cur; continue
    for n in cur:
        a, b = ref[n], cur[n]
        per_atom = a.size // cs[-1]
        d = np.flatnonzero(a != b)
        if len(d) and shown < 12:
            shown += 1
            atoms = d // per_atom; feats = d % per_atom
            ua = np.unique(atoms)
            print(f"rep {rep} {n}: {len(d)} differing values in {len(ua)} atoms; atoms {ua[:8].tolist()} chain {np.searchsorted(cs, ua[:8], side='right') - 1}; "
                  f"features of first atom {np.unique(feats[atoms == ua[0]])[:40].tolist()} max|d| {np.abs(a - b).max():.3g} rel {np.abs(a - b).max() / (np.abs(a).max() + 1e-30):.2g}")
            break   # first differing buffer in pipeline order
eng.close()
