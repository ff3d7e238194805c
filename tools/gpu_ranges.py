"""debug: magnitude ranges of the node-GEMM inputs (activations / adjoints) on the synthetic workload"""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from conftest import Golden
from surface_sampling_amd import backend, structures
g = Golden()
table, const = g.offset_table()
base = g.structure("SrTiO3_2x2_pristine").repeat((2, 2, 1))
chains = [structures.as_arrays(structures.synth_chain(base, c)) for c in range(32)]
os.environ["VSSR_L0_FACTORISE"] = "0"
eng = backend.PainnEngine(g.blobs, device=0, offset_per_z=table, offset_const=const)
eng.evaluate(chains)
for m in range(3):
    out = []
    for n in ["phi0", "phi1", "phi2", "s_msg0", "v_msg0", "s_msg1", "v_msg1", "s_msg2", "v_msg2", "s_upd0", "v_upd0", "s_upd1", "v_upd1",
              "s_upd2", "v_upd2", "sbar_msg0", "vbar_msg0"]:
        try:
            a = eng.debug_read(n, m)
            nz = np.abs(a[a != 0])
            out.append(f"{n}: max {np.abs(a).max():.3g} min|nz| {nz.min():.2g}")
        except Exception as e:
            out.append(f"{n}: n/a")
    print("model", m, " | ".join(out))
eng.close()
