"""Same inputs through two builds of the library (VSSR_EVAL_LIB), one process each: dump energies / forces per chain-size class
(`dump NAME`), then `cmp A B` prints the largest differences.  Used when a kernel change should leave results (nearly) unchanged."""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
out = os.path.join(root, "gpurun_out")
if sys.argv[1] == "cmp":
    a, b = (np.load(os.path.join(out, f"cmp_{n}.npz")) for n in sys.argv[2:4])
    for k in a.files:
        d = np.abs(a[k] - b[k])
        print(f"{k:16s} max|diff| {d.max():.3e}  at {np.unravel_index(d.argmax(), d.shape)}  n>1e-5: {(d > 1e-5).sum()}  |ref|max {np.abs(a[k]).max():.3e}")
    sys.exit(0)
from conftest import Golden
from surface_sampling_amd import backend, structures
g = Golden()
table, const = g.offset_table()
s60, s80 = g.structure("SrTiO3_2x2_pristine"), g.structure("SrTiO3_2x2x4_pristine")
n = int(os.environ.get("NCHAIN", "8"))
cfgs = {
    "small": [structures.synth_chain(s60, c, grid=(4, 4)) for c in range(n)],
    "d260": [structures.synth_chain(s60.repeat((2, 2, 1)), c) for c in range(n)],
    "fs16m": [structures.synth_chain(s60.repeat((3, 2, 1)), c, grid=(12, 8)) for c in range(n)],
    "fs8": [structures.synth_chain(s80.repeat((3, 2, 1)), c, grid=(12, 8)) for c in range(n)],
    "big8": [structures.synth_chain(s80.repeat((3, 3, 1)), c, grid=(12, 12)) for c in range(n)],
}
res = {}
for name, chains in cfgs.items():
    eng = backend.PainnEngine(g.blobs, device=0, offset_per_z=table, offset_const=const)
    eng.upload([structures.as_arrays(c) for c in chains])
    eng.run()
    r = eng.download()
    res[name + "_E"] = r["energy"].copy(); res[name + "_F"] = r["forces"].copy()
    print(name, "atoms", [len(c) for c in chains][:3], "E0", float(r["energy"][0]))
    eng.close()
np.savez(os.path.join(out, f"cmp_{sys.argv[2]}.npz"), **res)
