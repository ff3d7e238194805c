"""GPU diagnostic: bitwise batch-vs-single and run-to-run comparisons of intermediates."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from surface_sampling_amd import backend, structures

g = os.path.join(ROOT, "tests", "golden")
blobs = [np.fromfile(os.path.join(g, "weights", f"SrTiO3_painn_model0{m}.f32"), dtype="<f4") for m in (1, 2, 3)]
S = np.load(os.path.join(g, "structures.npz"))
k = "SrTiO3_2x2_pristine"
base = structures.Structure(S[f"{k}.numbers"], S[f"{k}.positions"], S[f"{k}.cell"], S[f"{k}.pbc"])
big = base.repeat((2, 2, 1))
chains = [structures.synth_chain(big, c) for c in (0, 7, 24)]
arr = lambda s: (s.numbers, s.positions, s.cell, s.pbc)
eng = backend.PainnEngine(blobs, device=0)
names = [f"{n}{l}" for l in range(3) for n in ("phi", "s_msg", "v_msg", "s_upd", "v_upd")] + ["e_atom", "sbar_msg0", "vbar_msg0"]

def snapshot(structs):
    res = eng.evaluate([arr(s) for s in structs])
    inter = {n: eng.debug_read(n, 1) for n in names}
    return res, inter

rb, ib = snapshot(chains)
rb2, ib2 = snapshot(chains)
print("run-to-run (batched): forces equal", np.array_equal(rb["forces"], rb2["forces"]),
      {n: int((ib[n] != ib2[n]).sum()) for n in names if (ib[n] != ib2[n]).any()})
cs = rb["cfg_start"]
for b, s in enumerate(chains):
    rs, isg = snapshot([s])
    a0, a1 = cs[b], cs[b + 1]
    df = np.abs(rs["forces"] - rb["forces"][a0:a1])
    print(f"chain {b}: atoms {a0}-{a1} (a0%32={a0%32}) forces differ at {int((df>0).sum())} of {df.size}, max {df.max():.3e}")
    for n in names:
        per = ib[n].size // cs[-1]
        x = ib[n].reshape(cs[-1], per)[a0:a1]
        y = isg[n].reshape(len(s), per)
        nd = int((x != y).sum())
        if nd:
            print(f"    {n}: {nd} of {x.size} differ, max abs {np.abs(x-y).max():.3e}, rows {np.unique(np.where(x!=y)[0])[:10]}")
