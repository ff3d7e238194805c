"""Stability soak of the C-ABI engine (not a parity test): (1) create / upload / run / destroy many times and watch the free device
memory (leaks), (2) a long run of lock-steps with positions changing every step (capacity regrows included: chains are compressed
towards the end so that degrees rise), checking finiteness, the saturation flag, and that the first batch gives bit-identical
results when it is evaluated again at the end, (3) repeated lock-step relaxations on one handle."""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
from conftest import Golden
from surface_sampling_amd import backend, structures
g = Golden()
table, const = g.offset_table()
s60 = g.structure("SrTiO3_2x2_pristine")
n = int(os.environ.get("NCHAIN", "64"))
steps = int(os.environ.get("STEPS", "600"))
chains = [structures.synth_chain(s60.repeat((2, 2, 1)), c) for c in range(n)]
packs = [structures.as_arrays(c) for c in chains]
free0 = None
for i in range(int(os.environ.get("CYCLES", "12"))):
    eng = backend.PainnEngine(g.blobs, device=0, offset_per_z=table, offset_const=const)
    eng.upload(packs[: 8 + (i % 5) * 8])
    eng.run(); r = eng.download()
    assert np.isfinite(r["energy"]).all()
    eng.close()
    torch.cuda.synchronize()
    free = torch.cuda.mem_get_info()[0]
    if i == 2: free0 = free        # after the allocator and the code objects have settled
    if i >= 2: print(f"cycle {i:2d} free {free / 2**20:10.1f} MiB  (delta {(free - free0) / 2**20:+.1f})")
leak = free0 - free
print("leak after create/destroy cycles: %.1f MiB" % (leak / 2**20))
assert leak < 64 * 2**20, leak

eng = backend.PainnEngine(g.blobs, device=0, offset_per_z=table, offset_const=const)
eng.upload(packs)
eng.run(); ref = eng.download()
ref = {k: np.array(ref[k], copy=True) for k in ("energy", "forces", "energy_std", "forces_std")}
pos0 = np.concatenate([p[1] for p in packs]).astype(np.float64)
rng = np.random.default_rng(7)
t0 = time.time()
bad = 0
for it in range(steps):
    amp = 0.02 + 0.10 * (it / steps)                      # growing displacements: neighbor counts move, capacity regrows happen
    eng.set_positions(pos0 + rng.normal(0.0, amp, pos0.shape))
    eng.run()
    if it % 25 == 0 or it == steps - 1:
        r = eng.download()
        fin = np.isfinite(r["energy"]).all() and np.isfinite(r["forces"]).all()
        bad += 0 if fin else 1
        print(f"step {it:4d} amp {amp:.3f} E[0] {float(r['energy'][0]):.4f} max|F| {float(np.abs(r['forces']).max()):.2f} saturated {int(np.asarray(r['saturated']).sum())} finite {fin}")
eng.set_positions(pos0); eng.run(); r = eng.download()
same = all(np.array_equal(r[k], ref[k]) for k in ref)
print("lock-steps %d in %.1f s; first batch bit-identical at the end: %s; non-finite samples: %d" % (steps, time.time() - t0, same, bad))
assert same and bad == 0
fixed = np.concatenate([(np.arange(len(c)) < len(c) // 2).astype(np.uint8) for c in chains])   # lower half of every chain frozen
for rep in range(3):
    eng.upload(packs)
    out = eng.relax_bfgs(fixed=fixed, max_steps=15, fmax=0.05)
    res = eng.download()
    print("relax", rep, "steps", np.asarray(out["n_steps"])[:6], "E[0] %.5f" % float(res["energy"][0]))
    if rep == 0: e_first, p_first = np.array(res["energy"], copy=True), out["positions"].copy()
    assert np.array_equal(np.asarray(res["energy"]), e_first) and np.array_equal(out["positions"], p_first)
eng.close()
print("OK")
