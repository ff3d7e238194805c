"""Latency of small batches of the reference's own slab size (~70 atoms): one evaluation per call on a resident batch, per kernel class."""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
from conftest import Golden
from surface_sampling_amd import backend, structures
g = Golden(); table, const = g.offset_table()
base = g.structure("SrTiO3_2x2_pristine")
for nb in (1, 8, 32):
    packs = [structures.as_arrays(structures.synth_chain(base, c, grid=(4, 4))) for c in range(nb)]
    eng = backend.PainnEngine(g.blobs, device=0, offset_per_z=table, offset_const=const)
    eng.upload(packs)
    for _ in range(10): eng.run(); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): eng.run(); eng.synchronize()
    dt = (time.perf_counter() - t0) / 100
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(20): eng.run()
    eng.synchronize()
    pr = {k: round(1e3 * v["total_ms"] / 20, 1) for k, v in eng.profile_read().items() if v["launches"]}
    print(f"{nb} chains of ~70 atoms: {1e6*dt:.0f} us per evaluation", pr)
    eng.close()
