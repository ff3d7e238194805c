"""Experiment: the same 256 chains as one resident batch vs split over two / four engines (own HIP streams) on one GPU."""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench
from surface_sampling_amd import backend
from surface_sampling_amd.calculators import stoich_offset_table
blobs, S, offset_data = bench.load_golden()
table, const = stoich_offset_table(offset_data)
chains = bench.build_chains(S, 0, 256)
want = backend.WANT_ENERGY | backend.WANT_FORCES | backend.WANT_STD
for n_eng in (1, 2, 4):
    per = 256 // n_eng
    engs = []
    for k in range(n_eng):
        e = backend.PainnEngine(blobs, device=0, offset_per_z=table, offset_const=const)
        e.upload([(s.numbers, s.positions, s.cell, s.pbc) for s in chains[k * per:(k + 1) * per]])
        engs.append(e)
    for _ in range(3):
        for e in engs: e.run(want)
    for e in engs: e.synchronize()
    steps = 20
    t0 = time.perf_counter()
    for _ in range(steps):
        for e in engs: e.run(want)
    for e in engs: e.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n_eng} engine(s) x {per} chains: {1e3 * dt / steps:7.3f} ms per 256 evaluations -> {256 * steps / dt:9.1f} evaluations/s")
    for e in engs: e.close()
