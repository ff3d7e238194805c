"""debug: bitwise repeatability of the benchmark batch itself (256 chains of 248-272 atoms), REPS evaluations on each of 3 fresh engines"""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench
from surface_sampling_amd import backend
from surface_sampling_amd.calculators import stoich_offset_table
blobs, S, offset_data = bench.load_golden()
table, const = stoich_offset_table(offset_data)
chains = bench.build_chains(S, 0, int(os.environ.get("NCHAIN", "256")))
packs = [(s.numbers, s.positions, s.cell, s.pbc) for s in chains]
reps = int(os.environ.get("REPS", "100"))
ref, bad = None, 0
for eng_i in range(3):
    eng = backend.PainnEngine(blobs, device=0, offset_per_z=table, offset_const=const)
    eng.upload(packs)
    for i in range(reps):
        eng.run(backend.WANT_ALL)
        r = eng.download(backend.WANT_ALL)
        if ref is None:
            ref = (r["energy"].copy(), r["forces"].copy(), r["forces_std"].copy())
            continue
        if not (np.array_equal(r["energy"], ref[0]) and np.array_equal(r["forces"], ref[1]) and np.array_equal(r["forces_std"], ref[2])):
            bad += 1
            if bad < 5:
                print("mismatch engine", eng_i, "rep", i, float(np.abs(r["forces"] - ref[1]).max()))
    eng.close()
print("evaluations", 3 * reps, "chains", len(chains), "mismatching evaluations", bad)
