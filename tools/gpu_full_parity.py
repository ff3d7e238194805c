"""One-off validation: every chain of the 256-chain benchmark batch, GPU (one lock-step evaluation) vs the fp64 CPU oracle."""
import json, os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import bench, oracle
from surface_sampling_amd import backend
from surface_sampling_amd.calculators import stoich_offset_table
oracle.build(); oracle.set_threads(min(os.cpu_count() or 1, 64))
blobs, S, offset_data = bench.load_golden()
table, const = stoich_offset_table(offset_data)
n = int(os.environ.get("NCHAIN", "256"))
chains = bench.build_chains(S, 0, n)
eng = backend.PainnEngine(blobs, device=0, offset_per_z=table, offset_const=const)
res = eng.evaluate([(s.numbers, s.positions, s.cell, s.pbc) for s in chains])
t0 = time.time()
dE, dF, dEs = [], [], []
for b, s in enumerate(chains):
    ref = oracle.ensemble(blobs, s.numbers, s.positions, s.cell, s.pbc, 64, table, const)
    a0, a1 = res["cfg_start"][b], res["cfg_start"][b + 1]
    dE.append(abs(float(res["energy"][b]) - ref["energy"]))
    dF.append(float(np.abs(res["forces"][a0:a1] - ref["forces"]).max()))
    dEs.append(abs(float(res["energy_std"][b]) - ref["energy_std"]))
out = {"chains": n, "atoms": int(res["cfg_start"][-1]), "max_abs_dE_eV": max(dE), "mean_abs_dE_eV": float(np.mean(dE)),
       "max_abs_dF_eV_per_A": max(dF), "mean_max_dF": float(np.mean(dF)), "max_abs_dEstd_eV": max(dEs),
       "oracle_seconds": round(time.time() - t0, 1), "energy_range_eV": [float(res["energy"].min()), float(res["energy"].max())]}
print(json.dumps(out))
eng.close()
