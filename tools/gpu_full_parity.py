"""Validation run: every chain of the benchmark workload, GPU (one lock-step evaluation per block) vs the fp64 CPU oracle.

    NCHAIN=256 python tools/gpu_full_parity.py                  BASELINE configs[3]: the 256 chains of one GPU
    NBLOCKS=8  python tools/gpu_full_parity.py                  BASELINE configs[4]: all 8 blocks = 2 048 chains, block r = the chains
                                                                rank r of the 8-GPU run owns (bench.shard_plan)

Prints one JSON line: per block and overall max / mean deviations of the energy (float32 result word AND the fp64 word of
vssr_batch_energy_f64), the forces and the ensemble spread.  The oracle is the checker (test infrastructure)."""
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
per_block = int(os.environ.get("NCHAIN", "256"))
n_blocks = int(os.environ.get("NBLOCKS", "1"))
plan = bench.shard_plan(n_blocks, per_block)
eng = backend.PainnEngine(blobs, device=0, offset_per_z=table, offset_const=const)
t0 = time.time()
blocks, tot = [], {"dE32": [], "dE64": [], "dF": [], "dEs": []}
e_lo, e_hi, atoms = np.inf, -np.inf, 0
for r, (first, count) in enumerate(plan):
    chains = bench.build_chains(S, first, count)
    res = eng.evaluate([(s.numbers, s.positions, s.cell, s.pbc) for s in chains])
    dE32, dE64, dF, dEs = [], [], [], []
    for b, s in enumerate(chains):
        ref = oracle.ensemble(blobs, s.numbers, s.positions, s.cell, s.pbc, 64, table, const)
        a0, a1 = res["cfg_start"][b], res["cfg_start"][b + 1]
        dE32.append(abs(float(res["energy"][b]) - ref["energy"]))
        dE64.append(abs(float(res["energy_f64"][b]) - ref["energy"]))
        dF.append(float(np.abs(res["forces"][a0:a1] - ref["forces"]).max()))
        dEs.append(abs(float(res["energy_std_f64"][b]) - ref["energy_std"]))
    blocks.append({"rank": r, "first_chain": first, "chains": count, "atoms": int(res["cfg_start"][-1]),
                   "max_abs_dE_f32word_eV": max(dE32), "max_abs_dE_f64word_eV": max(dE64), "max_abs_dF_eV_per_A": max(dF),
                   "max_abs_dEstd_eV": max(dEs), "saturated": int(res["saturated"].sum())})
    for k, v in (("dE32", dE32), ("dE64", dE64), ("dF", dF), ("dEs", dEs)):
        tot[k].extend(v)
    e_lo, e_hi, atoms = min(e_lo, float(res["energy"].min())), max(e_hi, float(res["energy"].max())), atoms + int(res["cfg_start"][-1])
    print(f"block {r}: {json.dumps(blocks[-1])}", file=sys.stderr, flush=True)
out = {"chains": len(tot["dF"]), "blocks": n_blocks, "atoms": atoms,
       "max_abs_dE_eV": max(tot["dE32"]), "mean_abs_dE_eV": float(np.mean(tot["dE32"])),
       "max_abs_dE_f64word_eV": max(tot["dE64"]), "mean_abs_dE_f64word_eV": float(np.mean(tot["dE64"])),
       "max_abs_dF_eV_per_A": max(tot["dF"]), "mean_max_dF": float(np.mean(tot["dF"])), "max_abs_dEstd_eV": max(tot["dEs"]),
       "oracle_seconds": round(time.time() - t0, 1), "energy_range_eV": [e_lo, e_hi], "per_block": blocks}
print(json.dumps(out))
eng.close()
