"""Secondary measurement (not the BASELINE metric): BASELINE configs[0], the reference's Cu(100) toy -- EAM (funcfl), semigrand MC with
adsorbate Cu, static acceptance energies (relax_atoms False), kT annealed from 1.0 by 0.99 per sweep, 20 sweeps of 2 proposals --
as batched MC over B chains (`mc.ChainEnsemble` / `mc.ConcurrentChains` + `LAMMPSRunSurfCalc`, fp64).  The reference's figure for
the same run of ONE chain: 2.184 s for 40 proposals = 55 ms per proposal (/root/reference/tutorials/example.ipynb:251: every energy
is an `lmp` subprocess).  Prints one JSON line per chain count.  Usage: python tools/bench_cu.py [--chains 1024,16384] [--groups 2]"""
import argparse, json, os, sys, time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", default="1024,16384")
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--sweeps", type=int, default=20)
    ap.add_argument("--sweep-size", type=int, default=2)
    args = ap.parse_args()
    from surface_sampling_amd import mc, structures
    from surface_sampling_amd.calculators import LAMMPSRunSurfCalc

    g = os.path.join(ROOT, "tests", "golden")
    d = np.load(os.path.join(g, "cu100.npz"))
    slab = structures.Structure(d["numbers"], d["positions"], d["cell"], d["pbc"])
    sites = d["ads_coords"][d["site_kind"] != 2]

    def new_calc():
        c = LAMMPSRunSurfCalc(files=[os.path.join(g, "Cu_u3.eam")], device="cuda:0")
        c.set(pair_style="eam", pair_coeff=["* * Cu_u3.eam"])
        return c
    for B in [int(x) for x in args.chains.split(",")]:
        cc = mc.ConcurrentChains.build(slab, sites, ("Cu",), B, [new_calc() for _ in range(args.groups)], seed=11, relax=False,
                                       temperature=1.0)
        cc.initialize()
        cc.steps(2)                                   # warm-up
        t0 = time.perf_counter()
        hist = cc.run(total_sweeps=args.sweeps, sweep_size=args.sweep_size, start_temp=1.0, perform_annealing=True, alpha=0.99)
        dt = time.perf_counter() - t0
        n_prop = B * args.sweeps * args.sweep_size
        E = np.array(hist["energy_hist"])
        line = {"metric": "batched semigrand MC proposals/s, Cu(100) toy, EAM funcfl, static acceptance energies", "chains": B,
                "groups": len(cc.groups), "slab_atoms": int(len(slab)), "sites": int(len(sites)), "sweeps": args.sweeps,
                "sweep_size": args.sweep_size, "wall_s": dt, "proposals_per_s": n_prop / dt, "min_energy_eV": float(E.min()),
                "acceptance": float(np.mean(hist["frac_accept_hist"])),
                "reference": {"s_per_proposal": 2.184 / 40, "min_energy_eV": -25.2893,
                              "where": "tutorials/example.ipynb:251 (one chain, lmp subprocess per energy); tests/test_Cu.py:19"}}
        line["speedup_vs_reference_per_proposal"] = line["proposals_per_s"] * line["reference"]["s_per_proposal"]
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
