"""Secondary measurement (not the BASELINE metric): the reference's GaN(0001) configuration -- BASELINE configs[1], Tersoff, canonical
sampling with 12 Ga adatoms on the 3 x 3 slab (36 + 12 atoms), every proposal relaxed with the LAMMPS-style CG minimiser (<= 100
iterations) before the Metropolis test -- as batched MC over B chains on one GPU (`mc.ChainEnsemble` + `TersoffSurfCalc`, fp64).
Prints one JSON line per chain count.  The reference's own figure for the same loop: 32.667 s for 1 040 proposals of ONE chain =
31 ms per proposal (/root/reference/tutorials/GaN_0001.ipynb:6356, CPU, in-process LAMMPS).
Usage: python tools/bench_gan.py [--chains 256,1024,4096] [--steps 5] [--relax-steps 100]"""
import argparse, json, os, sys, time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", default="256,1024,4096")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--relax-steps", type=int, default=100)
    ap.add_argument("--sites", type=int, default=6, help="n x n adsorption-site grid above the slab")
    ap.add_argument("--groups", type=int, default=1, help="chains in G mc.ConcurrentChains groups (own calculators / engines / host threads): "
                                                          "the slowest chain of one group's relaxation no longer idles the GPU")
    args = ap.parse_args()
    from surface_sampling_amd import backend, mc, structures
    from surface_sampling_amd.calculators import TersoffSurfCalc

    g = os.path.join(ROOT, "tests", "golden")
    S = np.load(os.path.join(g, "structures.npz"))
    with open(os.path.join(g, "GaN_tersoff_params.json")) as fh:
        params = np.array(json.load(fh)["params_ijk"], dtype=np.float64)
    k = "GaN_3x3_pristine"
    base = structures.Structure(S[f"{k}.numbers"], S[f"{k}.positions"], S[f"{k}.cell"], S[f"{k}.pbc"])
    ztop = base.positions[:, 2].max()
    n = args.sites
    coords = np.array([(i + 0.5) / n * base.cell[0] + (j + 0.5) / n * base.cell[1] for i in range(n) for j in range(n)], float)
    coords[:, 2] = ztop + 1.8
    fixed = np.flatnonzero(base.positions[:, 2] < ztop - 3.0)
    def composed(ens, B):                                           # the tutorial's composition: 12 Ga adatoms per chain, evenly spread
        state = ens.state
        for site in ens.even_adsorption_sites(12):
            state = ens.apply(state, np.full(B, int(site), np.int64), np.zeros(B, np.int64))
        ens.state = state
        assert (ens.num_adsorbates() == 12).all()

    if args.groups > 1:
        for B in [int(x) for x in args.chains.split(",")]:
            calcs = [TersoffSurfCalc(params, ["Ga", "N"], device="cuda:0") for _ in range(args.groups)]
            for c in calcs:
                c.set(relax_steps=args.relax_steps)
            cc = mc.ConcurrentChains.build(base, coords, ("Ga",), B, calcs, seed=4, relax=True, relax_steps=args.relax_steps,
                                           fixed_indices=fixed, temperature=0.3, optimizer="LAMMPS")
            for grp in cc.groups:
                composed(grp, len(grp.chain_ids))
            cc.initialize()
            cc.steps(1, canonical=True)                             # warm-up
            n0 = cc.n_evaluations
            t0 = time.perf_counter()
            acc = cc.steps(args.steps, canonical=True)
            dt = time.perf_counter() - t0
            print(json.dumps({"metric": "batched canonical MC proposals/s, GaN(0001) 3x3 Tersoff, every proposal CG-relaxed (<= %d iterations)" % args.relax_steps,
                              "chains": B, "groups": len(cc.groups), "mc_steps": args.steps, "s_per_lockstep": dt / args.steps,
                              "proposals_per_s": B * args.steps / dt, "acceptance": float(acc.mean() / args.steps),
                              "mean_energy_eV": float(np.mean(cc.energy)), "relaxations": int(cc.n_evaluations - n0)}), flush=True)
            del cc, calcs
        return
    for B in [int(x) for x in args.chains.split(",")]:
        calc = TersoffSurfCalc(params, ["Ga", "N"], device="cuda:0")
        calc.set(relax_steps=args.relax_steps)
        ens = mc.ChainEnsemble(base, coords, ("Ga",), B, calc, seed=4, relax=True, relax_steps=args.relax_steps,
                               fixed_indices=fixed, temperature=0.3, optimizer="LAMMPS")
        composed(ens, B)
        ens.initialize()
        ens.step_canonical()                                        # warm-up (engine capacities settle)
        # lock-step waste of the CG relaxations: chain-evaluations the chains NEEDED (their own n_eval + the final static one)
        # against what the lock-step driver DISPATCHED (vssr_batch_relax_counts), and why the chains stopped
        work = {"needed": 0, "dispatched": 0, "lockstep": 0, "relaxations": 0, "stop": {}, "evals": [], "calc_s": 0.0}
        inner = calc.evaluate_packed

        def counted(*a, **k):
            tc = time.perf_counter()
            out = inner(*a, **k)
            work["calc_s"] += time.perf_counter() - tc
            if "evaluations" in out:
                ev = np.asarray(out["evaluations"], dtype=np.int64)
                work["needed"] += int(ev.sum()) + len(ev)
                work["dispatched"] += int(out["dispatched_chain_evaluations"])
                work["lockstep"] += int(out["lockstep_evaluations"])
                work["relaxations"] += len(ev)
                work["evals"].append(ev)
                for r, c in zip(*np.unique(out["stop"], return_counts=True)):
                    name = backend.CG_STOP_REASONS.get(int(r), str(int(r)))
                    work["stop"][name] = work["stop"].get(name, 0) + int(c)
            return out

        calc.evaluate_packed = counted
        n0 = ens.n_evaluations
        t0 = time.perf_counter()
        acc = [ens.step_canonical().mean() for _ in range(args.steps)]
        dt = time.perf_counter() - t0
        line = {"metric": "batched canonical MC proposals/s, GaN(0001) 3x3 Tersoff, every proposal CG-relaxed (<= %d iterations)" % args.relax_steps,
                "chains": B, "atoms_per_chain": int(len(base) + 12), "sites": n * n, "mc_steps": args.steps,
                "s_per_lockstep": dt / args.steps, "proposals_per_s": B * args.steps / dt, "acceptance": float(np.mean(acc)),
                "mean_energy_eV": float(np.mean(ens.state.energy)), "relaxations": int(ens.n_evaluations - n0),
                "reference": {"s_per_proposal": 32.667 / 1040, "where": "tutorials/GaN_0001.ipynb:6356 (one chain, CPU, in-process LAMMPS)"}}
        ev = np.concatenate(work["evals"]) if work["evals"] else np.zeros(1, np.int64)
        line["lockstep_waste"] = {"needed_chain_evaluations": work["needed"], "dispatched_chain_evaluations": work["dispatched"],
                                  "dispatched_over_needed": work["dispatched"] / max(1, work["needed"]),
                                  "lockstep_evaluations_per_proposal": work["lockstep"] / max(1, args.steps),
                                  "evaluations_per_chain": {"min": int(ev.min()), "median": float(np.median(ev)), "mean": float(ev.mean()),
                                                            "p90": float(np.percentile(ev, 90)), "max": int(ev.max())},
                                  "stop_reasons": work["stop"]}
        line["calculator_share"] = work["calc_s"] / dt   # time inside calc.evaluate_packed (upload, relaxation, download) / wall time
        line["speedup_vs_reference_per_proposal"] = line["proposals_per_s"] * line["reference"]["s_per_proposal"]
        print(json.dumps(line), flush=True)
        del ens, calc


if __name__ == "__main__":
    main()
