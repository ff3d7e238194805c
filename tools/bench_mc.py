"""Secondary measurement (not the BASELINE metric): batched semigrand MC steps per second, B chains on one GPU, every
step = propose + change + lock-step device relaxation (ChainEnsemble default: BFGS, <= relax_steps evaluations) + Metropolis.
The wall time of the MC steps is split into the phases of ChainEnsemble.step_semigrand (wrappers around the methods, so the
split is of the code that ships, not of a copy):

    device_relax      the vssr_batch_relax_* call (lock-step evaluations + optimizer steps, host blocked on the device)
    upload / download vssr_batch_upload (incl. packing the B slabs into the ABI arrays) / vssr_batch_download
    host_structures   ChainEnsemble.structure(): the B unrelaxed slabs as arrays
    host_results      relax_batch's per-chain result tuples (relaxed slab copies, result dicts, out-of-bounds guard)
    host_energy       surface-energy arithmetic per chain
    host_mc           proposal, change_site, Metropolis, state bookkeeping (everything else)

Prints one JSON line.  Usage: python tools/bench_mc.py [--chains 256] [--steps 10] [--relax-steps 20]
The reference's own figure for the same loop: 606 s for 50 proposals of ONE 72-atom chain = 12.1 s per proposal
(/root/reference/tutorials/SrTiO3_001.ipynb:1558, RTX 2080 Ti + nff)."""
import argparse, json, os, sys, time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (golden loaders)


class Phases:
    def __init__(self):
        self.t = {}
        self.stack = []

    def wrap(self, obj, name, phase):
        fn = getattr(obj, name)

        def timed(*a, **k):
            t0 = time.perf_counter()
            self.stack.append(0.0)
            try:
                return fn(*a, **k)
            finally:
                dt = time.perf_counter() - t0
                inner = self.stack.pop()
                self.t[phase] = self.t.get(phase, 0.0) + dt - inner     # exclusive time
                if self.stack:
                    self.stack[-1] += dt

        setattr(obj, name, timed)

    def reset(self):
        self.t = {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--relax-steps", type=int, default=20)
    ap.add_argument("--optimizer", default="BFGS")
    ap.add_argument("--no-relax", action="store_true", help="single-point acceptance energies (the reference's relax_atoms: false)")
    ap.add_argument("--groups", type=int, default=1, help="> 1: mc.ConcurrentChains -- the chains in this many groups, each with its own "
                    "calculator / engine / host thread (no phase split: the phases of different groups overlap)")
    ap.add_argument("--verify", action="store_true", help="with --groups: walk the same steps with ONE ensemble afterwards and "
                    "require identical accept counts, occupations and energies")
    args = ap.parse_args()
    from surface_sampling_amd import mc, structures
    from surface_sampling_amd.calculators import EnsembleNFFSurface

    blobs, S, offset_data = bench.load_golden()
    k = "SrTiO3_2x2_pristine"
    base = structures.Structure(S[f"{k}.numbers"], S[f"{k}.positions"], S[f"{k}.cell"], S[f"{k}.pbc"]).repeat((2, 2, 1))
    n = 8
    ztop = base.positions[:, 2].max()
    coords = []
    for i in range(n):
        for j in range(n):
            p = (i + 0.5) / n * base.cell[0] + (j + 0.5) / n * base.cell[1]
            coords.append([p[0], p[1], ztop + 1.5])
    fixed = np.flatnonzero(base.positions[:, 2] < ztop - 4.0)
    def new_calc():
        c = EnsembleNFFSurface(blobs, device="cuda:0", model_units="kcal/mol", prediction_units="eV", offset_units="atomic")
        c.set(offset=True, offset_data=offset_data, chem_pots={"Sr": -2, "Ti": 0, "O": 0})
        return c
    if args.groups > 1:
        return grouped(args, base, np.array(coords), fixed, new_calc)
    calc = new_calc()
    ens = mc.ChainEnsemble(base, np.array(coords), ("Sr", "O"), args.chains, calc, seed=1, relax=not args.no_relax,
                           relax_steps=args.relax_steps, fmax=0.01, fixed_indices=fixed, temperature=0.1,
                           optimizer=args.optimizer)
    # pre-populate so that chains look like mid-run states (8..32 adsorbates)
    state = ens.state
    for s in range(1, 21):
        site, end, _, _ = ens.propose(10_000 + s, state)
        state = ens.apply(state, site, end)
    ens.state = state
    ph = Phases()
    eng = calc._get_engine()
    for name, phase in (("relax_bfgs", "device_relax"), ("relax_fire", "device_relax"), ("upload", "upload"),
                        ("download", "download"), ("set_positions", "upload"), ("run", "device_relax")):
        if hasattr(eng, name):
            ph.wrap(eng, name, phase)
    ph.wrap(eng, "evaluate", "device_relax")     # (relax=False: one lock-step evaluation; upload / download inside are exclusive)
    ph.wrap(calc, "calculate_batch", "host_results")
    ph.wrap(ens, "structure", "host_structures")
    ph.wrap(calc, "relax_batch", "host_results")
    ph.wrap(ens, "surface_energy_fn", "host_energy")
    for name in ("relax_chains",):
        if hasattr(calc, name):
            ph.wrap(calc, name, "host_results")
    ens.initialize()
    ph.reset()
    n_eval0 = getattr(ens, "n_chain_evaluations", 0)
    t0 = time.perf_counter()
    acc = []
    for _ in range(args.steps):
        acc.append(ens.step_semigrand().mean())
    dt = time.perf_counter() - t0
    split = {k: v / args.steps for k, v in ph.t.items()}
    split["host_mc"] = dt / args.steps - sum(split.values())
    dev = split.get("device_relax", 0.0)
    host = sum(v for k, v in split.items() if k.startswith("host_"))
    line = {"metric": "batched semigrand MC steps/s (all chains advance one Change event incl. the lock-step relaxation)",
            "chains": args.chains, "atoms_per_chain": int(len(base) + ens.num_adsorbates().mean()),
            "optimizer": args.optimizer if not args.no_relax else None, "relax_steps": args.relax_steps if not args.no_relax else 0, "mc_steps": args.steps,
            "s_per_lockstep": dt / args.steps, "proposals_per_s": args.chains * args.steps / dt,
            "acceptance": float(np.mean(acc)),
            "split_s_per_lockstep": {k: round(v, 5) for k, v in sorted(split.items())},
            "device_share": dev / (dt / args.steps), "transfer_share": (split.get("upload", 0) + split.get("download", 0)) / (dt / args.steps),
            "host_share": host / (dt / args.steps),
            "reference": {"s_per_proposal": 606.0 / 50, "where": "tutorials/SrTiO3_001.ipynb:1558 (one 72-atom chain, RTX 2080 Ti, nff)"}}
    line["speedup_vs_reference_per_proposal"] = line["proposals_per_s"] * line["reference"]["s_per_proposal"]
    print(json.dumps(line))


def grouped(args, base, coords, fixed, new_calc):
    from surface_sampling_amd import mc
    calcs = [new_calc() for _ in range(args.groups)]
    for c in calcs:
        c.streams = 1                        # one engine per group: the groups are the streams
    cc = mc.ConcurrentChains.build(base, coords, ("Sr", "O"), args.chains, calcs, seed=1, relax=not args.no_relax,
                                   relax_steps=args.relax_steps, fmax=0.01, fixed_indices=fixed, temperature=0.1,
                                   optimizer=args.optimizer)
    for g in cc.groups:                      # the same mid-run starting states as the single-ensemble measurement
        state = g.state
        for s in range(1, 21):
            site, end, _, _ = g.propose(10_000 + s, state)
            state = g.apply(state, site, end)
        g.state = state
    cc.initialize()
    cc.steps(2)                              # warm-up (engine capacities settle)
    t0 = time.perf_counter()
    acc = cc.steps(args.steps)
    dt = time.perf_counter() - t0
    line = {"metric": "batched semigrand MC steps/s, chains in concurrent groups (mc.ConcurrentChains)", "chains": args.chains,
            "groups": len(cc.groups), "atoms_per_chain": int(len(base) + cc.num_adsorbates().mean()),
            "optimizer": args.optimizer if not args.no_relax else None, "relax_steps": args.relax_steps if not args.no_relax else 0,
            "mc_steps": args.steps, "s_per_lockstep": dt / args.steps, "proposals_per_s": args.chains * args.steps / dt,
            "acceptance": float(acc.mean() / args.steps), "energy_checksum": float(np.sum(cc.energy)),
            "reference": {"s_per_proposal": 606.0 / 50, "where": "tutorials/SrTiO3_001.ipynb:1558 (one 72-atom chain, RTX 2080 Ti, nff)"}}
    line["speedup_vs_reference_per_proposal"] = line["proposals_per_s"] * line["reference"]["s_per_proposal"]
    if args.verify:
        one = mc.ChainEnsemble(base, coords, ("Sr", "O"), args.chains, new_calc(), seed=1, relax=not args.no_relax,
                               relax_steps=args.relax_steps, fmax=0.01, fixed_indices=fixed, temperature=0.1, optimizer=args.optimizer)
        state = one.state
        for s in range(1, 21):
            site, end, _, _ = one.propose(10_000 + s, state)
            state = one.apply(state, site, end)
        one.state = state
        one.initialize()
        ref = np.zeros(args.chains, np.int64)
        for k in range(2 + args.steps):
            a = one.step_semigrand()
            if k >= 2:
                ref += a
        same = bool(np.array_equal(ref, acc) and np.array_equal(one.state.species, cc.species)
                    and np.array_equal(one.state.energy, cc.energy))
        line["identical_to_one_ensemble"] = same
        if not same:
            print(json.dumps(line))
            raise SystemExit("grouped chains diverged from the single ensemble")
    print(json.dumps(line))


if __name__ == "__main__":
    main()
