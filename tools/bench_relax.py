"""Secondary measurement: cost of the lock-step relaxation driver against the number of chain-evaluations it needs.
256 bench chains (BASELINE configs[3] workload), lower slab layers held fixed; for several convergence thresholds:
wall time of vssr_batch_relax_{bfgs,fire}, sum over chains of (steps + 1) = evaluations an ideal driver performs, and the
time per chain-evaluation -- constant if converged chains cost nothing.  Prints one JSON line per run.
Usage: python tools/bench_relax.py [--chains 256] [--relax-steps 20]"""
import argparse, json, os, sys, time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (golden loaders)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=256)
    ap.add_argument("--relax-steps", type=int, default=20)
    args = ap.parse_args()
    from surface_sampling_amd import backend
    from surface_sampling_amd.calculators import stoich_offset_table

    blobs, S, offset_data = bench.load_golden()
    table, const = stoich_offset_table(offset_data)
    chains = bench.build_chains(S, 0, args.chains)
    packs = [(s.numbers, s.positions, s.cell, s.pbc) for s in chains]
    mask = np.concatenate([(s.positions[:, 2] < s.positions[:240, 2].max() - 4.0).astype(np.uint8) for s in chains])
    eng = backend.PainnEngine(blobs, device=0, offset_per_z=table, offset_const=const)
    eng.upload(packs)
    eng.run(); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.run()
    eng.synchronize()
    t_eval = (time.perf_counter() - t0) / 5
    for opt in ("BFGS", "FIRE"):
        for fmax in (0.01, 0.3, 1.0, 3.0):
            eng.upload(packs)
            eng.synchronize()
            t0 = time.perf_counter()
            info = eng.relax(opt, fixed=mask, max_steps=args.relax_steps, fmax=fmax)
            dt = time.perf_counter() - t0
            need = int((info["n_steps"] + 1).sum())
            lockstep, dispatched = eng.relax_counts()
            print(json.dumps({"optimizer": opt, "fmax": fmax, "chains": args.chains, "relax_steps": args.relax_steps,
                              "converged": int(info["converged"].sum()), "mean_steps": float(info["n_steps"].mean()),
                              "chain_evaluations_needed": need, "lockstep_evaluations": lockstep,
                              "chain_evaluations_dispatched": dispatched, "dispatched_over_needed": round(dispatched / max(1, need), 4),
                              "wall_s": round(dt, 4),
                              "ms_per_256_chain_evaluations": round(1e3 * dt / need * 256, 3),
                              "full_batch_evaluation_ms": round(1e3 * t_eval, 3),
                              "wall_if_no_chain_dropped_s": round((args.relax_steps + 1) * t_eval, 4)}))
    eng.close()


if __name__ == "__main__":
    main()
