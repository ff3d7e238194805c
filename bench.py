#!/usr/bin/env python3
"""bench.py — MC energy-evaluations/sec of the MI355X backend on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one lock-step energy+force evaluation of every resident chain: neighbor list +
3-model PaiNN ensemble forward + reverse pass (what one BFGS step costs in the reference,
mcmc/calculators/calculators.py:484).  Workload at N=1: BASELINE configs[3] — 256 independent chains of
the SrTiO3(001) 2x2 slab tiled 2x2 in-plane (240 atoms) plus 8..32 seeded adsorbates,
inputs resident in HBM before the timed region.  N > 1: every rank owns 256 more chains (weak scaling,
BASELINE configs[4]); the only collective is the RCCL all_gather of per-chain energies.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHAINS_PER_GPU = 256
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak (no xf32 on gfx950)
F = 128


def load_golden():
    g = os.path.join(ROOT, "tests", "golden")
    blobs = [np.fromfile(os.path.join(g, "weights", f"SrTiO3_painn_model0{m}.f32"), dtype="<f4") for m in (1, 2, 3)]
    S = np.load(os.path.join(g, "structures.npz"))
    with open(os.path.join(g, "offset_data.json")) as fh:
        offset_data = json.load(fh)
    return blobs, S, offset_data


def build_chains(S, first, count):
    from surface_sampling_amd import structures

    k = "SrTiO3_2x2_pristine"
    base = structures.Structure(S[f"{k}.numbers"], S[f"{k}.positions"], S[f"{k}.cell"], S[f"{k}.pbc"])
    big = base.repeat((2, 2, 1))
    return [structures.synth_chain(big, c) for c in range(first, first + count)]


def neighbor_sum_bytes(n_atoms, n_edges, n_models):
    """Algorithmic HBM bytes of ONE forward neighbor-sum launch (layer >= 1), SURVEY.md §8(d):
    read phi [N,3F], v [N,3,F], s [N,F] + 16 B per edge; write s' [N,F], v' [N,3,F]."""
    per_atom = (3 * F + 3 * F + F + F + 3 * F) * 4
    return n_models * (per_atom * n_atoms + 16 * n_edges)


def neighbor_sum_flops(n_slots, n_models):
    """fp32 FLOPs of ONE forward neighbor-sum launch: per slot and model the radial filter (3F x 21 GEMV, bias column
    included) plus the message arithmetic of painn_edge_mfma.hip (18 flops per feature: 3 products, 7 fma)."""
    return n_models * n_slots * (3 * F * 21 * 2 + F * 18)


def reverse_pass_bytes(n_atoms, n_edges, n_models):
    """Algorithmic HBM bytes of ONE reverse neighbor-pass launch: read sbar, vbar [N,4F], phi [N,3F], v [N,3,F];
    write phibar [N,3F], vbar_in [N,3,F]; per edge 16 B geometry in, 16 B edge gradient out."""
    return n_models * ((4 * F + 3 * F + 3 * F + 3 * F + 3 * F) * 4 * n_atoms + 32 * n_edges)


def reverse_pass_flops(n_slots, n_models):
    """fp32 FLOPs of ONE reverse neighbor-pass launch: the filter AND its radial derivative (2 x 3F x 21 GEMV) plus
    the per-feature adjoint arithmetic (36 flops per feature, `feature()` in k_edge_bwd_mfma).  These are the
    fp32-precision flops the algorithm needs; the kernel executes each GEMV as six bf16 partial products."""
    return n_models * n_slots * (2 * 3 * F * 21 * 2 + F * 36)


def measured_traffic(kernel):
    """HBM bytes per launch from committed PMC passes (profiles/: FETCH_SIZE, WRITE_SIZE in KB, separate passes;
    FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM for 16-B/lane streams).  None when no profile is committed."""
    path = os.path.join(ROOT, "profiles", "r01", "pmc_traffic_edge_kernels_v10.json")
    if not os.path.exists(path):
        return None
    raw = json.load(open(path))
    tot, n = 0.0, 0
    for name, d in raw.items():
        if kernel in name and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            k = d["FETCH_SIZE"]["n"]
            tot += k * (2.0 * d["FETCH_SIZE"]["mean_raw"] + d["WRITE_SIZE"]["mean_raw"]) * 1024.0
            n += k
    return tot / n if n else None


def cpu_baseline(blobs, chains, table, const, budget_s=20.0):
    """Time the CPU oracle (a port: the reference's CPU path is not installable) on a bounded
    sample of the same workload.  The oracle is only the reported baseline / checker, never the product."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle

    oracle.build()
    n, t0 = 0, time.perf_counter()
    threads = oracle.set_threads(min(os.cpu_count() or 1, 32))
    s = chains[0]
    oracle.ensemble(blobs, s.numbers, s.positions, s.cell, s.pbc, 32, table, const)  # warm-up (page-in, threads)
    t0 = time.perf_counter()
    while n < len(chains) and (time.perf_counter() - t0 < budget_s or n < 4):
        s = chains[n]
        oracle.ensemble(blobs, s.numbers, s.positions, s.cell, s.pbc, 32, table, const)
        n += 1
    dt = time.perf_counter() - t0
    cores = threads
    return {"value": n / dt, "unit": "evaluations/s", "cores": cores, "kind": "port",
            "sample": f"{n} chains of the same workload (fp32 oracle, OpenMP, {dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chains-per-gpu", type=int, default=CHAINS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="split this GPU's chains over S engines (own HIP streams) that run concurrently; default 1 keeps "
                         "the per-kernel launch durations of the roofline free of overlap (DESIGN.md section 5)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch

    from surface_sampling_amd import backend
    from surface_sampling_amd.calculators import stoich_offset_table
    from surface_sampling_amd.sharding import chain_range

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    blobs, S, offset_data = load_golden()
    table, const = stoich_offset_table(offset_data)
    B = args.chains_per_gpu
    first, count = chain_range(world * B, world, rank)   # block partition of the global chain list
    chains = build_chains(S, first, count)

    n_str = max(1, min(args.streams, count))
    engs = []
    for k in range(n_str):   # block partition of this rank's chains over its engines
        lo, hi = (k * count) // n_str, ((k + 1) * count) // n_str
        e = backend.PainnEngine(blobs, device=local_rank, offset_per_z=table, offset_const=const)
        e.upload([(s.numbers, s.positions, s.cell, s.pbc) for s in chains[lo:hi]])   # inputs resident in HBM
        engs.append(e)
    want = backend.WANT_ENERGY | backend.WANT_FORCES | backend.WANT_STD
    dev = torch.device("cuda", local_rank)
    gathered = torch.empty(world * count * 2, dtype=torch.float32, device=dev) if world > 1 else None

    def step():
        for e in engs:
            e.run(want)
        if world > 1:   # the path's only exchange: per-chain (E_mean, E_std) to every rank
            parts = [e.download(backend.WANT_ENERGY | backend.WANT_STD) for e in engs]
            mine = torch.from_numpy(np.concatenate([p["energy"] for p in parts] + [p["energy_std"] for p in parts])).to(dev)
            dist.all_gather_into_tensor(gathered, mine)

    def fence():
        for e in engs:
            e.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    for e in engs:
        e.profile_enable(True)
        e.profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = {}
    for e in engs:   # HIP-event times of every engine's own stream, summed per kernel class
        for name, v in e.profile_read().items():
            acc = prof.setdefault(name, {"launches": 0, "total_ms": 0.0})
            acc["launches"] += v["launches"]
            acc["total_ms"] += v["total_ms"]
        e.profile_enable(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    stats = {"atoms": 0, "edges": 0, "slots": 0}
    for e in engs:
        for k, v in e.stats().items():
            stats[k] += v
        res = e.download(want)
        if not (np.isfinite(res["energy"]).all() and np.isfinite(res["forces"]).all()):
            raise SystemExit("non-finite results in the timed region")

    if rank == 0:
        total_evals = world * count * args.steps
        value = total_evals / elapsed
        M = len(blobs)
        # Dominant kernel: the reverse neighbor pass (k_edge_bwd_mfma, layers 2 and 1; layer 0 runs through the species
        # factorisation and has its own profiler class).  Launch durations are HIP-event times on the handle's stream.
        def kernel_view(cls, flops, nbytes):
            k = prof.get(cls, {"launches": 0, "total_ms": 0.0})
            ms = k["total_ms"] / max(1, k["launches"])
            return {"avg_launch_ms": ms, "launches": k["launches"], "algorithmic_flops_per_launch": flops,
                    "achieved_TFLOPs": flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                    "algorithmic_bytes_per_launch": nbytes,
                    "achieved_GBps": nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0}

        # (per launch = per engine: with --streams S a launch covers 1/S of this GPU's chains and overlaps the other engines)
        bwd = kernel_view("edge_message_bwd", reverse_pass_flops(stats["slots"], M) / n_str,
                          reverse_pass_bytes(stats["atoms"], stats["edges"], M) / n_str)
        fwd = kernel_view("edge_message_fwd", neighbor_sum_flops(stats["slots"], M) / n_str,
                          neighbor_sum_bytes(stats["atoms"], stats["edges"], M) / n_str)
        step_ms = sum(v["total_ms"] for v in prof.values()) / args.steps
        line = {
            "metric": "MC energy-evaluations/sec (SrTiO3(001) ~250-atom slabs, 3-model PaiNN ensemble E+F incl. neighbor list)",
            "value": value, "unit": "evaluations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"SrTiO3(001) PaiNN x3, {count} batched independent chains per GPU "
                                   f"(BASELINE configs[{3 if world == 1 else 4}]), 248-272 atoms/chain",
                       "chains_per_gpu": count, "atoms_per_gpu": stats["atoms"], "edges_per_gpu": stats["edges"],
                       "streams_per_gpu": n_str,
                       "parallelism": f"chains sharded x{world}, RCCL all_gather of per-chain energies"},
            # The reverse neighbor pass is instruction / matrix-pipe bound, not HBM bound (its HBM view is given beside
            # it): `achieved` = fp32-precision algorithmic TFLOP/s (filter GEMVs + adjoint arithmetic, formulas above)
            # against the fp32 peak of MI355X_MICROARCH.md; the filter runs as 3 fp16 partial products per GEMV on
            # the 16-bit MFMA pipe (2-way split, fp32-level accuracy), the rest on the fp32 VALU.
            "roofline": {"bound": "mfma", "kernel": "edge_message_bwd (reverse neighbor pass, k_edge_bwd_mfma)",
                         "achieved": bwd["achieved_TFLOPs"], "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": bwd["achieved_TFLOPs"] / MFMA_F32_PEAK_TFLOPS,
                         "traffic": measured_traffic("k_edge_bwd_mfma"),
                         "algorithmic_flops_per_launch": bwd["algorithmic_flops_per_launch"],
                         "avg_launch_ms": bwd["avg_launch_ms"], "launches": bwd["launches"],
                         "hbm_view": {"algorithmic_bytes_per_launch": bwd["algorithmic_bytes_per_launch"],
                                      "achieved_GBps": bwd["achieved_GBps"], "peak_GBps": HBM_PEAK_GBS,
                                      "frac": bwd["achieved_GBps"] / HBM_PEAK_GBS},
                         "second_kernel": {"kernel": "edge_message_fwd (neighbor-sum, k_edge_fwd_mfma)",
                                           "achieved": fwd["achieved_TFLOPs"],
                                           "frac": fwd["achieved_TFLOPs"] / MFMA_F32_PEAK_TFLOPS,
                                           "avg_launch_ms": fwd["avg_launch_ms"],
                                           "traffic": measured_traffic("k_edge_fwd_mfma"),
                                           "algorithmic_bytes_per_launch": fwd["algorithmic_bytes_per_launch"]}},
            "kernel_ms_per_step": {k: v["total_ms"] / args.steps for k, v in prof.items() if v["launches"]},
            "device_ms_per_step": step_ms,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(blobs, chains, table, const)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    for e in engs:
        e.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
