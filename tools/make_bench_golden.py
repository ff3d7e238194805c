"""Generates tests/golden/bench_batch_fp64.npz: the CPU oracle's answers for ALL 256 chains of the benchmark batch (BASELINE
configs[3] = bench.build_chains(S, 0, 256): 248-272 atoms each, 3-model ensemble, offset on).

    python tools/make_bench_golden.py            (build container or any host; ~5 min on 8 cores; no GPU, no reference tree)

Stored: the fp64 oracle's ensemble energy, spread and forces per chain (the parity reference of tests/test_gpu_parity.py::
test_whole_bench_batch_against_the_committed_fp64_vectors), the same three from the oracle's fp32 mode (what plain fp32
arithmetic of the same algorithm gives: the yardstick for the device's fp16-split arithmetic), and a checksum of the inputs so
that the test knows it evaluates the configurations the vectors were made from.  The oracle is test infrastructure."""
import hashlib, os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import bench, oracle
from surface_sampling_amd.calculators import stoich_offset_table

N_CHAINS = 256


def inputs_digest(chains):
    h = hashlib.sha256()
    for s in chains:
        h.update(np.ascontiguousarray(s.numbers, dtype=np.int32).tobytes())
        h.update(np.ascontiguousarray(s.positions, dtype=np.float64).tobytes())
        h.update(np.ascontiguousarray(s.cell, dtype=np.float64).tobytes())
    return h.hexdigest()


def main():
    oracle.build(); oracle.set_threads(min(os.cpu_count() or 1, 64))
    blobs, S, offset_data = bench.load_golden()
    table, const = stoich_offset_table(offset_data)
    chains = bench.build_chains(S, 0, N_CHAINS)
    cfg_start = np.concatenate([[0], np.cumsum([len(s.numbers) for s in chains])]).astype(np.int64)
    out = {}
    for bits in (64, 32):
        t0 = time.time()
        E = np.zeros(N_CHAINS); Es = np.zeros(N_CHAINS); Fm = np.zeros((cfg_start[-1], 3))
        for b, s in enumerate(chains):
            r = oracle.ensemble(blobs, s.numbers, s.positions, s.cell, s.pbc, bits, table, const)
            E[b], Es[b] = r["energy"], r["energy_std"]
            Fm[cfg_start[b]:cfg_start[b + 1]] = r["forces"]
        print(f"fp{bits} oracle: {time.time() - t0:.1f} s", file=sys.stderr)
        out[bits] = (E, Es, Fm)
    dE = np.abs(out[32][0] - out[64][0]); dF = np.abs(out[32][2] - out[64][2])
    print(f"fp32-mode oracle vs fp64 oracle: max |dE| {dE.max():.3e} eV  max |dF| {dF.max():.3e} eV/A  max |dEstd| "
          f"{np.abs(out[32][1] - out[64][1]).max():.3e}", file=sys.stderr)
    np.savez_compressed(os.path.join(root, "tests", "golden", "bench_batch_fp64.npz"),
                        cfg_start=cfg_start, inputs_sha256=np.array(inputs_digest(chains)),
                        energy=out[64][0], energy_std=out[64][1], forces=out[64][2],
                        energy_fp32mode=out[32][0], energy_std_fp32mode=out[32][1], forces_fp32mode=out[32][2].astype(np.float32))


if __name__ == "__main__":
    main()
