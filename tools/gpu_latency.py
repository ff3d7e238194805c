"""Single-chain latency of the drop-in path (one evaluation per call, host arrays in / out), and small batches."""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench
from surface_sampling_amd import backend
from surface_sampling_amd.calculators import stoich_offset_table
blobs, S, offset_data = bench.load_golden()
table, const = stoich_offset_table(offset_data)
chains = bench.build_chains(S, 0, 64)
eng = backend.PainnEngine(blobs, device=0, offset_per_z=table, offset_const=const)
for nb in (1, 4, 16, 64):
    batch = [(s.numbers, s.positions, s.cell, s.pbc) for s in chains[:nb]]
    for _ in range(5):
        eng.evaluate(batch)
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        eng.evaluate(batch)
    dt = (time.perf_counter() - t0) / n
    # resident batch: upload once, then positions in / results out per call (the relaxation / MC inner loop)
    eng.upload(batch)
    pos = np.concatenate([np.asarray(b[1], float) for b in batch])
    want = backend.WANT_ENERGY | backend.WANT_FORCES | backend.WANT_STD
    for _ in range(5):
        eng.set_positions(pos); eng.run(want); eng.download(want)
    t0 = time.perf_counter()
    for _ in range(n):
        eng.set_positions(pos); eng.run(want); eng.download(want)
    dr = (time.perf_counter() - t0) / n
    print(f"batch {nb:3d}: {1e3 * dt:7.3f} ms per evaluate() call, {1e3 * dr:7.3f} ms per resident step -> {nb / dr:9.1f} evaluations/s")
eng.close()
# where a single-chain evaluation spends its time: host enqueue (vssr_batch_run returns after the launches) vs the whole step
eng = backend.PainnEngine(blobs, device=0, offset_per_z=table, offset_const=const)
batch = [(s.numbers, s.positions, s.cell, s.pbc) for s in chains[:1]]
eng.upload(batch)
want = backend.WANT_ENERGY | backend.WANT_FORCES | backend.WANT_STD
for _ in range(10):
    eng.run(want); eng.synchronize()
t_enq = t_all = 0.0
for _ in range(100):
    t0 = time.perf_counter(); eng.run(want); t1 = time.perf_counter(); eng.synchronize(); t2 = time.perf_counter()
    t_enq += t1 - t0; t_all += t2 - t0
print(f"1 chain, resident: enqueue {1e4 * t_enq:.1f} us, enqueue + wait {1e4 * t_all:.1f} us per evaluation")
eng.profile_enable(True); eng.profile_reset()
for _ in range(20):
    eng.run(want)
eng.synchronize()
print("kernel classes (us per evaluation, launches):", {k: (round(1e3 * v["total_ms"] / 20, 1), v["launches"] // 20) for k, v in eng.profile_read().items()})
eng.close()
