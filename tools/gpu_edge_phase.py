"""Debug: where a wave of the two neighbor-sum kernels spends its time inside the hot loop (library built with -DEDGE_PHASE_TIMING,
see painn_edge_mfma.hip; wall-clock ticks summed over all waves, so the shares are what matters, and every mark costs a clock read)."""
import ctypes, os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench
from surface_sampling_amd import backend
from surface_sampling_amd.calculators import stoich_offset_table
blobs, S, offset_data = bench.load_golden()
table, const = stoich_offset_table(offset_data)
chains = bench.build_chains(S, 0, 256)
eng = backend.PainnEngine(blobs, device=0, offset_per_z=table, offset_const=const)
eng.upload([(s.numbers, s.positions, s.cell, s.pbc) for s in chains])
lib = backend.load_library()
buf = (ctypes.c_ulonglong * 16)()
eng.run(); eng.synchronize()
lib.vssr_debug_edge_phases(buf, 1)
for _ in range(3):
    eng.run()
eng.synchronize()
lib.vssr_debug_edge_phases(buf, 0)
v = np.array(list(buf), dtype=np.float64)
names = ["(clock start)", "wait for the step's table entries", "bundle completion (1 step in ~11)", "gather issue + LDS wait",
         "matrix instructions issued", "matrix results + first feature", "prefetch issue + remaining features", "reduce-scatter + store"]
if os.environ.get("VSSR_EDGE_BWD_32", "0") != "0":   # the 32x32x16 reverse kernel marks other phases
    names32 = ["", "wait for the step's table entries", "bundle completion", "group 0: gathers + matrix instructions issued",
               "group 0: matrix results + vector work", "group 1: gathers + matrix instructions issued",
               "group 1: matrix results + refill issue + vector work", "reduce-scatter + store"]
    tot = v[9:16].sum()
    print("k_edge_bwd_mfma32")
    for k in range(1, 8):
        print(f"   {names32[k]:52s} {100 * v[8 + k] / tot:6.1f} %")
    sys.exit(0)
for lo, kern in ((0, "k_edge_fwd_mfma"), (8, "k_edge_bwd_mfma")):
    tot = v[lo + 1:lo + 8].sum()
    print(kern)
    for k in range(1, 8):
        if v[lo + k]:
            print(f"   {names[k]:44s} {100 * v[lo + k] / tot:6.1f} %")
eng.close()
