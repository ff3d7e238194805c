"""Debug: phase timing of the update kernels (library built with -DPHASE_TIMING, see painn_node_mfma.hip)."""
import ctypes, json, os, sys
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
want = backend.WANT_ENERGY | backend.WANT_FORCES | backend.WANT_STD
lib = backend.load_library()
buf = (ctypes.c_ulonglong * 64)()
eng.run(want); eng.synchronize()
lib.vssr_debug_phases(buf, 1)
for _ in range(3):
    eng.run(want)
eng.synchronize()
lib.vssr_debug_phases(buf, 0)
v = np.array(list(buf), dtype=np.float64)
names = {0: "fwd: load tiles + barrier", 1: "fwd: GEMM1 (U, V)", 2: "fwd: GEMM2a (W3 on s)", 3: "fwd: barrier", 4: "fwd: norms + plane stores + barrier",
         5: "fwd: GEMM2b (W3 on |Vv|) + swish stores", 6: "fwd: barrier", 7: "fwd: GEMM3 (W4)", 8: "fwd: barrier", 9: "fwd: T tile stores + barrier", 10: "fwd: coalesced residual pass",
         16: "bwd: load tiles + barrier", 17: "bwd: GEMM1", 18: "bwd: norms", 19: "bwd: barrier", 20: "bwd: GEMM2", 21: "bwd: barrier",
         22: "bwd: GEMM3", 23: "bwd: qb stores (scalar loads of sbar / vbar)", 24: "bwd: barrier", 25: "bwd: GEMM W4^T + stores", 26: "bwd: barrier",
         27: "bwd: GEMM W3^T", 28: "bwd: barrier", 29: "bwd: ab stores (scalar loads of vbar)", 30: "bwd: barrier", 31: "bwd: GEMM [U|V]^T",
         32: "bwd: barrier", 33: "bwd: tile + coalesced pass"}
names.update({40: "bwd head R: loads + barrier", 41: "bwd head R: readout GEMMs", 42: "bwd head P: loads + barrier",
              43: "bwd head P: GEMM W1, W2^T", 44: "bwd head P: barrier + v tile load", 45: "bwd head P: h1bar stores, barrier, GEMM W1^T"})
for lo, hi in ((0, 16), (16, 64)):
    tot = sum(v[k] for k in names if lo <= k < hi)
    for k in sorted(names):
        if lo <= k < hi:
            print(f"{names[k]:52s} {100 * v[k] / max(tot, 1):6.1f} %   {v[k]:14.0f} ticks")
    print()
eng.close()
