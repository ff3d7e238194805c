"""One ~70-atom chain, resident, run + synchronize in a loop (for rocprofv3 --kernel-trace --stats: kernels per evaluation and their durations)."""
import os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from conftest import Golden
from surface_sampling_amd import backend, structures
g = Golden(); table, const = g.offset_table()
base = g.structure("SrTiO3_2x2_pristine")
packs = [structures.as_arrays(structures.synth_chain(base, 0, grid=(4, 4)))]
eng = backend.PainnEngine(g.blobs, device=0, offset_per_z=table, offset_const=const)
eng.upload(packs)
for _ in range(20): eng.run(); eng.synchronize()
t0 = time.perf_counter()
for _ in range(200): eng.run(); eng.synchronize()
print("us per evaluation", 1e6 * (time.perf_counter() - t0) / 200)
eng.close()
