"""One rank of the world-size-2 GPU test (tests/test_sharding.py::test_two_ranks_share_one_gpu): started by the test as a
FRESH interpreter (``subprocess.Popen([sys.executable, this file, ...])``), both ranks on device 0, ``gloo`` process group,
host result path.  Writes ``<out>/rank<r>.npz``."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def site_grid(base, n=4, dz=1.5):
    ztop = base.positions[:, 2].max()
    a, b = base.cell[0], base.cell[1]
    pts = []
    for i in range(n):
        for j in range(n):
            p = (i + 0.5) / n * a + (j + 0.5) / n * b
            pts.append([p[0], p[1], ztop + dz])
    return np.array(pts, float)


def run_mc(golden, n_chains, first_chain, steps=3):
    from surface_sampling_amd import mc
    from surface_sampling_amd.calculators import EnsembleNFFSurface

    base = golden.structure("SrTiO3_2x2_pristine")
    fixed = np.flatnonzero(base.positions[:, 2] < base.positions[:, 2].max() - 4.0)
    calc = EnsembleNFFSurface(golden.blobs, device="cuda:0")
    calc.set(offset=True, offset_data=golden.offset_data, chem_pots={"Sr": -2, "Ti": 0, "O": 0})
    ens = mc.ChainEnsemble(base, site_grid(base), ("Sr", "O"), n_chains, calc, seed=11, first_chain=first_chain, relax=True,
                           relax_steps=3, fmax=0.05, fixed_indices=fixed, temperature=0.5)
    ens.initialize()
    acc = [ens.step_semigrand() for _ in range(steps)]
    return np.array(acc), ens.state.species.copy(), ens.state.energy.copy()


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch.distributed as dist

    import bench
    from conftest import Golden
    from surface_sampling_amd import backend, sharding
    from surface_sampling_amd.calculators import stoich_offset_table

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    golden = Golden()
    # (i) one lock-step evaluation of a sharded chain list, gathered to every rank
    blobs, S, offset_data = bench.load_golden()
    table, const = stoich_offset_table(offset_data)
    n_total = 6
    chains = bench.build_chains(S, 40, n_total)
    packs = [(s.numbers, s.positions, s.cell, s.pbc) for s in chains]
    eng = backend.PainnEngine(blobs, device=0, offset_per_z=table, offset_const=const)
    sh = sharding.ShardedEnsemble(eng, n_total, dist)
    assert sh.result_path == "host" and sh.world == 2
    sh.upload(all_chains=packs)
    want = backend.WANT_ENERGY | backend.WANT_FORCES | backend.WANT_STD
    g1 = sh.step(want)
    g2 = sh.step(want)          # a second lock-step on the resident block
    gathered = np.asarray(g2.cpu().numpy() if hasattr(g2, "cpu") else g2)
    assert np.array_equal(np.asarray(g1.cpu().numpy() if hasattr(g1, "cpu") else g1), gathered)
    one_shot = sh.evaluate(packs)
    eng.close()
    # (ii) batched MC with global chain ids: this rank owns chains [rank * n, rank * n + n)
    n = 3
    acc, species, energy = run_mc(golden, n, rank * n)
    np.savez(os.path.join(out, f"rank{rank}.npz"), gathered=gathered, one_shot_energy=one_shot["energy"],
             first=sh.first, count=sh.count, acc=acc, species=species, energy=energy)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
