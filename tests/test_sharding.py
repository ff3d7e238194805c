"""CPU, world_size 2 over gloo: the chain-sharding path used for N > 1 GPUs."""

import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_chain_range_partition():
    from surface_sampling_amd.sharding import all_ranges, chain_range

    for n, w in [(2048, 8), (256, 1), (10, 4), (3, 8), (0, 2)]:
        r = all_ranges(n, w)
        assert r[0][0] == 0 and sum(c for _, c in r) == n
        for (f0, c0), (f1, _) in zip(r, r[1:]):
            assert f0 + c0 == f1
        assert max(c for _, c in r) - min(c for _, c in r) <= 1
    with pytest.raises(ValueError):
        chain_range(8, 2, 2)


class _FakeEngine:
    """Stands in for the GPU engine: energy = f(chain content), so ordering mistakes are visible."""

    def evaluate(self, structs):
        e = np.array([float(np.sum(s[1])) for s in structs], dtype=np.float32)
        n = [len(s[0]) for s in structs]
        return {"energy": e, "energy_std": 0.5 * e, "forces": np.zeros((sum(n), 3), np.float32),
                "cfg_start": np.concatenate([[0], np.cumsum(n)])}


def _worker(rank, world, port, n_chains, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from surface_sampling_amd.sharding import ShardedEnsemble

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)
    chains = [(np.ones(3 + c % 4, np.int32), rng.normal(size=(3 + c % 4, 3)), np.eye(3), [1, 1, 1])
              for c in range(n_chains)]
    sh = ShardedEnsemble(_FakeEngine(), n_chains, dist)
    res = sh.evaluate(chains)
    want = np.array([float(np.sum(c[1])) for c in chains], dtype=np.float32)
    ok = np.array_equal(res["energy"], want) and np.array_equal(res["energy_std"], 0.5 * want)
    ok = ok and res["count"] == len(sh.local_slice(chains))
    np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([ok, res["first"], res["count"]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_chains", [8, 7])
def test_gather_world_size_2_gloo(tmp_path, n_chains):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, n_chains, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / f"ok{r}.npy") for r in (0, 1))
    assert r0[0] == 1 and r1[0] == 1
    assert r0[1] == 0 and r0[1] + r0[2] == r1[1] and r1[1] + r1[2] == n_chains
