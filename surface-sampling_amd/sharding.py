"""Chain sharding across the GPUs of one node.

Chains never interact: GPU g owns the contiguous block ``chain_range(B, G, g)`` of the global chain
list, weights are replicated, and the only exchange is a gather of per-chain scalars (energies,
flags) — one process per GPU, ``torch.distributed`` with the ``nccl`` (= RCCL over xGMI) backend on
the GPU box, ``gloo`` in CPU tests.  Forces stay on the owning GPU.
"""

from __future__ import annotations

import numpy as np


def chain_range(n_chains: int, world: int, rank: int) -> tuple[int, int]:
    """(first, count) of the block of chains owned by ``rank``; earlier ranks take the remainder."""
    if not (0 <= rank < world) or n_chains < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(n_chains, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def all_ranges(n_chains: int, world: int) -> list[tuple[int, int]]:
    return [chain_range(n_chains, world, r) for r in range(world)]


def gather_chain_scalars(local: np.ndarray, n_chains: int, dist=None, device=None) -> np.ndarray:
    """All ranks receive the per-chain array of every rank, in global chain order.

    ``local``: float32 [count, k] (or [count]) for this rank's block.  Uneven blocks are padded to the
    largest block for the collective and trimmed afterwards."""
    import torch

    local = np.ascontiguousarray(local, dtype=np.float32)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local.copy()
    world, rank = dist.get_world_size(), dist.get_rank()
    ranges = all_ranges(n_chains, world)
    if local.shape[0] != ranges[rank][1]:
        raise ValueError("local block does not match this rank's chain range")
    width = int(np.prod(local.shape[1:])) if local.ndim > 1 else 1
    cmax = max(c for _, c in ranges)
    buf = torch.zeros(cmax * width, dtype=torch.float32, device=device)
    buf[: local.size] = torch.from_numpy(local.reshape(-1)).to(buf.device)
    out = torch.empty(world * cmax * width, dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(out, buf)
    out = out.cpu().numpy().reshape(world, cmax, width)
    parts = [out[r, :c] for r, (_, c) in enumerate(ranges)]
    full = np.concatenate(parts, axis=0)
    return full.reshape((n_chains,) + local.shape[1:])


class ShardedEnsemble:
    """Runs one engine per rank on that rank's block of chains and gathers per-chain energies."""

    def __init__(self, engine, n_chains: int, dist=None, device=None):
        self.engine, self.n_chains, self.dist, self.device = engine, n_chains, dist, device
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.first, self.count = chain_range(n_chains, self.world, self.rank)

    def local_slice(self, chains: list) -> list:
        return chains[self.first:self.first + self.count]

    def evaluate(self, all_chains: list) -> dict:
        """``all_chains``: the global list (every rank passes the same list; only its block is evaluated).
        Returns global per-chain ``energy`` / ``energy_std`` on every rank and this rank's ``forces``."""
        res = self.engine.evaluate(self.local_slice(all_chains))
        scal = np.stack([res["energy"], res["energy_std"]], axis=1)
        full = gather_chain_scalars(scal, self.n_chains, self.dist, self.device)
        return {"energy": full[:, 0], "energy_std": full[:, 1], "forces_local": res["forces"],
                "cfg_start_local": res["cfg_start"], "first": self.first, "count": self.count}
