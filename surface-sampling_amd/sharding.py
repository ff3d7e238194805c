"""Chain sharding across the GPUs of one node.

Chains never interact: GPU g owns the contiguous block ``chain_range(B, G, g)`` of the global chain
list, weights are replicated, and the only exchange is a gather of per-chain scalars (energies,
flags) — one process per GPU, ``torch.distributed`` with the ``nccl`` (= RCCL over xGMI) backend on
the GPU box, ``gloo`` in CPU tests.  Forces stay on the owning GPU.

``ShardedEnsemble`` is the one code path for N = 1 and N > 1: ``bench.py`` times its ``step()``, the
world-size-2 gloo test runs the same ``step()`` with a stand-in engine.
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


def _world(dist):
    return dist.get_world_size() if dist is not None and dist.is_initialized() else 1


def gather_chain_scalars(local, n_chains: int, dist=None, device=None, keep_on_device: bool = False):
    """All ranks receive the per-chain array of every rank, in global chain order.

    ``local``: float32 ``[count, k]`` (or ``[count]``) for this rank's block — a numpy array, or a torch tensor that may
    already live on the GPU (then nothing passes through the host before the collective).  Uneven blocks are padded to the
    largest block for the collective and trimmed afterwards.  Returns a numpy array, or with ``keep_on_device`` the
    gathered torch tensor ``[n_chains, k]``."""
    import torch

    is_tensor = isinstance(local, torch.Tensor)
    world = _world(dist)
    if world == 1:
        if is_tensor:
            return local.clone() if keep_on_device else local.detach().cpu().numpy().copy()
        return np.ascontiguousarray(local, dtype=np.float32).copy()
    rank = dist.get_rank()
    ranges = all_ranges(n_chains, world)
    shape_tail = tuple(local.shape[1:])
    if int(local.shape[0]) != ranges[rank][1]:
        raise ValueError("local block does not match this rank's chain range")
    width = int(np.prod(shape_tail)) if shape_tail else 1
    cmax = max(c for _, c in ranges)
    if is_tensor:
        flat = local.to(dtype=torch.float32).reshape(-1)
        dev = flat.device if device is None else device
    else:
        flat = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float32).reshape(-1))
        dev = device
    buf = torch.zeros(cmax * width, dtype=torch.float32, device=dev)
    buf[: flat.numel()] = flat.to(buf.device)
    out = torch.empty(world * cmax * width, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, buf)
    out = out.reshape(world, cmax, width)
    full = torch.cat([out[r, :c] for r, (_, c) in enumerate(ranges)], dim=0).reshape((n_chains,) + shape_tail)
    return full if keep_on_device else full.cpu().numpy()


class ShardedEnsemble:
    """One engine per rank on that rank's block of chains; per-chain energies gathered to every rank.

    The engine needs ``upload(structs)``, ``run(want)``, ``synchronize()``, ``download(want)`` and, for the device result
    path, ``device_results()`` + ``device_context()`` (zero-copy device arrays and the engine's stream: the collective reads
    the energies where the kernels wrote them).

    ``result_path``: ``"device"`` -- the gather input never passes through the host and nothing synchronises the host: the
    staging copy waits for the engine's stream on the device, the engine's NEXT run waits for the staging copy, so step n + 1
    is enqueued while the gather of step n is still in flight (two staging buffers).  A neighbor-capacity overflow of a
    run cannot be seen without the host: its flag travels with the results (third column) and :meth:`check` repairs it.
    ``"host"`` -- ``download`` + numpy (the CPU stand-in of the tests; ``gloo`` groups; fp64 engines without device results).
    ``"auto"``: device for an ``nccl`` group or a single rank when the engine can, host otherwise."""

    def __init__(self, engine, n_chains: int, dist=None, device=None, result_path: str = "auto"):
        self.engine, self.n_chains, self.dist, self.device = engine, n_chains, dist, device
        self.world = _world(dist)
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.first, self.count = chain_range(n_chains, self.world, self.rank)
        self.gathered = None
        if result_path not in ("auto", "device", "host"):
            raise ValueError("result_path must be 'auto', 'device' or 'host'")
        can_device = hasattr(self.engine, "device_results") and hasattr(self.engine, "device_context")
        if result_path == "auto":
            backend_name = dist.get_backend() if self.world > 1 else "nccl"
            result_path = "device" if can_device and backend_name == "nccl" else "host"
        if result_path == "device" and not can_device:
            raise ValueError("this engine has no device result path")
        self.result_path = result_path
        self._staging, self._flip, self._ext = None, 0, None

    def local_slice(self, chains: list) -> list:
        return chains[self.first:self.first + self.count]

    # ---- resident batch: upload once, then lock-step evaluations ------------------------------------------
    def upload(self, all_chains: list | None = None, local_chains: list | None = None):
        """Make this rank's block resident (pass the global list or the block itself)."""
        block = local_chains if local_chains is not None else self.local_slice(all_chains)
        if len(block) != self.count:
            raise ValueError("block does not match this rank's chain range")
        self.engine.upload(block)
        self._staging = None

    def _device_scalars(self):
        """This rank's per-chain (E, sigma_E, overflow flag) as a ``[count, 3]`` tensor on the ENGINE's device, ordered
        behind the engine's stream by events only."""
        import torch

        ordinal, stream_ptr, flag_ptr = self.engine.device_context()
        dev = torch.device("cuda", ordinal)     # the engine's device, whatever torch's current device is
        try:
            e, s = self.engine.device_results()
        except Exception:
            return None                          # (an engine kind without device results: host path)
        if self._ext is None or self._ext[0] != stream_ptr:
            self._ext = (stream_ptr, torch.cuda.ExternalStream(stream_ptr, device=dev))
        ext = self._ext[1]
        if self._staging is None or self._staging[0].shape[0] != self.count or self._staging[0].device != dev:
            self._staging = [torch.zeros(self.count, 3, dtype=torch.float32, device=dev) for _ in range(2)]
        self._flip ^= 1
        buf = self._staging[self._flip]
        cur = torch.cuda.current_stream(dev)
        cur.wait_stream(ext)                     # device side: behind everything the engine has enqueued
        with torch.cuda.device(dev):
            buf[:, 0].copy_(torch.as_tensor(e, device=dev), non_blocking=True)
            buf[:, 1].copy_(torch.as_tensor(s, device=dev), non_blocking=True)
            if flag_ptr:
                from .backend import _DeviceArray

                flag = torch.as_tensor(_DeviceArray(flag_ptr, 1, "<i4"), device=dev)
                buf[:, 2] = flag.to(torch.float32)
        ext.wait_stream(cur)                     # the engine's next run overwrites its result buffers only after this copy
        return buf

    def _local_scalars(self, want_energy_flags):
        """This rank's per-chain (E, sigma_E[, overflow flag]) ``[count, 2 or 3]``: device tensor or numpy array."""
        if self.result_path == "device":
            out = self._device_scalars()
            if out is not None:
                return out
        res = self.engine.download(want_energy_flags)   # synchronises, repairs a capacity overflow
        return np.stack([res["energy"], res["energy_std"]], axis=1)

    def step(self, want, gather: bool | None = None):
        """One lock-step evaluation of the resident block (asynchronous on one GPU) and, when the chains are sharded, the
        path's only exchange: per-chain (E, sigma_E) to every rank.  Returns the gathered ``[n_chains, 2 or 3]`` (tensor or
        array; device path: stream-ordered, not yet complete on return) or None when nothing was gathered."""
        self.engine.run(want)
        if gather is None:
            gather = self.world > 1
        if not gather:
            return None
        from . import backend

        scal = self._local_scalars(backend.WANT_ENERGY | backend.WANT_STD)
        self.gathered = gather_chain_scalars(scal, self.n_chains, self.dist, self.device, keep_on_device=True)
        return self.gathered

    def check(self) -> bool:
        """Device result path: did any rank's last gathered evaluation overflow its neighbor capacity?  If so every rank
        repairs (``engine.synchronize()`` reruns with grown buffers) and gathers again; returns True when that happened.
        The host path repairs inside ``download`` and always returns False."""
        g = self.gathered
        if g is None or self.result_path != "device" or g.shape[1] < 3:
            return False
        if not bool((g[:, 2] != 0).any().item()):
            return False
        from . import backend

        self.engine.synchronize()
        scal = self._local_scalars(backend.WANT_ENERGY | backend.WANT_STD)
        self.gathered = gather_chain_scalars(scal, self.n_chains, self.dist, self.device, keep_on_device=True)
        return True

    # ---- one-shot form -------------------------------------------------------------------------------------
    def evaluate(self, all_chains: list) -> dict:
        """``all_chains``: the global list (every rank passes the same list; only its block is evaluated).
        Returns global per-chain ``energy`` / ``energy_std`` on every rank and this rank's ``forces``."""
        res = self.engine.evaluate(self.local_slice(all_chains))
        scal = np.stack([res["energy"], res["energy_std"]], axis=1)
        full = gather_chain_scalars(scal, self.n_chains, self.dist, self.device)
        return {"energy": full[:, 0], "energy_std": full[:, 1], "forces_local": res["forces"],
                "cfg_start_local": res["cfg_start"], "first": self.first, "count": self.count}
