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


def _flag_array(ptr):
    from .backend import _DeviceArray

    return _DeviceArray(ptr, 1, "<i4")


def _world(dist):
    return dist.get_world_size() if dist is not None and dist.is_initialized() else 1


class ChainGather:
    """The path's only collective with every buffer allocated ONCE: per-chain rows of all ranks in global chain order.

    ``all_gather_into_tensor`` needs equal contributions, so a rank's block is padded to the largest block.  With even blocks
    (``n_chains % world == 0``: BASELINE configs[4], 256 chains on each of 8 GPUs) a contiguous float32 ``[count, width]``
    tensor on the right device is handed to the collective as it is and the collective's output IS the result -- no staging
    copy, no ``cat``, no allocation per step.  Uneven blocks go through a preallocated pad buffer and are compacted into a
    preallocated result.  Two output buffers alternate, so the result of step n stays intact while step n + 1 is gathered."""

    def __init__(self, n_chains: int, width: int, dist, device=None, dtype=None):
        import torch

        self.n_chains, self.width, self.dist = int(n_chains), int(width), dist
        self.dtype = torch.float32 if dtype is None else dtype
        self.world = _world(dist)
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.ranges = all_ranges(self.n_chains, self.world)
        self.count = self.ranges[self.rank][1]
        self.cmax = max(c for _, c in self.ranges) if self.ranges else 0
        self.even = all(c == self.cmax for _, c in self.ranges)
        self.device = torch.device("cpu") if device is None else torch.device(device)
        kw = dict(dtype=self.dtype, device=self.device)
        self._pad = torch.zeros(self.cmax, self.width, **kw)
        self._out = [torch.empty(self.world * self.cmax, self.width, **kw) for _ in range(2)]
        self._full = None if self.even else [torch.empty(self.n_chains, self.width, **kw) for _ in range(2)]
        self._flip = 0
        # Transport.  "direct": the device (or CPU) tensor goes to the collective -- RCCL on the GPU box, gloo on CPU tensors.
        # "host_bounce": a DEVICE-resident block under a backend that has no device all_gather (gloo: the one-GPU rehearsal of
        # the N > 1 path, RCCL refuses two ranks on one device): pinned staging D2H -> gloo -> H2D into the same device result
        # buffers.  Everything before and after the transport (event-ordered staging, double buffering, overflow column, the
        # result tensors) is the code of the RCCL path.
        self.transport = "direct"
        if self.world > 1 and self.device.type == "cuda" and dist.get_backend() != "nccl":
            self.transport = "host_bounce"
            pin = dict(dtype=self.dtype, device="cpu", pin_memory=True)
            self._h_in = torch.zeros(self.cmax, self.width, **pin)
            self._h_out = torch.empty(self.world * self.cmax, self.width, **pin)
        self.n_bounces = 0
        self.n_staging_copies = 0        # (tests: stays 0 on the even, device-resident path)
        self.force_collective = False    # (tests: a one-rank group still goes through the collective, RCCL sees the real buffers)

    def __call__(self, local):
        """``local``: float32 ``[count, width]`` -- numpy array or torch tensor.  Returns the gathered ``[n_chains, width]``
        torch tensor (one of this object's two result buffers; stream-ordered for device tensors)."""
        import torch

        if isinstance(local, torch.Tensor):
            t = local
        else:
            t = torch.from_numpy(np.ascontiguousarray(local)).to(self.dtype)
        if tuple(t.shape) != (self.count, self.width):
            raise ValueError("local block does not match this rank's chain range")
        self._flip ^= 1
        out = self._out[self._flip]
        if self.world == 1 and not (self.force_collective and self.dist is not None and self.dist.is_initialized()):
            out.copy_(t, non_blocking=True)
            return out
        if self.transport == "host_bounce":
            # (rehearsal transport: the host waits for the producer's stream here -- the RCCL path never does)
            self._h_in[: self.count].copy_(t, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
            self.dist.all_gather_into_tensor(self._h_out, self._h_in)
            out.copy_(self._h_out, non_blocking=True)
            self.n_bounces += 1
        else:
            direct = self.even and t.dtype == self.dtype and t.device == self.device and t.is_contiguous()
            if not direct:
                self._pad[: self.count].copy_(t, non_blocking=True)
                self.n_staging_copies += 1
                t = self._pad
            self.dist.all_gather_into_tensor(out, t)
        if self.even:
            return out
        full = self._full[self._flip]
        blocks = out.view(self.world, self.cmax, self.width)
        for r, (f, c) in enumerate(self.ranges):
            full[f:f + c].copy_(blocks[r, :c], non_blocking=True)
        return full


def gather_chain_scalars(local, n_chains: int, dist=None, device=None, keep_on_device: bool = False):
    """One-shot form of :class:`ChainGather` (allocates; the per-step path of ``ShardedEnsemble`` keeps a ``ChainGather``).

    ``local``: float32 / float64 ``[count, k]`` (or ``[count]``) for this rank's block — a numpy array, or a torch tensor that
    may already live on the GPU (then nothing passes through the host before the collective); the gathered array keeps the
    floating type.  Returns a numpy array, or with ``keep_on_device`` the gathered torch tensor ``[n_chains, k]``."""
    import torch

    is_tensor = isinstance(local, torch.Tensor)
    shape_tail = tuple(local.shape[1:])
    width = int(np.prod(shape_tail)) if shape_tail else 1
    if is_tensor:
        dtype = torch.float64 if local.dtype == torch.float64 else torch.float32
        flat = local.to(dtype=dtype).reshape(int(local.shape[0]), width)
        dev = flat.device if device is None else device
    else:
        local = np.asarray(local)
        dtype = torch.float64 if local.dtype == np.float64 else torch.float32
        flat = np.ascontiguousarray(local, dtype=np.float64 if dtype == torch.float64 else np.float32).reshape(int(local.shape[0]), width)
        dev = device
    full = ChainGather(n_chains, width, dist, dev, dtype)(flat).reshape((n_chains,) + shape_tail)
    return full if keep_on_device else full.cpu().numpy()


class EngineGroup:
    """One GPU's block of chains split over S engines (S C-ABI handles = S HIP streams) that run concurrently: the
    latency-bound node kernels of one part fill issue slots under the neighbor-sum kernels of the other (+2.5 .. 4 % on
    256 chains).  Same resident-batch interface as one engine; chains keep their order (engine k owns
    the k-th contiguous part), and a chain's results do not depend on the split (batching changes nothing, bit-exact)."""

    def __init__(self, engines):
        self.engines = list(engines)
        if not self.engines:
            raise ValueError("at least one engine")
        self.bounds = None
        self.has_device_results = all(getattr(e, "has_device_results", False) for e in self.engines)

    def upload(self, structs):
        n, s = len(structs), len(self.engines)
        self.bounds = [((k * n) // s, ((k + 1) * n) // s) for k in range(s)]
        for e, (lo, hi) in zip(self.engines, self.bounds):
            e.upload(structs[lo:hi])

    def set_positions(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        o = 0
        for e in self.engines:
            n = e._n_atoms
            e.set_positions(pos[o:o + n])
            o += n

    def run(self, want):
        for e in self.engines:
            e.run(want)

    def synchronize(self):
        for e in self.engines:
            e.synchronize()

    def download(self, want):
        parts = [e.download(want) for e in self.engines]
        out = {k: np.concatenate([p[k] for p in parts]) for k in parts[0] if k != "cfg_start"}
        starts = [0]
        for p in parts:
            starts.extend((np.asarray(p["cfg_start"][1:]) + starts[-1]).tolist())
        out["cfg_start"] = np.asarray(starts, dtype=np.int64)
        return out

    def stats(self):
        tot = {}
        for e in self.engines:
            for k, v in e.stats().items():
                tot[k] = tot.get(k, 0) + v
        return tot

    def device_parts(self):
        """``[(engine, first local chain, end)]`` for the device result path of ``ShardedEnsemble``."""
        return [(e, lo, hi) for e, (lo, hi) in zip(self.engines, self.bounds)]

    def device_context(self):
        return self.engines[0].device_context()

    def device_results(self):
        raise RuntimeError("an EngineGroup has one result buffer per engine: use device_parts()")

    def close(self):
        for e in self.engines:
            e.close()


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
    ``"auto"``: device for an ``nccl`` group or a single rank when the engine can, host otherwise.  ``"device"`` under a
    ``gloo`` group is the one-GPU rehearsal of the RCCL path: everything is the device path except the transport
    (``ChainGather.transport == "host_bounce"``).

    The gathered values are float64: the ensemble mean / spread as the device forms them, before the narrowing to the
    reference's float32 result word (``vssr_batch_device_results_f64`` / ``energy_f64``; an engine without them contributes
    its float32 energies widened)."""

    def __init__(self, engine, n_chains: int, dist=None, device=None, result_path: str = "auto"):
        self.engine, self.n_chains, self.dist, self.device = engine, n_chains, dist, device
        self.world = _world(dist)
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.first, self.count = chain_range(n_chains, self.world, self.rank)
        self.gathered = None
        if result_path not in ("auto", "device", "host"):
            raise ValueError("result_path must be 'auto', 'device' or 'host'")
        # The path is decided HERE, from properties every rank shares (engine kind, backend) -- never per step from a caught
        # exception: a rank that silently fell back to the host path would contribute a [count, 2] CPU block to a collective
        # whose other contributions are [count, 3] device blocks (advisor r3).
        can_device = (hasattr(self.engine, "device_results") and hasattr(self.engine, "device_context")
                      and getattr(self.engine, "has_device_results", True))
        if result_path == "auto":
            backend_name = dist.get_backend() if self.world > 1 else "nccl"
            result_path = "device" if can_device and backend_name == "nccl" else "host"
        if result_path == "device" and not can_device:
            raise ValueError("this engine has no device result path")
        self.result_path = result_path
        self._staging, self._flip, self._views = None, 0, {}
        self._gather = None
        self._steps_since_check = 0

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
        """This rank's per-chain (E, sigma_E, overflow flag) as a float64 ``[count, 3]`` tensor on the ENGINE's device, ordered
        behind the engine's stream(s) by events only.  Errors of the engine propagate (no fallback, see ``__init__``).
        An ``EngineGroup`` contributes one row range per engine."""
        import torch

        parts = self.engine.device_parts() if hasattr(self.engine, "device_parts") else [(self.engine, 0, self.count)]
        ordinal = parts[0][0].device_context()[0]
        dev = torch.device("cuda", ordinal)     # the engine's device, whatever torch's current device is
        if self._staging is None or self._staging[0].shape[0] != self.count or self._staging[0].device != dev:
            self._staging = [torch.zeros(self.count, 3, dtype=torch.float64, device=dev) for _ in range(2)]
            self._views = {}
        self._flip ^= 1
        buf = self._staging[self._flip]
        cur = torch.cuda.current_stream(dev)
        exts = []
        for k, (eng, lo, hi) in enumerate(parts):
            _, stream_ptr, flag_ptr = eng.device_context()
            e, s = eng.device_results_f64() if hasattr(eng, "device_results_f64") else eng.device_results()
            ptrs = (e.__cuda_array_interface__["data"][0], s.__cuda_array_interface__["data"][0], flag_ptr, stream_ptr)
            v = self._views.get(k)
            if v is None or v[0] != ptrs:      # zero-copy views, made once per resident batch (again after a capacity regrow)
                v = (ptrs, torch.as_tensor(e, device=dev), torch.as_tensor(s, device=dev),
                     None if not flag_ptr else torch.as_tensor(_flag_array(flag_ptr), device=dev),
                     torch.cuda.ExternalStream(stream_ptr, device=dev))
                self._views[k] = v
            _, ve, vs, vf, ext = v
            cur.wait_stream(ext)                 # device side: behind everything this engine has enqueued
            with torch.cuda.device(dev):
                buf[lo:hi, 0].copy_(ve, non_blocking=True)
                buf[lo:hi, 1].copy_(vs, non_blocking=True)
                if vf is not None:
                    buf[lo:hi, 2].copy_(vf.expand(hi - lo), non_blocking=True)     # int32 -> float64 inside the copy
            exts.append(ext)
        for ext in exts:
            ext.wait_stream(cur)                 # an engine's next run overwrites its result buffers only after these copies
        return buf

    def _local_scalars(self, want_energy_flags):
        """This rank's per-chain (E, sigma_E[, overflow flag]) ``[count, 2 or 3]``: device tensor or numpy array."""
        if self.result_path == "device":
            return self._device_scalars()
        res = self.engine.download(want_energy_flags)   # synchronises, repairs a capacity overflow
        return np.stack([np.asarray(res.get("energy_f64", res["energy"]), dtype=np.float64),
                         np.asarray(res.get("energy_std_f64", res["energy_std"]), dtype=np.float64)], axis=1)

    def _gather_local(self, scal):
        width = int(scal.shape[1])
        if self.result_path == "device":
            dev = scal.device
        else:
            dev = self.device if (self.world > 1 and self.dist.get_backend() == "nccl") else None
        g = self._gather
        if g is None or g.width != width or g.count != self.count or str(g.device) != str("cpu" if dev is None else dev):
            import torch

            g = self._gather = ChainGather(self.n_chains, width, self.dist, dev, torch.float64)
        return g(scal)

    def step(self, want, gather: bool | None = None):
        """One lock-step evaluation of the resident block (asynchronous on one GPU) and, when the chains are sharded, the
        path's only exchange: per-chain (E, sigma_E) to every rank.  Returns the gathered ``[n_chains, 2 or 3]`` tensor
        (device path: stream-ordered, not yet complete on return; one of two alternating buffers, so it stays valid while the
        NEXT step is gathered and is overwritten by the one after) or None when nothing was gathered.  Nothing is allocated
        per step (``ChainGather``)."""
        self.engine.run(want)
        if gather is None:
            gather = self.world > 1
        if not gather:
            return None
        from . import backend

        scal = self._local_scalars(backend.WANT_ENERGY | backend.WANT_STD)
        self.gathered = self._gather_local(scal)
        self._steps_since_check += 1
        return self.gathered

    def gather_only(self):
        """The exchange step alone, on the results of the last evaluation: this rank's per-chain (E, sigma_E[, overflow flag]) ->
        every rank.  What ``step()`` does after ``engine.run``; ``bench.rank_diagnostics`` times it separately."""
        from . import backend

        self.gathered = self._gather_local(self._local_scalars(backend.WANT_ENERGY | backend.WANT_STD))
        return self.gathered

    def check(self) -> bool:
        """Device result path: did any rank's LAST gathered evaluation overflow its neighbor capacity?  If so every rank
        repairs (``engine.synchronize()`` reruns with grown buffers) and gathers again; returns True when that happened.
        The host path repairs inside ``download`` and always returns False.

        The flag that is tested is the one of the most recent ``step()``: a caller that consumes the energies of EVERY step
        (MC acceptance) calls ``check()`` after every step, before the next one; ``bench.py`` evaluates the same resident
        positions K times and checks once after the last (an overflow would show in every step alike).
        ``steps_since_check`` says how many gathered steps the verdict covers."""
        g = self.gathered
        self.steps_since_check, self._steps_since_check = self._steps_since_check, 0
        if g is None or self.result_path != "device" or g.shape[1] < 3:
            return False
        if not bool((g[:, 2] != 0).any().item()):
            return False
        from . import backend

        self.engine.synchronize()
        scal = self._local_scalars(backend.WANT_ENERGY | backend.WANT_STD)
        self.gathered = self._gather_local(scal)
        return True

    # ---- one-shot form -------------------------------------------------------------------------------------
    def evaluate(self, all_chains: list) -> dict:
        """``all_chains``: the global list (every rank passes the same list; only its block is evaluated).
        Returns global per-chain ``energy`` / ``energy_std`` on every rank and this rank's ``forces``."""
        res = self.engine.evaluate(self.local_slice(all_chains))
        scal = np.stack([np.asarray(res.get("energy_f64", res["energy"]), dtype=np.float64),
                         np.asarray(res.get("energy_std_f64", res["energy_std"]), dtype=np.float64)], axis=1)
        full = gather_chain_scalars(scal, self.n_chains, self.dist, self.device)
        return {"energy": full[:, 0], "energy_std": full[:, 1], "forces_local": res["forces"],
                "cfg_start_local": res["cfg_start"], "first": self.first, "count": self.count}
