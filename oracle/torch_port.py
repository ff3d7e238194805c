"""torch-CPU restatement of the reference's PaiNN ensemble evaluation (energy + autograd forces).

TEST INFRASTRUCTURE ONLY — the second CPU baseline of bench.py ("the build's C++ CPU oracle AND a
torch-CPU restatement") and a cross-check of the C oracle.  This is how the reference itself computes the path
(``EnsembleNFF.calculate``, reference call site ``mcmc/calculators/calculators.py:484``: per model a torch forward, then
``torch.autograd.grad``), written from SURVEY.md Appendix A items 2-10 because nff is not installable here; dense layers run
on torch's CPU BLAS; the neighbor sums (nff: ``torch_scatter.scatter_add``) run as dense reductions over edge rows padded to
the largest degree (``index_add_`` on [E, 3, F] tensors is single-threaded on CPU and was 6x slower).  Parity status: pinned
through the C oracle (tests/test_oracle_kat.py::test_torch_port_matches_oracle), which is pinned by the reference's
known answers.  The neighbor multigraph comes from the C oracle (``oracle.neighbors``).
"""

from __future__ import annotations

import numpy as np
import torch

import oracle

F, R, L, H = 128, 20, 3, 64
CUTOFF, SIGMA, POWER = 5.0, 1.5, 12


def _fields(blob):
    """Canonical blob layout (include/vssr_eval.h): embedding, per layer message + update weights, readout."""
    shapes = [("embed", (100, F))]
    for l in range(L):
        shapes += [(f"msg{l}.W1", (F, F)), (f"msg{l}.b1", (F,)), (f"msg{l}.W2", (3 * F, F)), (f"msg{l}.b2", (3 * F,)),
                   (f"msg{l}.Wd", (3 * F, R)), (f"msg{l}.bd", (3 * F,)), (f"upd{l}.U", (F, F)), (f"upd{l}.V", (F, F)),
                   (f"upd{l}.W3", (F, 2 * F)), (f"upd{l}.b3", (F,)), (f"upd{l}.W4", (3 * F, F)), (f"upd{l}.b4", (3 * F,))]
    shapes += [("readout.W5", (H, F)), ("readout.b5", (H,)), ("readout.w6", (1, H)), ("readout.b6", (1,))]
    out, off = {}, 0
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    for name, shp in shapes:
        n = int(np.prod(shp))
        out[name] = torch.from_numpy(blob[off:off + n].reshape(shp).copy())
        off += n
    assert off == blob.size, "blob does not match the canonical layout"
    return out


def swish(x):
    return x * torch.sigmoid(x)


class TorchEnsemble:
    def __init__(self, blobs, dtype=torch.float32):
        self.dtype = dtype
        self.models = [{k: v.to(dtype) for k, v in _fields(b).items()} for b in blobs]

    def model_energy(self, W, Z, ej, r, mask, N, D, per_atom=False):
        """Energy (model units) of one structure.  Edge rows are padded per centre to D slots: ``r`` [N D, 3] (requires
        grad), ``ej`` [N D] neighbor of every slot, ``mask`` [N D] 1 for real edges."""
        d = torch.sqrt((r * r).sum(dim=1))
        u = r / d[:, None]
        n = torch.arange(1, R + 1, dtype=self.dtype)
        rbf = torch.sin(n[None, :] * (np.pi / CUTOFF) * d[:, None]) / d[:, None]
        fcut = torch.where(d < CUTOFF, 0.5 * (torch.cos(np.pi * d / CUTOFF) + 1.0), torch.zeros_like(d)) * mask
        s = W["embed"][Z]
        v = torch.zeros(N, 3, F, dtype=self.dtype)
        for l in range(L):
            phi = swish(s @ W[f"msg{l}.W1"].T + W[f"msg{l}.b1"]) @ W[f"msg{l}.W2"].T + W[f"msg{l}.b2"]
            w = (rbf @ W[f"msg{l}.Wd"].T + W[f"msg{l}.bd"]) * fcut[:, None]
            a, b, c = torch.split(phi[ej] * w, F, dim=1)
            s = s + b.reshape(N, D, F).sum(dim=1)
            dv = c[:, None, :] * u[:, :, None] + a[:, None, :] * v[ej]
            v = v + dv.reshape(N, D, 3, F).sum(dim=1)
            Uv, Vv = v @ W[f"upd{l}.U"].T, v @ W[f"upd{l}.V"].T
            nrm = torch.sqrt((Vv * Vv + 1e-15).sum(dim=1))
            g = swish(torch.cat([s, nrm], dim=1) @ W[f"upd{l}.W3"].T + W[f"upd{l}.b3"]) @ W[f"upd{l}.W4"].T + W[f"upd{l}.b4"]
            a_vv, a_sv, a_ss = torch.split(g, F, dim=1)
            v = v + a_vv[:, None, :] * Uv
            s = s + a_sv * (Uv * Vv).sum(dim=1) + a_ss
        e_atom = (swish(s @ W["readout.W5"].T + W["readout.b5"]) @ W["readout.w6"].T + W["readout.b6"])[:, 0]
        if per_atom:   # (evaluate_batch: per-atom energies incl. the excluded volume of the atom's own slots)
            return e_atom + (((SIGMA / d) ** POWER) * mask).reshape(N, D).sum(dim=1)
        return e_atom.sum() + (((SIGMA / d) ** POWER) * mask).sum()

    def evaluate(self, Z, pos, cell, pbc, offset_per_z=None, offset_const=0.0, units_per_ev=oracle.EV_TO_KCAL_MOL):
        """dict(energy, energy_std, forces, forces_std, energy_models) in eV, like ``oracle.ensemble``."""
        ei, ej, eS, er = oracle.neighbors(pos, cell, pbc, CUTOFF)   # sorted by centre
        N = len(Z)
        deg = np.bincount(ei, minlength=N)
        D = int(max(deg.max(), 1))
        start = np.concatenate([[0], np.cumsum(deg)])[:-1]
        slot = ei.astype(np.int64) * D + (np.arange(len(ei)) - start[ei])      # padded row of every real edge
        r_pad = np.zeros((N * D, 3))
        r_pad[:, 0] = 1.0                                                       # dummy slots: unit length, masked out
        r_pad[slot] = er
        ej_pad = np.zeros(N * D, np.int64)
        ej_pad[slot] = ej
        ei_pad = np.repeat(np.arange(N, dtype=np.int64), D)
        m = np.zeros(N * D)
        m[slot] = 1.0
        Zt = torch.from_numpy(np.asarray(Z, dtype=np.int64))
        ej_t, ei_t, mask = torch.from_numpy(ej_pad), torch.from_numpy(ei_pad), torch.from_numpy(m).to(self.dtype)
        es, fs = [], []
        for W in self.models:
            r = torch.from_numpy(r_pad).to(self.dtype).requires_grad_(True)
            e = self.model_energy(W, Zt, ej_t, r, mask, N, D)
            (g,) = torch.autograd.grad(e, r)
            g = g * mask[:, None]
            # r_e = x_j - x_i (+ image shift): dE/dx_j += g_e, dE/dx_i -= g_e
            grad = torch.zeros(N, 3, dtype=self.dtype).index_add_(0, ej_t, g).index_add_(0, ei_t, -g)
            es.append(float(e.detach()) / units_per_ev)
            fs.append(-(grad.double().numpy()) / units_per_ev)
        off = 0.0
        if offset_per_z is not None:
            off = float(np.asarray(offset_per_z)[np.asarray(Z)].sum()) + offset_const
        es = np.array(es) + off
        fs = np.stack(fs)
        return {"energy": es.mean(), "energy_std": es.std(), "forces": fs.mean(0), "forces_std": fs.std(0),
                "energy_models": es}

    def evaluate_batch(self, structs, offset_per_z=None, offset_const=0.0, units_per_ev=oracle.EV_TO_KCAL_MOL):
        """Several independent structures ``(Z, pos, cell, pbc)`` as ONE graph (atoms concatenated, neighbor indices shifted): the
        dense layers see all atoms of the batch in one GEMM, which is how a CPU runs this model at its best (bench.py's
        ``cpu_baseline`` times it).  Returns per-structure ``energy`` / ``energy_std`` [B] and ``forces`` of the concatenated
        atoms, like ``evaluate`` does for one structure."""
        parts, n_at = [], []
        for Z, pos, cell, pbc in structs:
            parts.append(oracle.neighbors(pos, cell, pbc, CUTOFF))
            n_at.append(len(Z))
        first = np.concatenate([[0], np.cumsum(n_at)])
        N = int(first[-1])
        D = int(max(max(np.bincount(p[0], minlength=n).max() for p, n in zip(parts, n_at)), 1))
        r_pad = np.zeros((N * D, 3)); r_pad[:, 0] = 1.0
        ej_pad = np.zeros(N * D, np.int64); m = np.zeros(N * D)
        for (ei, ej, eS, er), n, a0 in zip(parts, n_at, first[:-1]):
            deg = np.bincount(ei, minlength=n)
            start = np.concatenate([[0], np.cumsum(deg)])[:-1]
            slot = (ei.astype(np.int64) + a0) * D + (np.arange(len(ei)) - start[ei])
            r_pad[slot] = er; ej_pad[slot] = ej.astype(np.int64) + a0; m[slot] = 1.0
        ei_pad = np.repeat(np.arange(N, dtype=np.int64), D)
        Zall = np.concatenate([np.asarray(s[0], dtype=np.int64) for s in structs])
        Zt, ej_t, ei_t = torch.from_numpy(Zall), torch.from_numpy(ej_pad), torch.from_numpy(ei_pad)
        mask = torch.from_numpy(m).to(self.dtype)
        chain_of = torch.from_numpy(np.repeat(np.arange(len(structs)), n_at))
        es, fs = [], []
        for W in self.models:
            r = torch.from_numpy(r_pad).to(self.dtype).requires_grad_(True)
            e_atoms = self.model_energy(W, Zt, ej_t, r, mask, N, D, per_atom=True)
            (g,) = torch.autograd.grad(e_atoms.sum(), r)
            g = g * mask[:, None]
            grad = torch.zeros(N, 3, dtype=self.dtype).index_add_(0, ej_t, g).index_add_(0, ei_t, -g)
            e_chain = torch.zeros(len(structs), dtype=torch.float64).index_add_(0, chain_of, e_atoms.detach().double())
            es.append(e_chain.numpy() / units_per_ev)
            fs.append(-(grad.double().numpy()) / units_per_ev)
        es = np.stack(es, axis=1)     # [B, M]
        if offset_per_z is not None:
            table = np.asarray(offset_per_z)
            es = es + np.array([table[np.asarray(s[0])].sum() + offset_const for s in structs])[:, None]
        fs = np.stack(fs)
        return {"energy": es.mean(1), "energy_std": es.std(1), "forces": fs.mean(0), "forces_std": fs.std(0), "energy_models": es}
