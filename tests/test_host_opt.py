"""Host-driven optimizers over lock-step evaluations (surface_sampling_amd/host_opt.py): the rendezvous and the ASE-SciPyFminCG
restatement, on the CPU with an analytic stand-in for the engine; on the GPU through EnsembleNFFSurface.relax_batch(optimizer="CG")."""
import numpy as np
import pytest

from surface_sampling_amd import host_opt


def _wells(n_atoms, seed):
    """Anisotropic harmonic wells + a quartic term, one per chain: E = sum_i k_i |x_i - c_i|^2 / 2 + q |x_i - c_i|^4."""
    rng = np.random.default_rng(seed)
    cfg = np.concatenate([[0], np.cumsum(n_atoms)])
    centres = rng.normal(0, 1, (cfg[-1], 3))
    k = rng.uniform(0.5, 4.0, (cfg[-1], 3))
    calls = []

    def evaluate(pos):
        calls.append(pos.copy())
        d = pos - centres
        e_atom = 0.5 * (k * d * d).sum(axis=1) + 0.1 * (d ** 4).sum(axis=1)
        f = -(k * d + 0.4 * d ** 3)
        return np.array([e_atom[cfg[b]:cfg[b + 1]].sum() for b in range(len(n_atoms))]), f

    return cfg, centres, evaluate, calls


def test_lockstep_cg_reaches_every_minimum_with_one_evaluation_per_round():
    n_atoms = [5, 9, 3, 7]
    cfg, centres, evaluate, calls = _wells(n_atoms, 3)
    start = centres + np.random.default_rng(4).normal(0, 0.4, centres.shape)
    out = host_opt.scipy_cg_batch(evaluate, cfg, start, steps=200, fmax=1e-3)
    assert out["converged"].all()
    assert np.abs(out["positions"] - centres).max() < 2e-3
    assert np.abs(out["forces"]).max() < 1e-3 * 1.0001
    assert len(calls) == out["rounds"]                       # one batched evaluation per rendezvous, nothing else
    assert out["rounds"] < 4 * out["n_steps"].max() + 8      # chains advance together, not one after the other
    # the same chains one at a time take the same path: a chain's optimizer only ever sees its own energies / forces
    for b in range(len(n_atoms)):
        sl = slice(cfg[b], cfg[b + 1])
        cfg1, _, ev1, _ = _wells(n_atoms, 3)

        def one(pos_b, ev1=ev1, sl=sl):
            full = start.copy()
            full[sl] = pos_b
            e, f = ev1(full)
            return np.array([e[b]]), f[sl]

        solo = host_opt.scipy_cg_batch(one, np.array([0, n_atoms[b]]), start[sl], steps=200, fmax=1e-3)
        assert np.array_equal(solo["positions"], out["positions"][sl]) and solo["n_steps"][0] == out["n_steps"][b]


def test_fixed_atoms_step_limit_and_trajectory():
    n_atoms = [6, 4]
    cfg, centres, evaluate, _ = _wells(n_atoms, 8)
    start = centres + 0.5
    fixed = np.zeros(10, np.uint8)
    fixed[[0, 1, 7]] = 1
    out = host_opt.scipy_cg_batch(evaluate, cfg, start, fixed=fixed, steps=3, fmax=1e-8, record_interval=1)
    assert np.array_equal(out["positions"][fixed.astype(bool)], start[fixed.astype(bool)])   # FixAtoms: never moved
    assert np.all(out["forces"][fixed.astype(bool)] == 0.0)
    assert not out["converged"].any() and (out["n_steps"] <= 4).all()
    for b in range(2):
        energies = [e for _, e, _ in out["traj"][b]]
        assert len(energies) >= 2 and all(e1 <= e0 + 1e-12 for e0, e1 in zip(energies, energies[1:]))   # accepted points only
        assert np.array_equal(out["traj"][b][0][0], start[cfg[b]:cfg[b + 1]])
    # an already converged start: no optimizer step, no scipy call
    out0 = host_opt.scipy_cg_batch(evaluate, cfg, centres, steps=5, fmax=1e-3)
    assert out0["converged"].all() and (out0["n_steps"] == 0).all() and out0["rounds"] == 1


def test_an_exception_in_the_evaluation_reaches_the_caller():
    cfg = np.array([0, 2, 4])

    def broken(pos):
        raise ValueError("device lost")

    with pytest.raises(ValueError, match="device lost"):
        host_opt.scipy_cg_batch(broken, cfg, np.zeros((4, 3)), steps=3)


@pytest.mark.gpu
def test_cg_optimizer_through_relax_batch_on_the_gpu(golden):
    """optimizer="CG" (reference optimize_slab -> SciPyFminCG): energies go down, FixAtoms are honoured, the result agrees with what
    the device BFGS finds from the same start, and the returned energy is the engine's energy at the returned geometry."""
    from surface_sampling_amd import structures
    from surface_sampling_amd.calculators import EnsembleNFFSurface

    base = golden.structure("SrTiO3_2x2_pristine")
    slabs = [structures.synth_chain(base, c, grid=(4, 4)) for c in range(4)]
    fixed = [np.flatnonzero(s.positions[:, 2] < base.positions[:, 2].max() - 4.0) for s in slabs]
    calc = EnsembleNFFSurface(golden.blobs, device="cuda:0", model_units="kcal/mol", prediction_units="eV", offset_units="atomic")
    calc.set(offset=True, offset_data=golden.offset_data)
    e0 = [float(r["energy"][0]) for r in calc.calculate_batch(slabs)]
    cg = calc.relax_batch(slabs, fixed_indices=fixed, relax_steps=12, fmax=0.05, optimizer="CG", save_traj=True, record_interval=3)
    bfgs = calc.relax_batch(slabs, fixed_indices=fixed, relax_steps=12, fmax=0.05, optimizer="BFGS")
    for b, (slab, (relaxed, traj, energy, oob, res)) in enumerate(zip(slabs, cg)):
        assert not oob and energy < e0[b] - 0.1
        assert np.array_equal(relaxed.positions[fixed[b]], slab.positions[fixed[b]])
        again = calc.calculate_batch([relaxed])[0]
        assert abs(float(again["energy"][0]) - energy) <= 2e-4
        assert traj is not None and len(traj["atoms"]) == len(traj["energies"]) >= 2
        assert traj["energies"][-1] <= traj["energies"][0]
        assert abs(energy - bfgs[b][2]) < 0.15 * abs(e0[b] - bfgs[b][2]) + 0.3   # comparable progress in 12 steps (60 eV downhill from these starts)


class _Descent:
    """Stand-in for an ASE optimizer class (ASE is not installable here): the protocol relax_batch relies on -- constructed with the
    atoms, observers attached with an interval, run(fmax, steps) driving atoms.get_forces() / set_positions()."""

    def __init__(self, atoms, step=0.05):
        self.atoms, self.step, self.nsteps, self.observers = atoms, step, 0, []

    def attach(self, fn, interval=1):
        self.observers.append((fn, interval))

    def run(self, fmax=0.05, steps=100):
        while True:
            f = self.atoms.get_forces()
            for fn, k in self.observers:
                if self.nsteps % k == 0:
                    fn()
            if (f ** 2).sum(axis=1).max() < fmax ** 2 or self.nsteps >= steps:
                return
            self.atoms.set_positions(self.atoms.get_positions() + self.step * f)
            self.nsteps += 1


def test_any_ase_protocol_optimizer_runs_in_lockstep():
    from surface_sampling_amd import structures
    from surface_sampling_amd.calculators import _LockstepProxy

    n_atoms = [4, 6, 5]
    cfg, centres, evaluate, calls = _wells(n_atoms, 21)
    start = centres + 0.3
    atoms = [structures.Structure(np.full(n, 8), start[cfg[b]:cfg[b + 1]], np.eye(3) * 20) for b, n in enumerate(n_atoms)]
    fixed = [np.array([0]), None, np.array([1, 2])]
    out = host_opt.optimizer_class_batch(_Descent, atoms, _LockstepProxy, evaluate, cfg, fixed_indices=fixed, steps=400, fmax=1e-3,
                                         record_interval=50, optimizer_kwargs={"step": 0.1})
    free = np.ones(len(start), bool)
    free[[0, cfg[2] + 1, cfg[2] + 2]] = False
    assert np.abs(out["positions"][free] - centres[free]).max() < 2e-3
    assert np.array_equal(out["positions"][~free], start[~free])
    assert len(calls) == out["rounds"] <= max(d.nsteps for d in out["optimizers"]) + 2   # lock-step: rounds = the longest chain's steps
    for b in range(3):
        rec = out["traj"][b]
        assert len(rec["atoms"]) == len(rec["energies"]) == len(rec["forces"]) >= 2
        assert all(e1 <= e0 for e0, e1 in zip(rec["energies"], rec["energies"][1:]))
        assert np.array_equal(rec["atoms"][0].positions, start[cfg[b]:cfg[b + 1]])
        assert np.array_equal(atoms[b].positions, start[cfg[b]:cfg[b + 1]])      # the caller's objects are untouched


@pytest.mark.gpu
def test_optimizer_class_through_relax_batch_on_the_gpu(golden):
    from surface_sampling_amd import structures
    from surface_sampling_amd.calculators import EnsembleNFFSurface

    base = golden.structure("SrTiO3_2x2_pristine")
    slabs = [structures.synth_chain(base, c, grid=(4, 4)) for c in range(3)]
    fixed = [np.flatnonzero(s.positions[:, 2] < base.positions[:, 2].max() - 4.0) for s in slabs]
    calc = EnsembleNFFSurface(golden.blobs, device="cuda:0", model_units="kcal/mol", prediction_units="eV", offset_units="atomic")
    calc.set(offset=True, offset_data=golden.offset_data)
    e0 = [float(r["energy"][0]) for r in calc.calculate_batch(slabs)]
    res = calc.relax_batch(slabs, fixed_indices=fixed, relax_steps=10, fmax=0.05,
                           optimizer=lambda atoms: _Descent(atoms, step=5e-4), save_traj=True, record_interval=5)
    for b, (relaxed, traj, energy, oob, r) in enumerate(res):
        assert not oob and energy < e0[b] - 0.05 and r["n_steps"] == 10, (energy, e0[b], r["n_steps"])
        assert np.array_equal(relaxed.positions[fixed[b]], slabs[b].positions[fixed[b]])
        assert abs(float(calc.calculate_batch([relaxed])[0]["energy"][0]) - energy) <= 2e-4
        assert len(traj["atoms"]) == 3 and traj["energies"][0] == pytest.approx(e0[b], abs=2e-4)
    with pytest.raises(Exception, match="BFGSLineSearch"):
        calc.relax_batch(slabs, optimizer="BFGSLineSearch")
