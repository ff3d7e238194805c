"""Host-driven optimizers over lock-step device evaluations.

The reference's ``optimize_slab`` (``mcmc/dynamics.py:119-141``) accepts ``optimizer="CG"`` -> ASE ``SciPyFminCG``, a wrapper around
``scipy.optimize.fmin_cg`` that owns its control flow (line searches call the energy / gradient callables whenever they like).  To run
such an optimizer for B chains at the price of one batched evaluation per round, every chain's optimizer runs in its own thread and
its energy / force requests meet at a rendezvous: when every live chain has posted a geometry, ONE ``set_positions`` + ``run`` +
``download`` of the resident batch serves them all (chains that have finished keep their last geometry).

``scipy_cg_batch`` restates ASE 3.22 ``ase/optimize/sciopt.py`` (``SciPyOptimizer`` / ``SciPyFminCG``) from its published behaviour --
ASE is not installable in the build container, so this path is parity-UNPINNED (checked by invariants only: monotone accepted
energies, FixAtoms honoured, agreement of the minimum with the device BFGS):
    f(x) = E(x) / alpha, f'(x) = -F(x).ravel() / alpha with alpha = 70 (ASE's default H0); forces of constrained atoms are zero;
    run(fmax, steps): convergence test on the start geometry, then fmin_cg(f, x0, fprime, gtol = 0.1 fmax / alpha, norm = inf,
    maxiter = steps, callback), the callback raising Converged once max_i |F_i| < fmax (it also drives the trajectory observer).
"""

import threading

import numpy as np

ALPHA = 70.0


class _Converged(Exception):
    pass


class LockstepEvaluator:
    """Rendezvous of B worker threads with one batched evaluation.  ``evaluate(pos_all) -> (energies [B], forces [N, 3])``."""

    def __init__(self, evaluate, cfg_start, positions):
        self._evaluate = evaluate
        self._cfg = np.asarray(cfg_start, dtype=np.int64)
        self._pos = np.array(positions, dtype=np.float64, copy=True)
        self._cv = threading.Condition()
        self._pending = set()
        self._live = set()
        self._round = 0
        self._results = {}
        self._error = None
        self.n_rounds = 0

    def request(self, b, pos):
        """Called by worker b: energy and forces of its chain at ``pos`` (blocks until the round is served)."""
        with self._cv:
            self._pos[self._cfg[b]:self._cfg[b + 1]] = pos
            self._pending.add(b)
            my_round = self._round
            self._cv.notify_all()
            while self._round == my_round and self._error is None:
                self._cv.wait()
            if self._error is not None:
                raise RuntimeError("batched evaluation failed") from self._error
            return self._results[b]

    def _worker(self, b, fn):
        try:
            fn(b)
        except BaseException as exc:   # surfaces in run()
            with self._cv:
                if self._error is None:
                    self._error = exc
                self._cv.notify_all()
        finally:
            with self._cv:
                self._live.discard(b)
                self._cv.notify_all()

    def run(self, fn):
        """Run ``fn(b)`` for every chain; returns when all have finished."""
        B = len(self._cfg) - 1
        self._live = set(range(B))
        threads = [threading.Thread(target=self._worker, args=(b, fn), daemon=True) for b in range(B)]
        for t in threads:
            t.start()
        with self._cv:
            while self._live and self._error is None:
                if self._pending and self._pending >= self._live:
                    try:
                        e, f = self._evaluate(self._pos)
                    except BaseException as exc:
                        self._error = exc
                        self._cv.notify_all()
                        break
                    self.n_rounds += 1
                    for b in self._pending:
                        self._results[b] = (float(e[b]), np.array(f[self._cfg[b]:self._cfg[b + 1]], dtype=np.float64, copy=True))
                    self._pending = set()
                    self._round += 1
                    self._cv.notify_all()
                else:
                    self._cv.wait()
        for t in threads:
            t.join()
        if self._error is not None:
            raise self._error
        return self._pos


def scipy_cg_batch(evaluate, cfg_start, positions, fixed=None, steps=20, fmax=0.01, record_interval=0):
    """ASE ``SciPyFminCG(atoms).run(fmax=fmax, steps=steps)`` for every chain of a resident batch.

    evaluate(pos_all [N, 3]) -> (energies [B], forces [N, 3]); fixed: uint8 mask [N] (FixAtoms) or None.
    Returns dict(positions [N, 3], energy [B], forces [N, 3] (constraints applied), n_steps [B], converged [B],
    traj (per chain list of (positions, energy, forces) every ``record_interval`` optimizer steps, or None), rounds)."""
    from scipy import optimize as opt

    cfg = np.asarray(cfg_start, dtype=np.int64)
    B = len(cfg) - 1
    free = np.ones(int(cfg[-1]), bool) if fixed is None else ~np.asarray(fixed, dtype=bool)
    ev = LockstepEvaluator(evaluate, cfg, positions)
    n_steps = np.zeros(B, np.int32)
    converged = np.zeros(B, bool)
    final_e = np.zeros(B)
    final_f = np.zeros((int(cfg[-1]), 3))
    traj = [[] for _ in range(B)] if record_interval else None

    def chain(b):
        a0, a1 = int(cfg[b]), int(cfg[b + 1])
        mask = free[a0:a1, None]
        cache = {}

        def ef(x):   # ASE's calculator cache: f(x) and fprime(x) at the same point cost one evaluation
            key = x.tobytes()
            if cache.get("key") != key:
                pos = np.where(mask, x.reshape(-1, 3), np.asarray(positions[a0:a1], dtype=np.float64))   # FixAtoms.adjust_positions
                e, f = ev.request(b, pos)
                cache.update(key=key, e=e, f=np.where(mask, f, 0.0), pos=pos)                            # adjust_forces
            return cache["e"], cache["f"]

        calls = [0]

        def callback(x):
            e, f = ef(x)
            if traj is not None and calls[0] % int(record_interval) == 0:
                traj[b].append((cache["pos"].copy(), e, f.copy()))
            calls[0] += 1
            if (f ** 2).sum(axis=1).max() < fmax ** 2:
                raise _Converged
            n_steps[b] += 1

        x0 = np.asarray(positions[a0:a1], dtype=np.float64).reshape(-1)
        x_last = x0
        try:
            callback(x0)
            out = opt.fmin_cg(lambda x: ef(x)[0] / ALPHA, x0, fprime=lambda x: -ef(x)[1].reshape(-1) / ALPHA,
                              gtol=fmax / ALPHA * 0.1, norm=np.inf, maxiter=int(steps), full_output=1, disp=0, callback=callback)
            x_last = np.asarray(out[0], dtype=np.float64)
        except _Converged:
            converged[b] = True
            x_last = None
        if x_last is not None:   # left through maxiter / scipy's own tests: the reference reads the energy at the final geometry
            ef(x_last)
        final_e[b], final_f[a0:a1] = cache["e"], cache["f"]
        ev._pos[a0:a1] = cache["pos"]

    pos = ev.run(chain)
    return {"positions": pos, "energy": final_e, "forces": final_f, "n_steps": n_steps, "converged": converged, "traj": traj,
            "rounds": ev.n_rounds}


def optimizer_class_batch(optimizer_cls, atoms_list, make_calculator, evaluate, cfg_start, fixed_indices=None, steps=20, fmax=0.01,
                          record_interval=0, optimizer_kwargs=None):
    """Any optimizer that follows the ASE protocol -- ``dyn = optimizer_cls(atoms, **kw); dyn.attach(fn, interval=k);
    dyn.run(fmax=..., steps=...)`` talking to ``atoms.get_forces() / get_potential_energy() / get_positions() / set_positions()`` --
    for every chain of a resident batch, one thread per chain, lock-step evaluations (``LockstepEvaluator``).  This is how
    ``relax_batch`` serves ``optimizer="BFGSLineSearch"`` (reference ``mcmc/dynamics.py:119-120``) where ASE is installed, and any
    optimizer class passed directly.

    make_calculator(request) -> a calculator whose energy / forces come from ``request(atoms) -> (energy, forces)`` (the caller's
    Calculator base class: ``calculators._LockstepProxy``).  fixed_indices: per chain, indices held fixed IN ADDITION to whatever
    constraints the atoms objects carry themselves (ase.Atoms apply their own FixAtoms; plain Structures have none).
    Returns dict(positions, atoms (the optimised copies), optimizers, traj (reference layout per chain or None), rounds)."""
    cfg = np.asarray(cfg_start, dtype=np.int64)
    B = len(cfg) - 1
    start = np.concatenate([np.asarray(a.get_positions(), dtype=np.float64).reshape(-1, 3) for a in atoms_list])
    ev = LockstepEvaluator(evaluate, cfg, start)
    work = [None] * B
    dyns = [None] * B
    traj = [None] * B

    def chain(b):
        a0, a1 = int(cfg[b]), int(cfg[b + 1])
        fixed = np.zeros(a1 - a0, bool)
        if fixed_indices is not None and fixed_indices[b] is not None and len(fixed_indices[b]):
            fixed[np.asarray(fixed_indices[b], dtype=np.int64)] = True
        atoms = atoms_list[b].copy()

        def request(at):
            pos = np.where(fixed[:, None], start[a0:a1], np.asarray(at.get_positions(), dtype=np.float64).reshape(-1, 3))
            e, f = ev.request(b, pos)
            return e, np.where(fixed[:, None], 0.0, f)

        atoms.calc = make_calculator(request)
        dyn = optimizer_cls(atoms, **(optimizer_kwargs or {}))
        if record_interval:
            rec = {"atoms": [], "energies": [], "forces": []}

            def observe():   # the reference's TrajectoryObserver (mcmc/dynamics.py:20-80)
                frame = atoms.copy()
                if hasattr(frame, "calc"):
                    frame.calc = None
                rec["atoms"].append(frame)
                rec["energies"].append(float(atoms.get_potential_energy()))
                rec["forces"].append(np.array(atoms.get_forces(), copy=True))

            dyn.attach(observe, interval=int(record_interval))
            traj[b] = rec
        dyns[b] = dyn
        dyn.run(fmax=fmax, steps=int(steps))
        pos = np.where(fixed[:, None], start[a0:a1], np.asarray(atoms.get_positions(), dtype=np.float64).reshape(-1, 3))
        ev._pos[a0:a1] = pos
        atoms.set_positions(pos)
        work[b] = atoms

    pos = ev.run(chain)
    return {"positions": pos, "atoms": work, "optimizers": dyns, "traj": traj if record_interval else None, "rounds": ev.n_rounds}
