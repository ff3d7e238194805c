"""Minimal stand-in for the ``ase`` package (TEST INFRASTRUCTURE, tests/fake_ase is put on sys.path by
tests/test_ase_branch.py only).  ASE is not installable in the build container, so the branch of
surface_sampling_amd.calculators that derives from ``ase.calculators.calculator.Calculator`` would never execute; this
package reproduces the real base class's constructor / set / get_property / check_state / calculate protocol (ASE 3.22
semantics, written from its documented behaviour) so that the branch runs in the CPU suite."""
__version__ = "0.0-standin"
