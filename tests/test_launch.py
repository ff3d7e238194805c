"""The launcher that runs a reference script unchanged (surface_sampling_amd.launch): with a stand-in ``mcmc`` package and a
script that imports its calculator the way ``scripts/sample_surface.py:19`` does, the classes the script receives are this
backend's, in the package namespace and in ``mcmc.calculators.calculators`` (the real reference environment -- ase, nff,
catkit -- is not installable here)."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT


def test_launcher_patches_the_reference_namespace_and_runs_the_script_unchanged(tmp_path):
    pkg = tmp_path / "mcmc" / "calculators"
    pkg.mkdir(parents=True)
    (tmp_path / "mcmc" / "__init__.py").write_text("")
    (pkg / "calculators.py").write_text(textwrap.dedent("""
        class EnsembleNFFSurface:          # stands for the reference class (torch / nff behind it)
            origin = "reference"
        class CHGNetSurfCalc:
            origin = "reference"
        def get_std_devs_single(a, c):
            return "reference"
    """))
    (pkg / "__init__.py").write_text("from .calculators import EnsembleNFFSurface, CHGNetSurfCalc, get_std_devs_single\n")
    script = tmp_path / "sample_surface.py"
    script.write_text(textwrap.dedent("""
        import sys
        from mcmc.calculators import EnsembleNFFSurface, CHGNetSurfCalc, get_std_devs_single
        import mcmc.calculators.calculators as inner
        print("argv", sys.argv[1:])
        print("class", EnsembleNFFSurface.__module__, inner.EnsembleNFFSurface.__module__)
        print("kept", CHGNetSurfCalc.origin, get_std_devs_single.__module__)
        if __name__ == "__main__":
            print("main-ok")
    """))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, str(tmp_path)]))
    r = subprocess.run([sys.executable, "-m", "surface_sampling_amd.launch", str(script), "--run_name", "x", "--device", "cuda"],
                       env=env, capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert "argv ['--run_name', 'x', '--device', 'cuda']" in out and "main-ok" in out
    assert "class surface_sampling_amd.calculators surface_sampling_amd.calculators" in out
    assert "kept reference surface_sampling_amd.calculators" in out      # classes without a replacement stay the reference's
    r2 = subprocess.run([sys.executable, "-m", "surface_sampling_amd.launch", "--vssr-keep", "EnsembleNFFSurface", str(script)],
                        env=env, capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
    assert r2.returncode == 0 and "class mcmc.calculators.calculators mcmc.calculators.calculators" in r2.stdout, r2.stdout + r2.stderr
