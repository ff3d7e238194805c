"""The CPU oracle under AddressSanitizer + UBSan: `make -C oracle asan`, then the known-answer checks of
the three restatements (PaiNN ensemble with forces, neighbor list, Tersoff) in a child interpreter that preloads the
sanitizer runtimes.  Any out-of-bounds access, use-after-free or undefined shift / overflow inside the oracle aborts the
child.  (GPU sanitizers are not available on the pool; the device code is covered by the parity tests instead.)"""
import os
import subprocess
import sys

from conftest import ROOT

CHILD = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from conftest import Golden
assert oracle._LIB_PATH.endswith("libvssr_oracle_asan.so")
oracle.set_threads(4)
g = Golden()
table, const = g.offset_table()
case = g.kat["painn_ensemble"][0]
s = g.structure(case["structure"])
r = oracle.ensemble(g.blobs, s.numbers, s.positions, s.cell, s.pbc, 32, table, const)
assert abs(r["energy"] - case["energy"]) <= 2e-4, r["energy"]
r64 = oracle.ensemble(g.blobs, s.numbers, s.positions, s.cell, s.pbc, 64, table, const)
assert abs(r64["energy"] - case["energy"]) <= 2e-4 and np.isfinite(r64["forces"]).all()
E, G, d = oracle.painn(g.blobs[1], s.numbers, s.positions, s.cell, s.pbc, 64, dump=True)
assert len(d["s_upd"]) >= 3 and np.isfinite(d["s_upd"][2]).all()
ei, ej, eS, er = oracle.neighbors(s.positions, s.cell, s.pbc, 5.0)
assert len(ei) == int(g.fine["S60.n_edges"])
t = g.structure("GaN_3x3_pristine")
types = np.array([0 if z == 31 else 1 for z in t.numbers], np.int32)
Et, ea, F = oracle.tersoff(g.tersoff_params, types, t.positions, t.cell, [1, 1, 1])
assert abs(Et - g.kat["tersoff"]["energy"]) <= 1e-3
# degenerate inputs: one atom, no neighbors; a non-periodic pair
one = oracle.ensemble(g.blobs, s.numbers[:1], s.positions[:1], s.cell, np.zeros(3, bool), 64, table, const)
assert np.isfinite(one["energy"])
print("asan-ok")
'''


def test_oracle_known_answers_under_address_sanitizer():
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", odir, "-s", "asan"])
    lib = os.path.join(odir, "libvssr_oracle_asan.so")
    preload = [subprocess.check_output(["gcc", f"-print-file-name={n}"], text=True).strip() for n in ("libasan.so", "libubsan.so")]
    env = dict(os.environ, VSSR_ORACLE_LIB=lib, LD_PRELOAD=":".join(preload),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "asan-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
