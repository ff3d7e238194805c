"""Hostile inputs to the two parsers of untrusted files: the C parser behind
``vssr_tersoff_create_from_text`` and the torch-zip checkpoint reader.  Truncated, duplicated, bit-flipped and oversized
inputs must end in an error code / a clean Python exception -- never a crash, a hang or a giant allocation.  The C parser
runs in a child process (a crash must not take pytest down); on a machine without a GPU a well-formed file ends in
VSSR_E_DEVICE *after* parsing, which is all this test needs."""
import ctypes as C
import io
import os
import subprocess
import sys
import zipfile

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

CHILD = r'''
import ctypes as C, json, os, random, sys
sys.path.insert(0, ROOT)
from surface_sampling_amd import backend
L = backend.load_library()
fn = L.vssr_tersoff_create_from_text
params = json.load(open(os.path.join(ROOT, "tests", "golden", "GaN_tersoff_params.json")))
sp = params["species"]
lines = []
for i, a in enumerate(sp):
    for j, b in enumerate(sp):
        for k, c in enumerate(sp):
            lines.append(" ".join([a, b, c] + [repr(float(x)) for x in params["params_ijk"][i][j][k]]))
good = ("# GaN\n" + "\n".join(lines) + "\n").encode()
arr = (C.c_char_p * 2)(b"Ga", b"N")
def call(text, n=2, species=arr):
    h = C.c_void_p(None)
    rc = fn(0, text, n, species, C.byref(h))
    if rc == 0:
        L.vssr_destroy(h)
    return rc
rc = call(good)
assert rc in (0, -2), rc            # parsed; -2 = no HIP device here
rng = random.Random(7)
seen = set()
for it in range(3000):
    b = bytearray(good)
    op = rng.randrange(8)
    if op == 0: b = b[:rng.randrange(len(b))]
    elif op == 1:
        for _ in range(rng.randrange(1, 8)): b[rng.randrange(len(b))] = rng.randrange(1, 256)
    elif op == 2:
        i = rng.randrange(len(b)); b[i:i] = bytes(rng.randrange(1, 256) for _ in range(rng.randrange(1, 64)))
    elif op == 3: b = b * rng.randrange(2, 5)
    elif op == 4:
        i = rng.randrange(len(b)); b[i:i] = b"9" * rng.randrange(100, 5000)
    elif op == 5:
        i = rng.randrange(len(b)); b[i:i] = rng.choice([b" nan ", b" inf ", b" -inf ", b" 1e999 ", b" 0x1p3 ", b" - ", b"#", b"\n\n\n", b"\t"])
    elif op == 6:
        toks = bytes(b).split(); rng.shuffle(toks); b = bytearray(b" ".join(toks))
    else:
        i, j = sorted(rng.randrange(len(b)) for _ in range(2)); del b[i:j]
    text = bytes(b).replace(b"\0", b" ")
    rc = call(text)
    seen.add(rc)
    assert rc in (0, -1, -2), (it, op, rc)
# argument abuse
assert call(good, 0) == -1 and call(good, 9) == -1 and call(None) == -1
bad = (C.c_char_p * 2)(b"Ga", None)
assert call(good, 2, bad) == -1
huge = good + b"Ga Ga Ga " + b" ".join([b"1.0"] * 14) + b"\n"
assert call(huge * 2000) in (0, -2)                       # 6 MB of repeated entries: later entries override, no growth
print("fuzz-ok", sorted(seen))
'''


def test_tersoff_text_parser_survives_hostile_input():
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fuzz-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert "-1" in r.stdout                                # the mutations did reach the error paths


def _checkpoint_bytes():
    """``torch.save`` of a whole module whose classes live under ``nff.*`` -- the format of the reference's ``best_model``
    files -- carrying the first golden model: the fuzz base."""
    import types

    import torch

    from surface_sampling_amd import checkpoint

    blob = np.fromfile(os.path.join(GOLDEN, "weights", "SrTiO3_painn_model01.f32"), dtype="<f4")
    names = ("nff", "nff.nn", "nff.nn.models", "nff.nn.models.painn", "nff.nn.modules", "nff.nn.modules.painn")
    mods = {n: types.ModuleType(n) for n in names}

    class Painn(torch.nn.Module):
        pass

    class Block(torch.nn.Module):
        pass

    Painn.__module__, Painn.__qualname__ = "nff.nn.models.painn", "Painn"
    Block.__module__, Block.__qualname__ = "nff.nn.modules.painn", "Block"
    mods["nff.nn.models.painn"].Painn = Painn
    mods["nff.nn.modules.painn"].Block = Block
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        fields = checkpoint.blob_to_fields(blob)
        top = Painn()
        for field, key in checkpoint.painn_blob_order(3):
            parts = key.split(".")
            node = top
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, Block())
                node = node._modules[part]
            node.register_parameter(parts[-1], torch.nn.Parameter(torch.from_numpy(np.array(fields[field], dtype=np.float32))))
        top.excl_vol, top.power, top.sigma, top.cutoff = True, 12, 1.5, 5.0
        buf = io.BytesIO()
        torch.save(top, buf)
        return buf.getvalue(), blob
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_checkpoint_reader_rejects_mutated_archives(tmp_path):
    """Bit flips / truncations / header damage of a checkpoint archive: ``load_painn_blob`` raises (ValueError, KeyError,
    zipfile / pickle errors ...) or returns a blob of the right size -- it never hangs and never allocates more than the
    archive warrants."""
    import pickle
    import random
    import resource

    from surface_sampling_amd import checkpoint

    base, want = _checkpoint_bytes()
    path = tmp_path / "m"
    path.write_bytes(base)
    blob = checkpoint.load_painn_blob(str(path))
    assert np.array_equal(blob, want)          # the undamaged archive loads, bit for bit
    n = blob.size
    rng = random.Random(3)
    soft, hard = resource.getrlimit(resource.RLIMIT_AS)
    resource.setrlimit(resource.RLIMIT_AS, (8 << 30, hard))       # a runaway allocation becomes a MemoryError
    try:
        outcomes = {"ok": 0, "rejected": 0}
        for it in range(300):
            b = bytearray(base)
            op = rng.randrange(5)
            if op == 0: b = b[:rng.randrange(len(b))]
            elif op == 1:
                for _ in range(rng.randrange(1, 6)): b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            elif op == 2:
                # damage inside the pickle (first 30 kB hold data.pkl: tensor offsets, sizes, strides)
                for _ in range(rng.randrange(1, 6)): b[rng.randrange(min(len(b), 30000))] = rng.randrange(256)
            elif op == 3:
                i = rng.randrange(len(b)); b[i:i] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 200)))
            else:
                i, j = sorted(rng.randrange(len(b)) for _ in range(2)); del b[i:min(j, i + 5000)]
            path.write_bytes(bytes(b))
            try:
                out = checkpoint.load_painn_blob(str(path))
                assert out.size == n and out.dtype == np.float32
                outcomes["ok"] += 1
            except (ValueError, KeyError, IndexError, EOFError, OSError, zipfile.BadZipFile, pickle.UnpicklingError,
                    AttributeError, TypeError, MemoryError, OverflowError, UnicodeDecodeError, NotImplementedError,
                    struct_error()):
                outcomes["rejected"] += 1
        assert outcomes["rejected"] > 50, outcomes
    finally:
        resource.setrlimit(resource.RLIMIT_AS, (soft, hard))


def struct_error():
    import struct

    return struct.error
