"""Read NFF PaiNN checkpoints (torch whole-module zip, ``best_model``) without torch or nff.

The reference loads these with ``nff.utils.cuda.load_model`` (call site
``scripts/sample_surface.py:164-167``); neither ``nff`` nor a matching ``torch``
pickle environment is needed here: the archive is a plain zip holding
``archive/data.pkl`` (the pickled module tree) and ``archive/data/<key>`` raw
little-endian fp32 storages.  A restricted ``pickle.Unpickler`` maps every class
in the module tree to an inert stub and every tensor to a numpy view, then the
tree is walked to produce the flat state dict.  Nothing from the pickle is ever
executed: ``find_class`` only hands out stubs from a fixed whitelist.

The flat weight blob layout (``PAINN_BLOB_ORDER``) is the one documented in
``include/vssr_eval.h`` and consumed by ``vssr_create``.
"""

from __future__ import annotations

import io
import pickle
import zipfile
from collections import OrderedDict

import numpy as np

# Hyper-parameters of the PaiNN family shipped with the reference
# (tutorials/data/SrTiO3_001/nff/model01/params.json: feat_dim 128, n_rbf 20,
#  cutoff 5.0, num_conv 3, activation swish, excl_vol, V_ex_power 12, V_ex_sigma 1.5).
DEFAULT_HPARAMS = {
    "feat_dim": 128,
    "n_rbf": 20,
    "num_conv": 3,
    "cutoff": 5.0,
    "excl_vol": True,
    "V_ex_power": 12,
    "V_ex_sigma": 1.5,
    "readout_hidden": 64,
    "n_embed": 100,
}


class _Stub:
    """Inert stand-in for any class found in the checkpoint pickle."""

    def __init__(self, *args, **kwargs):
        self._args = args
        self._kwargs = kwargs

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self._state = state

    def __call__(self, *args, **kwargs):  # functools.partial(...) results etc.
        return _Stub(*args, **kwargs)


class _StorageRef:
    def __init__(self, key, numel):
        self.key = key
        self.numel = numel


def _rebuild_tensor_v2(storage, storage_offset, size, stride, *unused):
    return ("tensor", storage, storage_offset, tuple(size), tuple(stride))


def _rebuild_parameter(data, requires_grad, backward_hooks, *unused):
    return data


_ALLOWED_PREFIXES = (
    "nff.",
    "torch.nn.",
    "torch._utils",
    "torch.FloatStorage",
    "torch.",
    "functools.",
    "collections.",
    "__builtin__.",
    "builtins.",
)


class _CheckpointUnpickler(pickle.Unpickler):
    def __init__(self, file, archive: zipfile.ZipFile, prefix: str):
        super().__init__(file, encoding="latin1")
        self._archive = archive
        self._prefix = prefix
        self._cache: dict[str, np.ndarray] = {}

    def find_class(self, module, name):
        full = f"{module}.{name}"
        if full == "collections.OrderedDict":
            return OrderedDict
        if full in ("__builtin__.set", "builtins.set"):
            return set
        if full == "torch._utils._rebuild_tensor_v2":
            return _rebuild_tensor_v2
        if full == "torch._utils._rebuild_parameter":
            return _rebuild_parameter
        if not full.startswith(_ALLOWED_PREFIXES):
            raise pickle.UnpicklingError(f"global {full!r} is not allowed in a checkpoint")
        # every other global (module classes, storages, init functions) -> inert stub type
        return type(name, (_Stub,), {"__module__": module})

    def persistent_load(self, pid):
        # ('storage', <storage type>, key, location, numel)
        if not (isinstance(pid, tuple) and pid and pid[0] == "storage"):
            raise pickle.UnpicklingError(f"unexpected persistent id {pid!r}")
        _, storage_type, key, _location, numel = pid
        if getattr(storage_type, "__name__", "") != "FloatStorage":
            raise pickle.UnpicklingError("only torch.FloatStorage is supported")
        return _StorageRef(str(key), int(numel))

    def storage(self, ref: _StorageRef) -> np.ndarray:
        if ref.key not in self._cache:
            raw = self._archive.read(f"{self._prefix}/data/{ref.key}")
            arr = np.frombuffer(raw, dtype="<f4")
            if arr.size != ref.numel:
                raise ValueError(f"storage {ref.key}: {arr.size} floats, expected {ref.numel}")
            self._cache[ref.key] = arr
        return self._cache[ref.key]


def _materialise(t, unpickler) -> np.ndarray:
    kind, storage, offset, size, stride = t
    assert kind == "tensor"
    base = unpickler.storage(storage)
    # the pickle is untrusted: the strided view must stay inside its storage
    size, stride = tuple(int(x) for x in size), tuple(int(x) for x in stride)
    offset = int(offset)
    if len(size) != len(stride) or offset < 0 or any(n < 0 for n in size) or any(st < 0 for st in stride):
        raise ValueError("checkpoint tensor with a negative offset, size or stride")
    last = offset + sum((n - 1) * st for n, st in zip(size, stride)) if all(n > 0 for n in size) else offset
    if last >= base.size and (len(size) == 0 or all(n > 0 for n in size)):
        raise ValueError(f"checkpoint tensor reaches element {last} of a storage with {base.size}")
    if len(size) == 0:
        return np.array(base[offset], dtype=np.float32)
    if any(n == 0 for n in size):
        return np.zeros(size, dtype=np.float32)
    view = np.lib.stride_tricks.as_strided(
        base[offset:], shape=size, strides=tuple(s * 4 for s in stride), writeable=False
    )
    return np.ascontiguousarray(view, dtype=np.float32)


def _walk(module, prefix, unpickler, out):
    params = getattr(module, "_parameters", None) or {}
    for name, val in params.items():
        if val is not None:
            out[prefix + name] = _materialise(val, unpickler)
    buffers = getattr(module, "_buffers", None) or {}
    for name, val in buffers.items():
        if isinstance(val, tuple) and val and val[0] == "tensor":
            out[prefix + name] = _materialise(val, unpickler)
    for name, child in (getattr(module, "_modules", None) or {}).items():
        if child is not None:
            _walk(child, f"{prefix}{name}.", unpickler, out)


def read_state_dict(path: str) -> "OrderedDict[str, np.ndarray]":
    """Return the flat ``name -> float32 ndarray`` state dict of a ``best_model`` file."""
    with zipfile.ZipFile(path) as archive:
        pkl = [n for n in archive.namelist() if n.endswith("/data.pkl")]
        if len(pkl) != 1:
            raise ValueError(f"{path}: not a torch zip checkpoint")
        prefix = pkl[0][: -len("/data.pkl")]
        unpickler = _CheckpointUnpickler(io.BytesIO(archive.read(pkl[0])), archive, prefix)
        root = unpickler.load()
        out: OrderedDict[str, np.ndarray] = OrderedDict()
        _walk(root, "", unpickler, out)
    return out


def read_model_attrs(path: str) -> dict:
    """Scalar attributes of the pickled Painn module (excl_vol, power, sigma, ...)."""
    with zipfile.ZipFile(path) as archive:
        pkl = [n for n in archive.namelist() if n.endswith("/data.pkl")][0]
        prefix = pkl[: -len("/data.pkl")]
        unpickler = _CheckpointUnpickler(io.BytesIO(archive.read(pkl)), archive, prefix)
        root = unpickler.load()
    attrs = {}
    for k, v in vars(root).items():
        if isinstance(v, (bool, int, float, str)) or v is None:
            attrs[k] = v
        elif isinstance(v, (list, tuple, dict)) and not k.startswith("_"):
            try:
                attrs[k] = _plain(v)
            except TypeError:
                pass
    return attrs


def _plain(v):
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, dict):
        return {str(k): _plain(x) for k, x in v.items()}
    raise TypeError(type(v))


def painn_blob_order(num_conv: int = 3) -> list[tuple[str, str]]:
    """Canonical (field, state-dict key) order of the flat weight blob (see include/vssr_eval.h)."""
    order = [("embed", "embed_block.atom_embed.weight")]
    for l in range(num_conv):
        m = f"message_blocks.{l}.inv_message."
        u = f"update_blocks.{l}."
        order += [
            (f"msg{l}.W1", m + "inv_dense.layers.0.weight"),
            (f"msg{l}.b1", m + "inv_dense.layers.0.bias"),
            (f"msg{l}.W2", m + "inv_dense.layers.1.weight"),
            (f"msg{l}.b2", m + "inv_dense.layers.1.bias"),
            (f"msg{l}.Wd", m + "dist_embed.block.1.weight"),
            (f"msg{l}.bd", m + "dist_embed.block.1.bias"),
            (f"upd{l}.U", u + "u_mat.weight"),
            (f"upd{l}.V", u + "v_mat.weight"),
            (f"upd{l}.W3", u + "s_dense.0.weight"),
            (f"upd{l}.b3", u + "s_dense.0.bias"),
            (f"upd{l}.W4", u + "s_dense.1.weight"),
            (f"upd{l}.b4", u + "s_dense.1.bias"),
        ]
    r = "readout_blocks.0.readoutdict.energy."
    order += [
        ("readout.W5", r + "0.weight"),
        ("readout.b5", r + "0.bias"),
        ("readout.w6", r + "1.weight"),
        ("readout.b6", r + "1.bias"),
    ]
    return order


def painn_blob_shapes(hp: dict | None = None) -> "OrderedDict[str, tuple[int, ...]]":
    hp = {**DEFAULT_HPARAMS, **(hp or {})}
    F, R, H = hp["feat_dim"], hp["n_rbf"], hp["readout_hidden"]
    shapes: OrderedDict[str, tuple[int, ...]] = OrderedDict()
    shapes["embed"] = (hp["n_embed"], F)
    for l in range(hp["num_conv"]):
        shapes[f"msg{l}.W1"] = (F, F)
        shapes[f"msg{l}.b1"] = (F,)
        shapes[f"msg{l}.W2"] = (3 * F, F)
        shapes[f"msg{l}.b2"] = (3 * F,)
        shapes[f"msg{l}.Wd"] = (3 * F, R)
        shapes[f"msg{l}.bd"] = (3 * F,)
        shapes[f"upd{l}.U"] = (F, F)
        shapes[f"upd{l}.V"] = (F, F)
        shapes[f"upd{l}.W3"] = (F, 2 * F)
        shapes[f"upd{l}.b3"] = (F,)
        shapes[f"upd{l}.W4"] = (3 * F, F)
        shapes[f"upd{l}.b4"] = (3 * F,)
    shapes["readout.W5"] = (H, F)
    shapes["readout.b5"] = (H,)
    shapes["readout.w6"] = (1, H)
    shapes["readout.b6"] = (1,)
    return shapes


def state_dict_to_blob(sd: dict, hp: dict | None = None) -> np.ndarray:
    """Flatten a PaiNN state dict into the canonical fp32 blob."""
    hp = {**DEFAULT_HPARAMS, **(hp or {})}
    shapes = painn_blob_shapes(hp)
    parts = []
    for field, key in painn_blob_order(hp["num_conv"]):
        arr = np.asarray(sd[key], dtype=np.float32)
        if tuple(arr.shape) != shapes[field]:
            raise ValueError(f"{key}: shape {arr.shape}, expected {shapes[field]}")
        parts.append(arr.reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts), dtype="<f4")


def blob_to_fields(blob: np.ndarray, hp: dict | None = None) -> "OrderedDict[str, np.ndarray]":
    """Split a canonical blob back into named arrays (views)."""
    shapes = painn_blob_shapes(hp)
    out: OrderedDict[str, np.ndarray] = OrderedDict()
    off = 0
    for field, shape in shapes.items():
        n = int(np.prod(shape))
        out[field] = blob[off : off + n].reshape(shape)
        off += n
    if off != blob.size:
        raise ValueError(f"blob has {blob.size} floats, layout needs {off}")
    return out


# pickled module attribute -> hyper-parameter of this backend
_ATTR_TO_HPARAM = (("excl_vol", "excl_vol"), ("power", "V_ex_power"), ("sigma", "V_ex_sigma"), ("cutoff", "cutoff"))


def check_model_against_hparams(path: str, sd: dict, hp: dict | None = None, attrs: dict | None = None) -> None:
    """A checkpoint whose stored hyper-parameters differ from the ones the engine will be created with would load and give
    wrong energies silently: compare the pickled module attributes (excluded volume on/off, power, sigma, cutoff) and the
    radial-basis frequencies (a learnable ``n`` that is no longer 1..n_rbf) with ``hp`` and raise on any difference."""
    hp = {**DEFAULT_HPARAMS, **(hp or {})}
    if attrs is None:
        attrs = read_model_attrs(path)
    for attr, key in _ATTR_TO_HPARAM:
        if attr in attrs and key in hp and attrs[attr] is not None:
            a, b = attrs[attr], hp[key]
            same = (bool(a) == bool(b)) if isinstance(a, bool) or isinstance(b, bool) else abs(float(a) - float(b)) < 1e-9
            if not same:
                raise ValueError(f"{path}: checkpoint has {attr}={a!r} but the backend is configured with {key}={b!r}")
    n_ref = np.arange(1, hp["n_rbf"] + 1, dtype=np.float64)
    consumed = {key for _, key in painn_blob_order(hp["num_conv"])}
    for key, arr in sd.items():
        if key in consumed:
            continue
        if key.endswith("dist_embed.block.0.n"):
            if arr.shape != n_ref.shape or np.abs(np.asarray(arr, np.float64) - n_ref).max() > 1e-6:
                raise ValueError(f"{path}: learnable radial frequencies {key} differ from 1..{hp['n_rbf']}")
        elif key.endswith(("means", "stddevs")) and arr.size:
            raise ValueError(f"{path}: non-empty ScaleShift parameter {key} is not supported")
        else:
            raise ValueError(f"{path}: parameter {key} {tuple(arr.shape)} is not part of the supported PaiNN layout")


def load_painn_blob(path: str, hp: dict | None = None) -> np.ndarray:
    """``best_model`` (torch zip) or ``.f32`` raw blob -> canonical fp32 blob."""
    if zipfile.is_zipfile(path):
        sd = read_state_dict(path)
        check_model_against_hparams(path, sd, hp)
        return state_dict_to_blob(sd, hp)
    blob = np.fromfile(path, dtype="<f4")
    blob_to_fields(blob, hp)  # validates the size
    return blob
