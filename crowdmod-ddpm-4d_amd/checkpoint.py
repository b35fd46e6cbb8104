"""Read and write the reference's checkpoint files without torch.

The reference stores `torch.save({"opt": optimizer.state_dict(), "model": model.state_dict()}, path)`
(/root/reference/utils/utils.py:140-147) and reads `torch.load(path, map_location="cpu",
weights_only=True)["model"]` (models/diffusion/ddpm.py:161,288,340).  A `.pth` written by
torch >= 1.6 is a ZIP archive: `<name>/data.pkl` (a pickle whose tensors are persistent-id
references to storages), `<name>/data/<key>` (raw little-endian storage bytes) and
`<name>/version`.  This module parses exactly that with a restricted unpickler (no code from
the file is executed: only the handful of globals a state_dict needs are honoured) and
writes the same layout, so files round-trip with the reference in both directions.
"""
from __future__ import annotations

import io
import pickle
import struct
import zipfile
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

_STORAGE_DTYPES = {
    "FloatStorage": np.float32, "DoubleStorage": np.float64, "HalfStorage": np.float16,
    "LongStorage": np.int64, "IntStorage": np.int32, "ShortStorage": np.int16,
    "CharStorage": np.int8, "ByteStorage": np.uint8, "BoolStorage": np.bool_,
}


class _StorageType:
    def __init__(self, name):
        self.name = name
        self.dtype = _STORAGE_DTYPES[name]


class _LazyStorage:
    def __init__(self, zf, prefix, key, dtype, numel):
        self.zf, self.prefix, self.key, self.dtype, self.numel = zf, prefix, key, dtype, numel
        self._arr = None

    def array(self):
        if self._arr is None:
            raw = self.zf.read(f"{self.prefix}/data/{self.key}")
            self._arr = np.frombuffer(raw, dtype=self.dtype, count=self.numel)
        return self._arr


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None,
                       metadata=None):
    flat = storage.array()
    size, stride = tuple(int(v) for v in size), tuple(int(v) for v in stride)
    storage_offset = int(storage_offset)
    # Bounds of the strided view, as torch's weights_only loader enforces them: a crafted file must not
    # make as_strided read outside the storage bytes of its zip entry.
    if len(size) != len(stride) or storage_offset < 0 or any(v < 0 for v in size) or any(v < 0 for v in stride):
        raise pickle.UnpicklingError(f"invalid tensor geometry: offset {storage_offset}, size {size}, stride {stride}")
    numel = 1
    for v in size:
        numel *= v
    if numel > flat.size:
        # a broadcast (stride-0) view stays inside the storage whatever its size: torch keeps it a view, this reader
        # materialises it -- a crafted (1 << 40,) x stride 0 entry must not become a terabyte host allocation
        raise pickle.UnpicklingError(f"tensor view of {numel} elements over a storage of {flat.size}: a state_dict "
                                     f"tensor never has more elements than its storage")
    if numel > 0:
        last = storage_offset + sum((n - 1) * st for n, st in zip(size, stride))
        if last >= flat.size:
            raise pickle.UnpicklingError(f"tensor view (offset {storage_offset}, size {size}, stride {stride}) "
                                         f"reaches element {last} of a storage of {flat.size}")
    else:
        return np.zeros(size, dtype=flat.dtype)
    if len(size) == 0:
        return np.array(flat[storage_offset], dtype=flat.dtype)
    itemsize = flat.dtype.itemsize
    view = np.lib.stride_tricks.as_strided(flat[storage_offset:], shape=size,
                                           strides=tuple(s * itemsize for s in stride), writeable=False)
    return np.ascontiguousarray(view)


def _rebuild_parameter(data, requires_grad, backward_hooks):
    return data


class _Unpickler(pickle.Unpickler):
    def __init__(self, fh, zf, prefix):
        super().__init__(fh)
        self.zf, self.prefix = zf, prefix

    def find_class(self, module, name):
        if module == "collections" and name == "OrderedDict":
            return OrderedDict
        if module == "torch._utils" and name == "_rebuild_tensor_v2":
            return _rebuild_tensor_v2
        if module == "torch._utils" and name == "_rebuild_parameter":
            return _rebuild_parameter
        if module == "torch" and name in _STORAGE_DTYPES:
            return _StorageType(name)
        if module == "torch" and name == "Size":
            return tuple
        raise pickle.UnpicklingError(f"checkpoint references {module}.{name}, which a state_dict does not need")

    def persistent_load(self, pid):
        kind, stype, key, _location, numel = pid[:5]
        if kind != "storage":
            raise pickle.UnpicklingError(f"unknown persistent id {kind!r}")
        return _LazyStorage(self.zf, self.prefix, key, stype.dtype, numel)


def load(path: str):
    """Equivalent of torch.load(path, map_location='cpu', weights_only=True) with numpy leaves."""
    with zipfile.ZipFile(path) as zf:
        pkl = [n for n in zf.namelist() if n.endswith("/data.pkl")]
        if not pkl:
            raise ValueError(f"{path}: not a torch zip checkpoint (legacy tar/pickle files are not supported)")
        prefix = pkl[0][: -len("/data.pkl")]
        return _Unpickler(io.BytesIO(zf.read(pkl[0])), zf, prefix).load()


def load_model_state(path: str) -> Dict[str, np.ndarray]:
    """The `['model']` state_dict of a reference checkpoint (ddpm.py:288)."""
    obj = load(path)
    return OrderedDict(obj["model"]) if isinstance(obj, dict) and "model" in obj else OrderedDict(obj)


# ------------------------------------------------------------------------------------
# writer: a protocol-2 pickle emitted by hand (only the opcodes a state_dict needs)
# ------------------------------------------------------------------------------------
class _P2:
    def __init__(self):
        self.b = io.BytesIO()
        self.b.write(b"\x80\x02")

    def glob(self, module, name):
        self.b.write(b"c" + module.encode() + b"\n" + name.encode() + b"\n")

    def string(self, s):
        raw = s.encode("utf-8")
        self.b.write(b"X" + struct.pack("<I", len(raw)) + raw)

    def integer(self, v):
        v = int(v)
        if 0 <= v < 256:
            self.b.write(b"K" + struct.pack("<B", v))
        elif 0 <= v < 65536:
            self.b.write(b"M" + struct.pack("<H", v))
        elif -2**31 <= v < 2**31:
            self.b.write(b"J" + struct.pack("<i", v))
        else:
            raw = v.to_bytes((v.bit_length() + 8) // 8, "little", signed=True)
            self.b.write(b"\x8a" + struct.pack("<B", len(raw)) + raw)

    def floating(self, v):
        self.b.write(b"G" + struct.pack(">d", float(v)))

    def boolean(self, v):
        self.b.write(b"\x88" if v else b"\x89")

    def none(self):
        self.b.write(b"N")

    def tuple_of(self, emitters):
        self.b.write(b"(")
        for e in emitters:
            e()
        self.b.write(b"t")

    def int_tuple(self, vals):
        self.tuple_of([(lambda v=v: self.integer(v)) for v in vals])

    def empty_ordered_dict(self):
        self.glob("collections", "OrderedDict")
        self.b.write(b")R")

    def value(self, v, storages):
        if isinstance(v, np.ndarray) or isinstance(v, np.generic):
            self.tensor(np.asarray(v), storages)
        elif isinstance(v, bool):
            self.boolean(v)
        elif isinstance(v, int):
            self.integer(v)
        elif isinstance(v, float):
            self.floating(v)
        elif isinstance(v, str):
            self.string(v)
        elif v is None:
            self.none()
        elif isinstance(v, dict):
            self.mapping(v, storages, ordered=isinstance(v, OrderedDict))
        elif isinstance(v, (list, tuple)):
            if isinstance(v, tuple):
                self.tuple_of([(lambda x=x: self.value(x, storages)) for x in v])
            else:
                self.b.write(b"](")
                for x in v:
                    self.value(x, storages)
                self.b.write(b"e")
        else:
            raise TypeError(f"cannot serialise {type(v)} into a checkpoint")

    def mapping(self, d, storages, ordered):
        if ordered:
            self.empty_ordered_dict()
        else:
            self.b.write(b"}")
        self.b.write(b"(")
        for k, v in d.items():
            self.value(k, storages)
            self.value(v, storages)
        self.b.write(b"u")

    def tensor(self, arr, storages):
        arr = np.ascontiguousarray(arr)
        sname = {np.dtype(v): k for k, v in _STORAGE_DTYPES.items()}.get(arr.dtype)
        if sname is None:
            raise TypeError(f"dtype {arr.dtype} has no torch storage type")
        key = str(len(storages))
        storages.append((key, arr))
        self.glob("torch._utils", "_rebuild_tensor_v2")
        self.b.write(b"(")
        # persistent id ('storage', torch.XStorage, key, 'cpu', numel)
        self.b.write(b"(")
        self.string("storage")
        self.glob("torch", sname)
        self.string(key)
        self.string("cpu")
        self.integer(arr.size)
        self.b.write(b"tQ")
        self.integer(0)
        self.int_tuple(arr.shape)
        strides = []
        acc = 1
        for s in reversed(arr.shape):
            strides.append(acc)
            acc *= int(s)
        self.int_tuple(tuple(reversed(strides)))
        self.boolean(False)
        self.empty_ordered_dict()
        self.b.write(b"tR")

    def finish(self):
        self.b.write(b".")
        return self.b.getvalue()


def save(obj, path: str, archive_name: str = "archive") -> None:
    """Write `obj` (nested dict / list / scalars / numpy arrays) as a torch zip checkpoint that
    `torch.load(path, weights_only=True)` reads back with tensors in place of the arrays."""
    p = _P2()
    storages = []
    p.value(obj, storages)
    payload = p.finish()
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_STORED) as zf:
        zf.writestr(f"{archive_name}/data.pkl", payload)
        zf.writestr(f"{archive_name}/byteorder", "little")
        for key, arr in storages:
            zf.writestr(f"{archive_name}/data/{key}", arr.tobytes())
        zf.writestr(f"{archive_name}/version", "3\n")


def save_checkpoint(model_state: Dict[str, np.ndarray], path: str, opt_state=None) -> None:
    """utils/utils.py:140-147: {'opt': ..., 'model': ...}."""
    save({"opt": opt_state if opt_state is not None else {}, "model": OrderedDict(model_state)}, path)
