"""Byte-level builder of Burn `NamedMpkFileRecorder<HalfPrecisionSettings>` records for the tests of the C ABI's `.mpk` reader
(`md_depth_pro_load` / `md_da3_load` / `md_checkpoint_info`, csrc/md_weights.cpp).

It shares NO code with either reader (csrc/md_weights.cpp, burn_depth_amd/mpk.py) and does not use the `msgpack` package: every
MessagePack byte is emitted here with `struct`, from the format's public specification. What it writes follows Burn 0.19's
published record layout (burn-core record/file.rs, record/tensor.rs -- the reference's `DepthPro::load` argument,
depth_pro/mod.rs:193-208; no `.mpk` exists in the reference tree, so the layout itself stays unvalidated on a real record):

    { "metadata": {float, int, format, version, settings}, "item": <record> }

<record>: nested maps by field name, `Vec<Module>` as arrays, `Option::None` as nil, a parameter as
{ "id": str, "param": { "bytes": bin, "shape": [..], "dtype": "F16" } }, `nn::Linear` weights as [d_input, d_output].
"""
from __future__ import annotations

import struct
from typing import Dict

import numpy as np


def _str(s: str) -> bytes:
    b = s.encode()
    n = len(b)
    if n < 32:
        return bytes([0xA0 | n]) + b
    if n < 256:
        return b"\xd9" + struct.pack(">B", n) + b
    return b"\xda" + struct.pack(">H", n) + b


def _uint(v: int) -> bytes:
    if v < 128:
        return bytes([v])
    if v < 1 << 8:
        return b"\xcc" + struct.pack(">B", v)
    if v < 1 << 16:
        return b"\xcd" + struct.pack(">H", v)
    if v < 1 << 32:
        return b"\xce" + struct.pack(">I", v)
    return b"\xcf" + struct.pack(">Q", v)


def _bin(b: bytes) -> bytes:
    n = len(b)
    if n < 256:
        return b"\xc4" + struct.pack(">B", n) + b
    if n < 1 << 16:
        return b"\xc5" + struct.pack(">H", n) + b
    return b"\xc6" + struct.pack(">I", n) + b


def _map_head(n: int) -> bytes:
    if n < 16:
        return bytes([0x80 | n])
    if n < 1 << 16:
        return b"\xde" + struct.pack(">H", n)
    return b"\xdf" + struct.pack(">I", n)


def _array_head(n: int) -> bytes:
    if n < 16:
        return bytes([0x90 | n])
    if n < 1 << 16:
        return b"\xdc" + struct.pack(">H", n)
    return b"\xdd" + struct.pack(">I", n)


class _Leaf:
    def __init__(self, name: str, arr: np.ndarray, dtype: str):
        self.name, self.arr, self.dtype = name, arr, dtype


def _encode(node) -> bytes:
    if node is None:
        return b"\xc0"
    if isinstance(node, bool):
        return b"\xc3" if node else b"\xc2"
    if isinstance(node, int):
        return _uint(node) if node >= 0 else b"\xd3" + struct.pack(">q", node)
    if isinstance(node, float):
        return b"\xcb" + struct.pack(">d", node)
    if isinstance(node, str):
        return _str(node)
    if isinstance(node, _Leaf):
        a = node.arr
        if node.dtype == "F16":
            raw = a.astype("<f2").tobytes()
        elif node.dtype == "F32":
            raw = a.astype("<f4").tobytes()
        elif node.dtype == "BF16":
            u = a.astype("<f4").view(np.uint32)
            raw = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype("<u2").tobytes()  # round to nearest even (finite values)
        else:
            raise ValueError(node.dtype)
        param = (_map_head(3) + _str("bytes") + _bin(raw) + _str("shape") + _array_head(a.ndim) + b"".join(_uint(int(d)) for d in a.shape) +
                 _str("dtype") + _str(node.dtype))
        return _map_head(2) + _str("id") + _str(node.name) + _str("param") + param
    if isinstance(node, list):
        return _array_head(len(node)) + b"".join(_encode(v) for v in node)
    if isinstance(node, dict):
        return _map_head(len(node)) + b"".join(_str(k) + _encode(v) for k, v in node.items())
    raise TypeError(type(node))


def build_record(tensors: Dict[str, np.ndarray], dtype: str = "F16", extras: bool = True) -> bytes:
    """`tensors`: the engine's inventory (names = Burn field paths, Linear weights [out, in]). Returns the bytes of a Burn record:
    Linear weights transposed to [d_input, d_output], digit-keyed levels as arrays. `extras` sprinkles values a tolerant reader has
    to skip: an `Option::None` field, an integer and a float leaf, a boolean."""
    root: dict = {}
    for name, arr in tensors.items():
        node = root
        parts = name.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        a = np.asarray(arr, np.float32)
        if a.ndim == 2:
            a = a.T
        node[parts[-1]] = _Leaf(name, np.ascontiguousarray(a), dtype)

    def listify(n):
        if not isinstance(n, dict):
            return n
        n = {k: listify(v) for k, v in n.items()}
        if n and all(k.isdigit() for k in n) and sorted(int(k) for k in n) == list(range(len(n))):
            return [n[str(i)] for i in range(len(n))]
        return n

    item = listify(root)
    if extras:
        item["mask_token"] = None             # an Option<Param> that is None
        item["record_revision"] = 70000        # a uint32 leaf
        item["scale_hint"] = 0.5               # a float64 leaf
        item["frozen"] = False
    doc = {"metadata": {"float": "f16" if dtype == "F16" else "f32", "int": "i32", "format": "burn_core::record::file::NamedMpkFileRecorder",
                        "version": "0.19.1", "settings": "HalfPrecisionSettings" if dtype == "F16" else "FullPrecisionSettings"},
           "item": item}
    return _encode(doc)


def write_record(path: str, tensors: Dict[str, np.ndarray], dtype: str = "F16", extras: bool = True) -> None:
    with open(path, "wb") as f:
        f.write(build_record(tensors, dtype, extras))
