"""Reader for the reference's on-disk checkpoints: Burn `NamedMpkFileRecorder<HalfPrecisionSettings>` files
(`.mpk`, depth_pro/mod.rs:193-208; src/model/mod.rs:62-100).

Format as published by Burn 0.19 (`burn-core` record/file.rs + record/tensor.rs; the crate is not vendored in the
reference tree and no `.mpk` exists there -- `.gitignore:8,17,32` -- so this reader is **validated only against files
this module writes itself**): one MessagePack map

    { "metadata": {float, int, format, version, settings}, "item": <record> }

where <record> nests maps by field name (named = `rmp_serde::to_vec_named`), `Vec<Module>` as arrays, `Option::None`
as nil, and every parameter as  { "id": str, "param": { "bytes": bin, "shape": [..], "dtype": "F16"|"F32"|"BF16" } }.
The walker is tolerant: any map holding `bytes` + `shape` (+ `dtype`) is a tensor, `param` / `item` wrappers and
`id` entries do not contribute to the dotted path. The resulting names are the Burn field paths the engine's
container uses.

Layout: a Burn record holds `nn::Linear` weights as `[d_input, d_output]` (the reference's importers write their records
behind `PyTorchToBurnAdapter`, tool/import_da3.rs:199, which transposes PyTorch's `[out, in]`); the engine's inventory holds
them `[out, in]`. Every rank-2 tensor of these models is such a weight, so `read_mpk` transposes rank-2 tensors on the way
in and `write_mpk` on the way out (`burn_layout=True`, the default). The C ABI reads the same files natively
(`md_depth_pro_load` / `md_da3_load` dispatch on the first bytes; csrc/md_weights.cpp): this module is the Python twin for the
importer and the tests."""
from __future__ import annotations

from typing import Dict, List

import numpy as np

_DTYPES = {"F16": np.float16, "F32": np.float32, "F64": np.float64, "f16": np.float16, "f32": np.float32}


def _bf16_to_f32(raw: bytes) -> np.ndarray:
    u = np.frombuffer(raw, np.uint16).astype(np.uint32) << 16
    return u.view(np.float32)


def _tensor(node: dict) -> np.ndarray:
    dt = node.get("dtype", "F32")
    if isinstance(dt, dict):  # externally tagged enum variants serialise as {"F16": null} under some settings
        dt = next(iter(dt))
    raw = bytes(node["bytes"])
    shape = [int(d) for d in node["shape"]]
    if dt in ("BF16", "bf16"):
        arr = _bf16_to_f32(raw)
    elif dt in _DTYPES:
        arr = np.frombuffer(raw, _DTYPES[dt]).astype(np.float32)
    else:
        raise ValueError(f"unsupported tensor dtype `{dt}` in .mpk")
    if arr.size != int(np.prod(shape)) if shape else arr.size != 1:
        raise ValueError(f".mpk tensor has {arr.size} elements for shape {shape}")
    return arr.reshape(shape)


def _walk(node, path: List[str], out: Dict[str, np.ndarray]) -> None:
    if node is None:
        return
    if isinstance(node, dict):
        keys = {k.decode() if isinstance(k, bytes) else k for k in node}
        node = {(k.decode() if isinstance(k, bytes) else k): v for k, v in node.items()}
        if "bytes" in keys and "shape" in keys:
            out[".".join(path)] = _tensor(node)
            return
        for k, v in node.items():
            if k == "id":
                continue
            _walk(v, path if k in ("param", "item") else path + [str(k)], out)
    elif isinstance(node, (list, tuple)):
        for i, v in enumerate(node):
            _walk(v, path + [str(i)], out)


def read_mpk(path: str, burn_layout: bool = True) -> Dict[str, np.ndarray]:
    """All tensors of a Burn `.mpk` record as fp32 arrays keyed by dotted field path, Linear weights as `[out, in]`."""
    import msgpack
    with open(path, "rb") as f:
        root = msgpack.unpackb(f.read(), raw=False, strict_map_key=False)
    if not isinstance(root, dict) or "item" not in root:
        raise ValueError(f"{path}: not a Burn record (no `item` entry)")
    out: Dict[str, np.ndarray] = {}
    _walk(root["item"], [], out)
    if not out:
        raise ValueError(f"{path}: no tensors found")
    if burn_layout:
        out = {k: (np.ascontiguousarray(v.T) if v.ndim == 2 else v) for k, v in out.items()}
    return out


def write_mpk(path: str, tensors: Dict[str, np.ndarray], dtype: str = "F16", burn_layout: bool = True) -> None:
    """Writes the structure described above (used by the tests; mirrors `HalfPrecisionSettings` for dtype F16). `tensors` holds
    Linear weights `[out, in]`; the file holds them `[d_input, d_output]` like a Burn record."""
    import msgpack
    root: dict = {}
    for name, arr in tensors.items():
        node = root
        parts = name.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        a = np.ascontiguousarray(arr.T if (burn_layout and np.ndim(arr) == 2) else arr, np.float32)
        raw = a.astype(np.float16).tobytes() if dtype == "F16" else a.tobytes()
        node[parts[-1]] = {"id": name, "param": {"bytes": raw, "shape": list(a.shape), "dtype": dtype}}

    def listify(n):  # maps whose keys are 0..k-1 become arrays, like Vec<Module> records
        if not isinstance(n, dict) or "bytes" in n:
            return n
        n = {k: listify(v) for k, v in n.items()}
        if n and all(k.isdigit() for k in n) and sorted(int(k) for k in n) == list(range(len(n))):
            return [n[str(i)] for i in range(len(n))]
        return n

    doc = {"metadata": {"float": "f16" if dtype == "F16" else "f32", "int": "i32", "format": "burn_core::record::file::NamedMpkFileRecorder",
                        "version": "0.19.1", "settings": "HalfPrecisionSettings" if dtype == "F16" else "FullPrecisionSettings"},
           "item": listify(root)}
    with open(path, "wb") as f:
        f.write(msgpack.packb(doc, use_bin_type=True))
