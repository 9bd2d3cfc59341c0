"""Model-agnostic API layer (reference: src/inference.rs).

* ``DepthPrediction``      -- src/inference.rs:10-20
* ``rgb_to_input_tensor``  -- src/inference.rs:79-121 (device kernel; Err on a wrong byte length)
* ``infer_from_rgb``       -- src/inference.rs:128-137
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from .depth_pro import DepthPro, Device, _stream_ptr


@dataclass
class DepthPrediction:
    depth: torch.Tensor
    focallength_px: Optional[torch.Tensor]
    fovy_rad: Optional[torch.Tensor]

    def has_intrinsics(self) -> bool:
        return self.focallength_px is not None or self.fovy_rad is not None


def rgb_to_input_tensor(rgb: bytes, width: int, height: int, device: Device) -> torch.Tensor:
    """Packed RGB bytes (row-major, width*height*3) -> normalised [1,3,H,W] fp32 on the GPU."""
    expected = width * height * 3
    if len(rgb) != expected:  # inference.rs:90-95
        raise _lib.MdError(_lib.MD_ERR_SHAPE, f"expected {expected} RGB bytes for {width}x{height}, got {len(rgb)}")
    dev = torch.device("cuda", device.ordinal)
    src = torch.frombuffer(bytearray(rgb), dtype=torch.uint8).to(dev)
    out = torch.empty((1, 3, height, width), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().md_op_rgb_to_input(device.handle, C.c_void_p(src.data_ptr()), len(rgb), width, height,
                                              C.c_void_p(out.data_ptr()), _stream_ptr(device.ordinal)))
    return out


def infer_depth(model: DepthPro, x: torch.Tensor) -> DepthPrediction:
    """`DepthModel::infer_depth` for DepthPro (src/inference.rs:36-40)."""
    r = model.infer(x)
    return DepthPrediction(r.depth, r.focallength_px, r.fovy_rad)


def infer_from_rgb(model: DepthPro, rgb: bytes, width: int, height: int) -> DepthPrediction:
    r = model.infer_from_rgb(rgb, width, height)
    return DepthPrediction(r.depth, r.focallength_px, r.fovy_rad)
