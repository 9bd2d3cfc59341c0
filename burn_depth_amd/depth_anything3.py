"""Host-side mirror of the reference's Depth-Anything-v3 interface (metric_large / mono head).

* ``DepthAnything3.new(device, config)``  -- depth_anything3/mod.rs:253-286
* ``DepthAnything3.load_file``            -- example/correctness.rs:977-982 (record load)
* ``DepthAnything3.infer(x)``             -- depth_anything3/mod.rs:288-291
* ``img_size`` / ``patch_size``           -- depth_anything3/mod.rs:566-572
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from .config import DepthAnything3Config
from .depth_pro import DepthPro, Device, _stream_ptr


@dataclass
class DepthAnything3Inference:
    """depth_anything3/mod.rs:231-239 (the mono head fills `depth` only; the dual head every field)."""
    depth: torch.Tensor
    depth_confidence: Optional[torch.Tensor] = None
    aux: Optional[torch.Tensor] = None
    aux_confidence: Optional[torch.Tensor] = None
    pose_encoding: Optional[torch.Tensor] = None
    extrinsics: Optional[torch.Tensor] = None
    intrinsics: Optional[torch.Tensor] = None


def _c_cfg(cfg: DepthAnything3Config):
    keep = cfg.variant.encode()
    return _lib.MdDa3Cfg(keep, int(cfg.image_size), int(cfg.precision), int(cfg.max_batch), float(cfg.ln_eps),
                         int(cfg.image_width)), keep


class DepthAnything3(DepthPro):
    """Shares the record / timing / query plumbing of the md_model_t handle with DepthPro."""

    @staticmethod
    def new(device: Device, config: Optional[DepthAnything3Config] = None, seed: int = 0, init_scheme: int = 0) -> "DepthAnything3":
        config = config or DepthAnything3Config.metric_large()
        c, keep = _c_cfg(config)
        h = C.c_void_p()
        _lib.check(_lib.load().md_da3_create(device.handle, C.byref(c), C.c_uint64(seed), int(init_scheme), C.byref(h)))
        return DepthAnything3(device, h, config)

    @staticmethod
    def load_file(device: Device, config: DepthAnything3Config, path: str) -> "DepthAnything3":
        c, keep = _c_cfg(config)
        h = C.c_void_p()
        _lib.check(_lib.load().md_da3_load(device.handle, C.byref(c), os.fspath(path).encode(), C.byref(h)))
        return DepthAnything3(device, h, config)

    def patch_size(self) -> int:
        return self.config.patch_size

    def infer_with_camera(self, x: torch.Tensor, extrinsics: torch.Tensor, intrinsics: torch.Tensor) -> DepthAnything3Inference:
        """`DepthAnything3::infer_with_camera` (mod.rs:301-309): extrinsics [B, V, 3, 4] (world-to-camera), intrinsics [B, V, 3, 3].
        The camera encoder's token (camera.rs:89-110) conditions the backbone; a variant without an encoder ignores both
        (mod.rs:522-527)."""
        B = x.shape[0]
        if extrinsics.dim() != 4 or tuple(extrinsics.shape[2:]) != (3, 4) or extrinsics.shape[0] != B:
            raise _lib.MdError(_lib.MD_ERR_SHAPE, f"expected extrinsics [{B},V,3,4], got {tuple(extrinsics.shape)}")
        V = extrinsics.shape[1]
        if tuple(intrinsics.shape) != (B, V, 3, 3):
            raise _lib.MdError(_lib.MD_ERR_SHAPE, f"expected intrinsics [{B},{V},3,3], got {tuple(intrinsics.shape)}")
        return self.infer(x, (extrinsics, intrinsics))

    def infer_raw(self, x: torch.Tensor) -> torch.Tensor:
        """`DepthAnything3::infer_raw` (mod.rs:364-380): [B, C, H, W] -- the dual head's main logits (C = 2), the mono head's
        `forward_raw` result (C = 1)."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise _lib.MdError(_lib.MD_ERR_SHAPE, f"expected [B,3,H,W], got {tuple(x.shape)}")
        x = x.contiguous().to(torch.float32)
        B, _, H, W = x.shape
        out = torch.empty((B, self.config.output_dim, H, W), dtype=torch.float32, device=torch.device("cuda", self.device.ordinal))
        in_kind = _lib.MD_MEM_DEVICE if x.is_cuda else _lib.MD_MEM_HOST
        _lib.check(self._lib.md_da3_infer_raw(self._h, C.c_void_p(x.data_ptr()), B, H, W, in_kind, C.c_void_p(out.data_ptr()),
                                              _lib.MD_MEM_DEVICE, _stream_ptr(self.device.ordinal)))
        return out

    def infer_from_tokens(self, patches, height: int, width: int) -> DepthAnything3Inference:
        """`DepthAnything3::infer_from_tokens` (mod.rs:389-469): the head alone on the four hooks' patch tokens, each
        [B, P | P + 1, din]; no camera prediction. The aux trace is read through the taps (`aux_neck`, `aux_head_input`)."""
        if len(patches) != 4:
            raise _lib.MdError(_lib.MD_ERR_LEVELS, f"Backbone returned fewer hooks ({len(patches)}) than requested (4)")
        toks = [t.contiguous().to(torch.float32) for t in patches]
        B, T, din = toks[0].shape
        if any(tuple(t.shape) != (B, T, din) or t.device != toks[0].device for t in toks):
            raise _lib.MdError(_lib.MD_ERR_SHAPE, "the four hook tensors must share one shape and device")
        dev = torch.device("cuda", self.device.ordinal)
        f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        out = DepthAnything3Inference(depth=f(B, height, width))
        if self.config.dual_head:
            ah, aw = 8 * (height // self.config.patch_size), 8 * (width // self.config.patch_size)
            out.depth_confidence, out.aux, out.aux_confidence = f(B, height, width), f(B, self.config.aux_output_dim - 1, ah, aw), f(B, ah, aw)
        ptr = lambda t: t.data_ptr() if t is not None else None
        o = _lib.MdDa3Outputs(ptr(out.depth), ptr(out.depth_confidence), ptr(out.aux), ptr(out.aux_confidence), None, None, None)
        arr = (C.c_void_p * 4)(*(t.data_ptr() for t in toks))
        in_kind = _lib.MD_MEM_DEVICE if toks[0].is_cuda else _lib.MD_MEM_HOST
        _lib.check(self._lib.md_da3_infer_from_tokens(self._h, arr, int(T), int(B), int(height), int(width), in_kind, C.byref(o),
                                                      _lib.MD_MEM_DEVICE, _stream_ptr(self.device.ordinal)))
        return out

    def infer(self, x: torch.Tensor, _camera=None) -> DepthAnything3Inference:
        if x.dim() != 4 or x.shape[1] != 3:
            raise _lib.MdError(_lib.MD_ERR_SHAPE, f"expected [B,3,H,W], got {tuple(x.shape)}")
        x = x.contiguous().to(torch.float32)
        B, _, H, W = x.shape
        dev = torch.device("cuda", self.device.ordinal)
        depth = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        if not self.config.dual_head:  # no camera encoder either: camera inputs are ignored (mod.rs:522-527)
            self.infer_into(x, depth)
            return DepthAnything3Inference(depth=depth)
        # dual head (`small`): every field of DepthAnything3Inference (mod.rs:231-239, 605-621)
        ps = self.config.patch_size
        ah, aw = 8 * (H // ps), 8 * (W // ps)
        f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        out = DepthAnything3Inference(depth=depth, depth_confidence=f(B, H, W), aux=f(B, self.config.aux_output_dim - 1, ah, aw),
                                      aux_confidence=f(B, ah, aw), pose_encoding=f(B, 1, 9), extrinsics=f(B, 1, 3, 4),
                                      intrinsics=f(B, 1, 3, 3))
        o = _lib.MdDa3Outputs(*(t.data_ptr() for t in (out.depth, out.depth_confidence, out.aux, out.aux_confidence,
                                                       out.pose_encoding, out.extrinsics, out.intrinsics)))
        in_kind = _lib.MD_MEM_DEVICE if x.is_cuda else _lib.MD_MEM_HOST
        if _camera is not None:  # camera inputs travel in the memory kind of the image
            e, k = (t.to(torch.float32).to(x.device).contiguous() for t in _camera)
            _lib.check(self._lib.md_da3_infer_with_camera(self._h, C.c_void_p(x.data_ptr()), B, H, W, in_kind, C.c_void_p(e.data_ptr()),
                                                          C.c_void_p(k.data_ptr()), int(e.shape[1]), C.byref(o), _lib.MD_MEM_DEVICE,
                                                          _stream_ptr(self.device.ordinal)))
            return out
        _lib.check(self._lib.md_da3_infer_ex(self._h, C.c_void_p(x.data_ptr()), B, H, W, in_kind, C.byref(o), _lib.MD_MEM_DEVICE,
                                             _stream_ptr(self.device.ordinal)))
        return out

    def infer_into(self, x: torch.Tensor, depth: torch.Tensor, *unused) -> None:
        B, _, H, W = x.shape
        in_kind = _lib.MD_MEM_DEVICE if x.is_cuda else _lib.MD_MEM_HOST
        _lib.check(self._lib.md_da3_infer(self._h, C.c_void_p(x.data_ptr()), B, H, W, in_kind, C.c_void_p(depth.data_ptr()),
                                          _lib.MD_MEM_DEVICE, _stream_ptr(self.device.ordinal)))

    def infer_from_rgb(self, rgb: bytes, width: int, height: int):
        from .inference import rgb_to_input_tensor
        return self.infer(rgb_to_input_tensor(rgb, width, height, self.device))
