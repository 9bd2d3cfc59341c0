"""Stand-alone operators of libmi_depth.so on torch device tensors (used by the parity tests and
bench.py; each wraps one ``md_op_*`` entry point of include/mi_depth.h)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from .depth_pro import Device, _stream_ptr


def _p(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    assert t.is_cuda, "operators take device tensors"
    return t.contiguous().to(torch.float32)


def resize_bilinear(dev: Device, x: torch.Tensor, out_hw: Tuple[int, int], method: int = 0) -> torch.Tensor:
    x = _f32c(x)
    B, Cn, H, W = x.shape
    out = torch.empty((B, Cn, int(out_hw[0]), int(out_hw[1])), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().md_op_resize_bilinear(dev.handle, _p(x), B, Cn, H, W, _p(out), int(out_hw[0]), int(out_hw[1]),
                                                 int(method), _stream_ptr(dev.ordinal)))
    return out


def resize_bilinear_into(dev: Device, x: torch.Tensor, out: torch.Tensor, method: int = 0) -> None:
    """Allocation-free form (bench.py): fp32 NCHW x -> out, both contiguous device tensors."""
    B, Cn, H, W = x.shape
    _lib.check(_lib.load().md_op_resize_bilinear(dev.handle, _p(x), B, Cn, H, W, _p(out), int(out.shape[2]), int(out.shape[3]),
                                                 int(method), _stream_ptr(dev.ordinal)))


_NHWC_PREC = {torch.bfloat16: 0, torch.float32: 1, torch.float16: 3}


def resize_nhwc_into(dev: Device, x: torch.Tensor, out: torch.Tensor, method: int = 1) -> None:
    """NHWC bilinear resize of the Depth-Anything-v3 head (md_op_resize_nhwc): x [B,H,W,C] -> out [B,OH,OW,C]."""
    assert x.is_cuda and out.is_cuda and x.is_contiguous() and out.is_contiguous() and x.dtype == out.dtype
    B, H, W, Cn = x.shape
    _lib.check(_lib.load().md_op_resize_nhwc(dev.handle, _p(x), B, H, W, Cn, _p(out), int(out.shape[1]), int(out.shape[2]), int(method),
                                             _NHWC_PREC[x.dtype], _stream_ptr(dev.ordinal)))


def resize_nhwc(dev: Device, x: torch.Tensor, out_hw: Tuple[int, int], method: int = 1) -> torch.Tensor:
    out = torch.empty((x.shape[0], int(out_hw[0]), int(out_hw[1]), x.shape[3]), dtype=x.dtype, device=x.device)
    resize_nhwc_into(dev, x.contiguous(), out, method)
    return out


_PREC_DTYPE = {0: torch.bfloat16, 1: torch.float32, 3: torch.float16}


def pyramid_patchify(dev: Device, x: torch.Tensor, window: int, patch: int = 16, method: int = 0, precision: int = 1,
                     force_generic: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Pyramid + sliding-window split + patch extraction of DepthProEncoder::forward (encoder.rs:326-344) as the A matrix
    of the patch-embed GEMM (md_op_pyramid_patchify): x [B,3,S,S] fp32, S = 4 * window -> [35B * (window/patch)^2, 3 * patch^2]."""
    x = _f32c(x)
    B, _, S, _ = x.shape
    rows, cols = C.c_int(), C.c_int()
    lib = _lib.load()
    _lib.check(lib.md_op_pyramid_patchify(dev.handle, None, B, S, window, patch, method, precision, 0, None, C.byref(rows), C.byref(cols), None))
    if out is None:
        out = torch.empty((rows.value, cols.value), dtype=_PREC_DTYPE[precision], device=x.device)
    _lib.check(lib.md_op_pyramid_patchify(dev.handle, _p(x), B, S, window, patch, method, precision, int(force_generic), _p(out),
                                          C.byref(rows), C.byref(cols), _stream_ptr(dev.ordinal)))
    return out


def resize_output_size(H: int, W: int, scale: Tuple[float, float]) -> Tuple[int, int]:
    oh, ow = C.c_int(), C.c_int()
    _lib.check(_lib.load().md_op_resize_output_size(H, W, C.c_float(scale[0]), C.c_float(scale[1]), C.byref(oh), C.byref(ow)))
    return oh.value, ow.value


def resize_bilinear_scale(dev: Device, x: torch.Tensor, scale: Tuple[float, float], method: int = 0) -> torch.Tensor:
    return resize_bilinear(dev, x, resize_output_size(x.shape[2], x.shape[3], scale), method)


def split(dev: Device, x: torch.Tensor, window: int, overlap: float) -> Tuple[torch.Tensor, int]:
    x = _f32c(x)
    B, Cn, S, _ = x.shape
    steps = C.c_int()
    _lib.check(_lib.load().md_op_split(dev.handle, _p(x), B, Cn, S, window, C.c_float(overlap), C.c_void_p(0), C.byref(steps), None))
    out = torch.empty((steps.value * steps.value * B, Cn, window, window), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().md_op_split(dev.handle, _p(x), B, Cn, S, window, C.c_float(overlap), _p(out), C.byref(steps),
                                       _stream_ptr(dev.ordinal)))
    return out, steps.value


def merge(dev: Device, tiles: torch.Tensor, batch: int, padding: int) -> torch.Tensor:
    tiles = _f32c(tiles)
    n, Cn, h, w = tiles.shape
    oh, ow = C.c_int(), C.c_int()
    _lib.check(_lib.load().md_op_merge(dev.handle, C.c_void_p(0), n, Cn, h, w, batch, padding, C.c_void_p(0), C.byref(oh), C.byref(ow), None))
    out = torch.empty((batch, Cn, oh.value, ow.value), dtype=torch.float32, device=tiles.device)
    _lib.check(_lib.load().md_op_merge(dev.handle, _p(tiles), n, Cn, h, w, batch, padding, _p(out), C.byref(oh), C.byref(ow),
                                       _stream_ptr(dev.ordinal)))
    return out


def layernorm(dev: Device, x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float) -> torch.Tensor:
    x = _f32c(x)
    rows, D = x.shape
    out = torch.empty_like(x)
    _lib.check(_lib.load().md_op_layernorm(dev.handle, _p(x), _p(gamma), _p(beta), rows, D, C.c_float(eps), _p(out),
                                           _stream_ptr(dev.ordinal)))
    return out


STORAGE_OUT = 0x100  # MD_OP_STORAGE_OUT: result through the engine's storage type (bf16) instead of the fp32 store


def linear(dev: Device, x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0, precision: int = 0,
           tile: int = _lib.TILE_AUTO, storage_out: bool = False) -> torch.Tensor:
    precision |= STORAGE_OUT if storage_out else 0
    x, w = _f32c(x), _f32c(w)
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().md_op_linear_tile(dev.handle, _p(x), _p(w), _p(bias), M, N, K, act, precision, tile, _p(out),
                                             _stream_ptr(dev.ordinal)))
    return out


def attention(dev: Device, qkv: torch.Tensor, heads: int, precision: int = 0) -> torch.Tensor:
    qkv = _f32c(qkv)
    T, N, _ = qkv.shape
    out = torch.empty((T, N, heads * 64), dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.load().md_op_attention(dev.handle, _p(qkv), T, N, heads, precision, _p(out), _stream_ptr(dev.ordinal)))
    return out


def conv3x3(dev: Device, x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], pre_relu: bool = False,
            precision: int = 0, storage_out: bool = False) -> torch.Tensor:
    precision |= STORAGE_OUT if storage_out else 0
    x, w = _f32c(x), _f32c(w)
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().md_op_conv3x3(dev.handle, _p(x), _p(w), _p(bias), B, Cin, H, W, Cout, int(pre_relu), precision,
                                         _p(out), _stream_ptr(dev.ordinal)))
    return out


def deconv2x2(dev: Device, x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], precision: int = 0,
              storage_out: bool = False) -> torch.Tensor:
    precision |= STORAGE_OUT if storage_out else 0
    x, w = _f32c(x), _f32c(w)
    B, Cin, H, W = x.shape
    Cout = w.shape[1]
    out = torch.empty((B, Cout, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().md_op_deconv2x2(dev.handle, _p(x), _p(w), _p(bias), B, Cin, H, W, Cout, precision, _p(out),
                                           _stream_ptr(dev.ordinal)))
    return out


def conv2d_direct(dev: Device, x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], stride: int, pad: int,
                  relu: bool) -> torch.Tensor:
    x, w = _f32c(x), _f32c(w)
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((B, Cout, max(OH, 0), max(OW, 0)), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().md_op_conv2d_direct(dev.handle, _p(x), _p(w), _p(bias), B, Cin, H, W, Cout, k, stride, pad,
                                               int(relu), _p(out), _stream_ptr(dev.ordinal)))
    return out


def fov_to_focal(fovx_deg: float, H: int, W: int) -> Tuple[float, float]:
    f, y = C.c_float(), C.c_float()
    _lib.check(_lib.load().md_op_fov_to_focal(C.c_float(fovx_deg), H, W, C.byref(f), C.byref(y)))
    return f.value, y.value
