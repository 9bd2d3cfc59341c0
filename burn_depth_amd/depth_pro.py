"""Host-side mirror of the reference's Depth Pro interface over the C ABI.

Same names, argument meaning and error behaviour as the reference (file:line under the
reference repository):

* ``DepthPro.new(device, config)``            -- depth_pro/mod.rs:145-191
* ``DepthPro.load(device, path)``             -- depth_pro/mod.rs:193-198
* ``DepthPro.load_with_config``               -- depth_pro/mod.rs:200-208
* ``DepthPro.infer(x) -> DepthProInference``  -- depth_pro/mod.rs:312-364
* ``img_size`` / ``interpolation_method``     -- depth_pro/mod.rs:296,308
* ``decoder_from_features`` / ``head_debug``  -- depth_pro/mod.rs:262-267, 289-307
* debug taps                                  -- encoder.rs:106-123, mod.rs:135-142,285-287

PyTorch is used only as plumbing: device buffers (``torch.empty(..., device="cuda")``), the
current HIP stream, and ``torch.distributed``.  All arithmetic happens in libmi_depth.so.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .config import DepthProConfig, InterpolationMethod, Precision


class Device:
    """`<B as Backend>::Device::default()` (README.md:21): one engine device = one GPU."""

    def __init__(self, ordinal: int = 0):
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.md_device_open(int(ordinal), C.byref(h)))
        self.handle = h
        self.ordinal = int(ordinal)
        torch.cuda.set_device(self.ordinal)

    @staticmethod
    def default() -> "Device":
        return Device(int(os.environ.get("LOCAL_RANK", "0")))

    def synchronize(self) -> None:
        _lib.check(_lib.load().md_device_synchronize(self.handle))

    def close(self) -> None:
        if self.handle:
            _lib.load().md_device_close(self.handle)
            self.handle = None


@dataclass
class DepthProInference:
    """depth_pro/mod.rs:128-133."""
    depth: torch.Tensor           # [B, H, W]
    focallength_px: torch.Tensor  # [B]
    fovx_deg: torch.Tensor        # [B]
    fovy_rad: torch.Tensor        # [B]


@dataclass
class HeadDebug:
    """depth_pro/mod.rs:135-142."""
    conv0: torch.Tensor      # [B, F/2, s, s]
    deconv: torch.Tensor     # [B, F/2, 2s, 2s]
    conv1: torch.Tensor      # [B, 32, 2s, 2s]
    relu: torch.Tensor       # [B, 32, 2s, 2s]
    pre_out: torch.Tensor    # [B, 1, 2s, 2s]
    canonical: torch.Tensor  # [B, 1, 2s, 2s]


def _c_cfg(cfg: DepthProConfig) -> Tuple[_lib.MdDepthProCfg, list]:
    keep = [cfg.patch_encoder_preset.encode(), cfg.image_encoder_preset.encode(),
            cfg.fov_encoder_preset.encode() if cfg.fov_encoder_preset else None]
    c = _lib.MdDepthProCfg(keep[0], keep[1], keep[2], int(cfg.decoder_features), int(bool(cfg.use_fov_head)),
                           int(cfg.interpolation), int(cfg.precision), int(cfg.max_batch), float(cfg.ln_eps))
    return c, keep


def _stream_ptr(device_ordinal: int) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device_ordinal).cuda_stream)


class DepthPro:
    def __init__(self, device: Device, handle: C.c_void_p, config: DepthProConfig):
        self.device = device
        self._h = handle
        self.config = config
        self._lib = _lib.load()
        self._parent: Optional["DepthPro"] = None  # the root model of a fork (keeps it alive)
        self._forks = weakref.WeakSet()            # live forks of a root

    # ---- construction ---------------------------------------------------------------------
    @staticmethod
    def new(device: Device, config: Optional[DepthProConfig] = None, seed: int = 0, init_scheme: int = 0) -> "DepthPro":
        config = config or DepthProConfig()
        c, keep = _c_cfg(config)
        h = C.c_void_p()
        _lib.check(_lib.load().md_depth_pro_create(device.handle, C.byref(c), C.c_uint64(seed), int(init_scheme), C.byref(h)))
        return DepthPro(device, h, config)

    @staticmethod
    def load(device: Device, checkpoint_path: str) -> "DepthPro":
        return DepthPro.load_with_config(device, DepthProConfig(), checkpoint_path)

    @staticmethod
    def load_with_config(device: Device, config: DepthProConfig, checkpoint_path: str) -> "DepthPro":
        c, keep = _c_cfg(config)
        h = C.c_void_p()
        _lib.check(_lib.load().md_depth_pro_load_with_config(device.handle, C.byref(c), os.fspath(checkpoint_path).encode(),
                                                             C.byref(h)))
        return DepthPro(device, h, config)

    def fork(self) -> "DepthPro":
        """`model.clone()` / sharing `&DepthPro` across threads (depth_pro/mod.rs:119-126,312): a second inference
        context (own workspace, own default stream) on the SAME device weights (md_model_fork). Destroy forks before
        the model they were forked from."""
        h = C.c_void_p()
        _lib.check(self._lib.md_model_fork(self._h, C.byref(h)))
        f = type(self)(self.device, h, self.config)
        # the fork aliases this model's weight arenas: keep the root alive until every fork is gone, and let the root
        # find its live forks when it is destroyed first
        f._parent = self._parent if self._parent is not None else self
        f._parent._forks.add(f)
        return f

    def destroy(self) -> None:
        """md_model_destroy. A root with live forks refuses (MD_ERR_INVALID_ARG): they alias its weights and go first."""
        if not self._h:
            return
        _lib.check(self._lib.md_model_destroy(self._h))
        self._h = None
        if self._parent is not None:
            self._parent._forks.discard(self)
            self._parent = None

    def __del__(self):
        try:
            for f in list(self._forks):  # only reachable at interpreter shutdown: a live fork holds a reference to its root
                f.destroy()
            self.destroy()
        except Exception as e:  # noqa: BLE001 -- at interpreter shutdown module globals (even `_lib.MdError`) may already be None
            if _lib is not None and getattr(_lib, "MdError", None) is not None and isinstance(e, _lib.MdError):
                import warnings  # never silent: a failed destroy leaks the weight / workspace arenas
                warnings.warn(f"DepthPro.__del__: {e}", ResourceWarning)

    # ---- introspection --------------------------------------------------------------------
    def query(self, key: str) -> int:
        v = C.c_int64()
        _lib.check(self._lib.md_model_query(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    def set_option(self, key: str, value: int) -> None:
        """md_model_set_option: "batch_invariant" = 1 makes a Depth-Anything-v3 model's 16-bit results independent of the batch an image
        sits in (no launch-size-dependent kernel form), as the reference's `infer` is a pure batch map."""
        _lib.check(self._lib.md_model_set_option(self._h, key.encode(), C.c_int64(int(value))))

    def img_size(self) -> int:
        return self.query("img_size")

    def interpolation_method(self) -> int:
        return self.query("interpolation")

    def param_names(self) -> List[Tuple[str, int]]:
        out = []
        for i in range(self._lib.md_model_param_count(self._h)):
            name, n = C.c_char_p(), C.c_size_t()
            _lib.check(self._lib.md_model_param_info(self._h, i, C.byref(name), C.byref(n)))
            out.append((name.value.decode(), int(n.value)))
        return out

    # ---- records (Module::into_record / load_record, src/lib.rs:163-177) -------------------
    def get_tensor(self, name: str, count: int) -> np.ndarray:
        buf = np.empty(count, dtype=np.float32)
        _lib.check(self._lib.md_model_get_tensor(self._h, name.encode(), buf.ctypes.data_as(C.c_void_p), count))
        return buf

    def set_tensor(self, name: str, values: np.ndarray) -> None:
        v = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
        _lib.check(self._lib.md_model_set_tensor(self._h, name.encode(), v.ctypes.data_as(C.c_void_p), v.size))

    def commit_weights(self) -> None:
        _lib.check(self._lib.md_model_commit_weights(self._h))

    def round_weights_to_f16(self) -> "DepthPro":
        """Round every parameter to the nearest IEEE half, in place, and commit: what `DepthPro::load` of the reference's
        f16 record (`HalfPrecisionSettings`, depth_pro/mod.rs:193-208) of these weights holds. In `Precision.F16X2` the
        weights are then exact MFMA operands (`query("weight_terms") == 2`)."""
        _lib.check(self._lib.md_model_round_weights_f16(self._h))
        return self

    def into_record(self) -> Dict[str, np.ndarray]:
        return {n: self.get_tensor(n, c) for n, c in self.param_names()}

    def load_record(self, record: Dict[str, np.ndarray]) -> "DepthPro":
        for n, _ in self.param_names():
            self.set_tensor(n, record[n])
        self.commit_weights()
        return self

    def weight_arena(self) -> Tuple[int, int]:
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(self._lib.md_model_weight_arena(self._h, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    # ---- inference ------------------------------------------------------------------------
    def infer(self, x: torch.Tensor) -> DepthProInference:
        """x: [B,3,H,W] fp32, ImageNet-normalised (any H, W), on the GPU or the host."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise _lib.MdError(_lib.MD_ERR_SHAPE, f"expected [B,3,H,W], got {tuple(x.shape)}")
        x = x.contiguous().to(torch.float32)
        B, _, H, W = x.shape
        dev = torch.device("cuda", self.device.ordinal)
        depth = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        focal = torch.empty((B,), dtype=torch.float32, device=dev)
        fovx = torch.empty((B,), dtype=torch.float32, device=dev)
        fovy = torch.empty((B,), dtype=torch.float32, device=dev)
        in_kind = _lib.MD_MEM_DEVICE if x.is_cuda else _lib.MD_MEM_HOST
        _lib.check(self._lib.md_depth_pro_infer(self._h, C.c_void_p(x.data_ptr()), B, H, W, in_kind,
                                                C.c_void_p(depth.data_ptr()), C.c_void_p(focal.data_ptr()),
                                                C.c_void_p(fovx.data_ptr()), C.c_void_p(fovy.data_ptr()),
                                                _lib.MD_MEM_DEVICE, _stream_ptr(self.device.ordinal)))
        return DepthProInference(depth, focal, fovx, fovy)

    def infer_into(self, x: torch.Tensor, depth: torch.Tensor, focal: torch.Tensor, fovx: torch.Tensor,
                   fovy: torch.Tensor) -> None:
        """Allocation-free variant used by the benchmark loop (all tensors on this GPU)."""
        B, _, H, W = x.shape
        _lib.check(self._lib.md_depth_pro_infer(self._h, C.c_void_p(x.data_ptr()), B, H, W, _lib.MD_MEM_DEVICE,
                                                C.c_void_p(depth.data_ptr()), C.c_void_p(focal.data_ptr()),
                                                C.c_void_p(fovx.data_ptr()), C.c_void_p(fovy.data_ptr()),
                                                _lib.MD_MEM_DEVICE, _stream_ptr(self.device.ordinal)))

    def infer_windows(self, x: torch.Tensor, parts: int, timings: bool = False):
        """`infer` with the ViT stage run as `parts` consecutive windows of its 37 B sequences on this GPU -- the launches the
        ranks of the tile-parallel mode (`NativeComm.infer_tiles`) issue, without the exchange. Bit-identical to `infer`.
        With `timings`, also returns (window_ms list, tail_ms): GPU milliseconds of each window and of everything behind the
        ViT stage."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise _lib.MdError(_lib.MD_ERR_SHAPE, f"expected [B,3,H,W], got {tuple(x.shape)}")
        x = x.contiguous().to(torch.float32)
        B, _, H, W = x.shape
        dev = torch.device("cuda", self.device.ordinal)
        depth = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        focal, fovx, fovy = (torch.empty((B,), dtype=torch.float32, device=dev) for _ in range(3))
        wms = (C.c_float * max(int(parts), 1))()
        tms = C.c_float()
        in_kind = _lib.MD_MEM_DEVICE if x.is_cuda else _lib.MD_MEM_HOST
        _lib.check(self._lib.md_depth_pro_infer_windows(self._h, C.c_void_p(x.data_ptr()), B, H, W, in_kind,
                                                        C.c_void_p(depth.data_ptr()), C.c_void_p(focal.data_ptr()),
                                                        C.c_void_p(fovx.data_ptr()), C.c_void_p(fovy.data_ptr()), _lib.MD_MEM_DEVICE,
                                                        int(parts), wms if timings else None, C.byref(tms) if timings else None,
                                                        _stream_ptr(self.device.ordinal)))
        out = DepthProInference(depth, focal, fovx, fovy)
        return (out, [float(v) for v in wms], float(tms.value)) if timings else out

    # ---- the decoder / the head alone on caller tensors (depth_pro/mod.rs:262-307) --------------
    def decoder_level_shapes(self) -> List[Tuple[int, int]]:
        """(channels, size) of the encoder feature each decoder level takes, finest first (encoder.rs:416-434)."""
        return [(self.query(f"decoder_level{l}_channels"), self.query(f"decoder_level{l}_size"))
                for l in range(self.query("decoder_levels"))]

    def _view(self, t: torch.Tensor) -> Tuple[torch.Tensor, "_lib.MdNchwView"]:
        if t.dim() != 4:
            raise _lib.MdError(_lib.MD_ERR_SHAPE, f"expected [B,C,H,W], got {tuple(t.shape)}")
        t = t.contiguous().to(torch.float32)
        return t, _lib.MdNchwView(C.c_void_p(t.data_ptr()), int(t.shape[1]), int(t.shape[2]), int(t.shape[3]))

    def decoder_from_features(self, features: List[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor, List[torch.Tensor]]:
        """`DepthPro::decoder_from_features(&self, features)` (depth_pro/mod.rs:262-267): the decoder alone on the caller's
        encoder features (finest first; all on this GPU or all on the host) -> (features, lowres_features, fusion_outputs),
        fusion_outputs[0] the finest. A wrong level count raises MdError(MD_ERR_LEVELS) where the reference panics
        (decoder.rs:200-205), a wrong shape MdError(MD_ERR_SHAPE)."""
        if len(features) == 0:
            raise _lib.MdError(_lib.MD_ERR_LEVELS, "Got encoder output levels = 0")
        B = int(features[0].shape[0])
        if any(int(f.shape[0]) != B or f.is_cuda != features[0].is_cuda for f in features):
            raise _lib.MdError(_lib.MD_ERR_SHAPE, "features differ in batch size or memory kind")
        keep, views = zip(*[self._view(f) for f in features])
        arr = (_lib.MdNchwView * len(views))(*views)
        dev = torch.device("cuda", self.device.ordinal)
        F = self.query("decoder_features")
        shapes = self.decoder_level_shapes()
        s0, s_last = shapes[0][1], shapes[-1][1]
        out_feat = torch.empty((B, F, s0, s0), dtype=torch.float32, device=dev)
        out_low = torch.empty((B, F, s_last, s_last), dtype=torch.float32, device=dev)
        fus = [torch.empty((B, F, s0 if l == 0 else 2 * shapes[l][1], s0 if l == 0 else 2 * shapes[l][1]), dtype=torch.float32, device=dev)
               for l in range(len(shapes))]
        fptr = (C.c_void_p * len(fus))(*[C.c_void_p(t.data_ptr()) for t in fus])
        in_kind = _lib.MD_MEM_DEVICE if features[0].is_cuda else _lib.MD_MEM_HOST
        _lib.check(self._lib.md_depth_pro_decoder_from_features(self._h, arr, len(views), B, in_kind, C.c_void_p(out_feat.data_ptr()),
                                                                C.c_void_p(out_low.data_ptr()), fptr, _lib.MD_MEM_DEVICE,
                                                                _stream_ptr(self.device.ordinal)))
        del keep
        return out_feat, out_low, fus

    def head_debug(self, feature: torch.Tensor) -> HeadDebug:
        """`DepthPro::head_debug(&self, feature)` (depth_pro/mod.rs:289-307): the depth head layer by layer on the caller's
        decoder feature [B, F, s, s]."""
        t, view = self._view(feature)
        B, _, s, _ = t.shape
        dev = torch.device("cuda", self.device.ordinal)
        F2 = self.query("decoder_features") // 2
        mk = lambda c, hw: torch.empty((B, c, hw, hw), dtype=torch.float32, device=dev)  # noqa: E731
        hd = HeadDebug(mk(F2, s), mk(F2, 2 * s), mk(32, 2 * s), mk(32, 2 * s), mk(1, 2 * s), mk(1, 2 * s))
        out = _lib.MdHeadDebug(*[C.c_void_p(x.data_ptr()) for x in (hd.conv0, hd.deconv, hd.conv1, hd.relu, hd.pre_out, hd.canonical)])
        in_kind = _lib.MD_MEM_DEVICE if t.is_cuda else _lib.MD_MEM_HOST
        _lib.check(self._lib.md_depth_pro_head_debug(self._h, C.byref(view), int(B), in_kind, C.byref(out), _lib.MD_MEM_DEVICE,
                                                     _stream_ptr(self.device.ordinal)))
        return hd

    @staticmethod
    def infer_tiles_loopback(contexts: List["DepthPro"], x: torch.Tensor, root: int = 0) -> DepthProInference:
        """The tile-parallel single-image call (`NativeComm.infer_tiles`) with a loopback transport: `contexts` (a model and its forks)
        stand for the ranks, each runs its window of the ViT stage in its own workspace, the root copies the other windows' tokens and
        hook rows where RCCL would deliver them (md_depth_pro_infer_tiles_loopback). Bit-identical to `infer`; a one-GPU test entry."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise _lib.MdError(_lib.MD_ERR_SHAPE, f"expected [B,3,H,W], got {tuple(x.shape)}")
        x = x.contiguous().to(torch.float32)
        B, _, H, W = x.shape
        me = contexts[root]
        dev = torch.device("cuda", me.device.ordinal)
        depth = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        focal, fovx, fovy = (torch.empty((B,), dtype=torch.float32, device=dev) for _ in range(3))
        arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
        in_kind = _lib.MD_MEM_DEVICE if x.is_cuda else _lib.MD_MEM_HOST
        _lib.check(me._lib.md_depth_pro_infer_tiles_loopback(arr, len(contexts), int(root), C.c_void_p(x.data_ptr()), B, H, W, in_kind,
                                                             C.c_void_p(depth.data_ptr()), C.c_void_p(focal.data_ptr()), C.c_void_p(fovx.data_ptr()),
                                                             C.c_void_p(fovy.data_ptr()), _lib.MD_MEM_DEVICE, _stream_ptr(me.device.ordinal)))
        return DepthProInference(depth, focal, fovx, fovy)

    def infer_from_rgb(self, rgb: bytes, width: int, height: int) -> DepthProInference:
        """`infer_from_rgb` (src/inference.rs:128-137); raises MdError(MD_ERR_SHAPE) on a bad length."""
        dev = torch.device("cuda", self.device.ordinal)
        depth = torch.empty((1, height, width), dtype=torch.float32, device=dev) if width > 0 and height > 0 else None
        focal = torch.empty((1,), dtype=torch.float32, device=dev)
        fovy = torch.empty((1,), dtype=torch.float32, device=dev)
        buf = (C.c_uint8 * len(rgb)).from_buffer_copy(rgb) if len(rgb) else (C.c_uint8 * 1)()
        _lib.check(self._lib.md_infer_from_rgb(self._h, C.cast(buf, C.c_void_p), len(rgb), int(width), int(height),
                                               _lib.MD_MEM_HOST, C.c_void_p(depth.data_ptr() if depth is not None else 0),
                                               C.c_void_p(focal.data_ptr()), C.c_void_p(fovy.data_ptr()),
                                               _lib.MD_MEM_DEVICE, _stream_ptr(self.device.ordinal)))
        return DepthProInference(depth, focal, torch.empty(0), fovy)

    # ---- debug taps / timing --------------------------------------------------------------
    def enable_taps(self, enable: bool = True) -> None:
        _lib.check(self._lib.md_model_enable_taps(self._h, int(enable)))

    def read_tap(self, name: str) -> np.ndarray:
        dims = (C.c_int64 * 4)()
        _lib.check(self._lib.md_model_read_tap(self._h, name.encode(), None, 0, C.byref(dims)))
        shape = [int(d) for d in dims]
        n = int(np.prod([max(d, 1) for d in shape]))
        out = np.empty(n, dtype=np.float32)
        _lib.check(self._lib.md_model_read_tap(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), n, C.byref(dims)))
        return out.reshape([max(d, 1) for d in shape])

    def enable_timing(self, enable: bool = True) -> None:
        _lib.check(self._lib.md_model_enable_timing(self._h, int(enable)))

    def set_timing_filter(self, family: Optional[str] = None) -> None:
        """Time only the launches of one kernel family (None = all): md_model_set_timing_filter."""
        _lib.check(self._lib.md_model_set_timing_filter(self._h, family.encode() if family else None))

    def enable_graph(self, enable: bool = True) -> None:
        """Replay the launch schedule from a hipGraph for repeated calls with the same buffers (md_model_enable_graph)."""
        _lib.check(self._lib.md_model_enable_graph(self._h, int(enable)))

    def read_launch_order(self) -> List[str]:
        n = C.c_int()
        _lib.check(self._lib.md_model_read_launch_order(self._h, None, 0, C.byref(n)))
        names = (C.c_char_p * max(n.value, 1))()
        _lib.check(self._lib.md_model_read_launch_order(self._h, names, n.value, C.byref(n)))
        return [names[i].decode() for i in range(n.value)]

    def read_timing(self) -> Dict[str, Tuple[float, int]]:
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        calls = (C.c_int * cap)()
        n = C.c_int()
        _lib.check(self._lib.md_model_read_timing(self._h, names, ms, calls, cap, C.byref(n)))
        return {names[i].decode(): (float(ms[i]), int(calls[i])) for i in range(min(n.value, cap))}
