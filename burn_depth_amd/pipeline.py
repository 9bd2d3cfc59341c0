"""Host-side pre/post-processing either side of the hot path, mirroring the reference's CLI flow
`example/inference.rs`: AnyDepthModel::load -> prepare_input_image -> infer_from_rgb -> save_depth_map.

* `AnyDepthModel` / `DepthModelKind`           -- src/model/mod.rs:17-142
* `prepare_depth_anything3_image`             -- src/model/mod.rs:162-210 (shortest-side resize + centre crop). The
  resize itself is `image::imageops::resize(.., FilterType::CatmullRom)` from the un-vendored `image` crate: its
  separable resampler is restated here from the crate's published algorithm (**parity unpinned**: no value-level
  test of the reference touches it).
* `crop_depth_field`, `resize_depth_field`, `sample_depth_bilinear`, the min-max normalisation of `save_depth_map`
                                              -- example/inference.rs:103-273 (restated exactly, fp32)
* `write_gray_png`                            -- the reference uses `image::GrayImage::save`; a stdlib-zlib PNG writer
  stands in (8-bit grayscale, filter 0), `read_gray_png` reads it back for the tests.
JPEG decoding stays out of scope: images come in as uint8 arrays."""
from __future__ import annotations

import enum
import os
import struct
import zlib
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

f32 = np.float32


class DepthModelKind(enum.Enum):
    DEPTH_PRO = "depth-pro"            # src/model/mod.rs:31-36 (`as_str`)
    DEPTH_ANYTHING3 = "depth-anything-3"


@dataclass
class ImageCropRegion:
    x: int
    y: int
    width: int
    height: int


@dataclass
class PreparedModelImage:
    width: int
    height: int
    rgb: np.ndarray                    # uint8 [H, W, 3]
    crop: Optional[ImageCropRegion] = None


# ------------------------------------------------------------------------------------------------------------
# image crate resampling (imageops::sample: vertical_sample then horizontal_sample through an f32 image)
# ------------------------------------------------------------------------------------------------------------
def _catmull_rom(x: np.ndarray) -> np.ndarray:
    """`bc_cubic_spline(x, 0, 0.5)`, support 2."""
    a = np.abs(x).astype(f32)
    b, c = f32(0.0), f32(0.5)
    k = np.where(a < 1, (12 - 9 * b - 6 * c) * a ** 3 + (-18 + 12 * b + 6 * c) * a ** 2 + (6 - 2 * b),
                 np.where(a < 2, (-b - 6 * c) * a ** 3 + (6 * b + 30 * c) * a ** 2 + (-12 * b - 48 * c) * a + (8 * b + 24 * c), 0.0))
    return (k / 6).astype(f32)


def _sample_axis(img: np.ndarray, new_len: int, axis: int) -> np.ndarray:
    """One pass of the separable resampler along `axis` of an f32 [H, W, C] image."""
    n = img.shape[axis]
    ratio = f32(n) / f32(new_len)
    sratio = max(ratio, f32(1.0))
    support = f32(2.0) * sratio
    out_shape = list(img.shape)
    out_shape[axis] = new_len
    out = np.empty(out_shape, f32)
    src = np.moveaxis(img, axis, 0)
    dst = np.moveaxis(out, axis, 0)
    for o in range(new_len):
        centre = (f32(o) + f32(0.5)) * ratio
        left = int(min(max(np.floor(centre - support), 0), n - 1))
        right = int(min(max(np.ceil(centre + support), left + 1), n))
        c = centre - f32(0.5)
        w = _catmull_rom((np.arange(left, right, dtype=f32) - c) / sratio)
        w = (w / w.sum(dtype=f32)).astype(f32)
        dst[o] = np.tensordot(w, src[left:right], axes=(0, 0))
    return out


def resize_catmull_rom(rgb: np.ndarray, new_width: int, new_height: int) -> np.ndarray:
    """`imageops::resize(image, w, h, FilterType::CatmullRom)` on an 8-bit RGB image."""
    if rgb.dtype != np.uint8 or rgb.ndim != 3:
        raise ValueError("expected uint8 [H,W,C]")
    if rgb.shape[1] == new_width and rgb.shape[0] == new_height:
        return rgb.copy()
    tmp = _sample_axis(rgb.astype(f32), new_height, 0)   # vertical pass keeps f32
    out = _sample_axis(tmp, new_width, 1)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)  # clamp + round-to-nearest on the way back to u8


def prepare_depth_anything3_image(rgb: np.ndarray, target: int) -> PreparedModelImage:
    """src/model/mod.rs:162-210."""
    if target == 0:
        raise ValueError("depth_anything3 requires a non-zero target resolution")
    oh, ow = rgb.shape[:2]
    if ow == target and oh == target:
        return PreparedModelImage(target, target, rgb.copy(), None)
    shortest = f32(max(min(ow, oh), 1))
    scale = f32(target) / shortest
    sw = max(int(np.round(f32(ow) * scale)), target)   # f32::round = half away from zero; sizes are never at .5 here
    sh = max(int(np.round(f32(oh) * scale)), target)
    resized = resize_catmull_rom(rgb, sw, sh)
    cx, cy = max(sw - target, 0) // 2, max(sh - target, 0) // 2
    return PreparedModelImage(target, target, np.ascontiguousarray(resized[cy:cy + target, cx:cx + target]), None)


# ------------------------------------------------------------------------------------------------------------
# save_depth_map (example/inference.rs:103-273)
# ------------------------------------------------------------------------------------------------------------
def crop_depth_field(values: np.ndarray, region: ImageCropRegion) -> np.ndarray:
    h, w = values.shape
    if region.x + region.width > w or region.y + region.height > h:
        raise ValueError(f"Crop region {region} exceeds depth tensor bounds {w}x{h}")
    return values[region.y:region.y + region.height, region.x:region.x + region.width].copy()


def resize_depth_field(values: np.ndarray, dst_width: int, dst_height: int) -> np.ndarray:
    """`resize_depth_field` + `sample_depth_bilinear`: half-pixel centres, x0 = clamp(floor(x)), and the fraction
    taken against the CLAMPED x0 (so the border extrapolates, unlike Depth Pro's own resize)."""
    sh, sw = values.shape
    if sw == dst_width and sh == dst_height:
        return values.astype(f32, copy=True)
    v = values.astype(f32)
    sx = f32(sw) / f32(dst_width) if dst_width > 1 else f32(0)
    sy = f32(sh) / f32(dst_height) if dst_height > 1 else f32(0)
    xs = ((np.arange(dst_width, dtype=f32) + f32(0.5)) * sx - f32(0.5)) if dst_width > 1 else np.zeros(dst_width, f32)
    ys = ((np.arange(dst_height, dtype=f32) + f32(0.5)) * sy - f32(0.5)) if dst_height > 1 else np.zeros(dst_height, f32)

    def idx(c, n):
        c0 = np.clip(np.floor(c), 0, n - 1).astype(np.int64)
        c1 = np.clip(c0 + 1, 0, n - 1)
        return c0, c1, (c - c0.astype(f32)).astype(f32)

    x0, x1, fx = idx(xs, sw)
    y0, y1, fy = idx(ys, sh)
    one = f32(1)
    top = v[y0][:, x0] * (one - fx) + v[y0][:, x1] * fx
    bot = v[y1][:, x0] * (one - fx) + v[y1][:, x1] * fx
    return (top * (one - fy)[:, None] + bot * fy[:, None]).astype(f32)


def depth_to_u8(depth: np.ndarray, crop: Optional[ImageCropRegion] = None,
                target_dims: Optional[Tuple[int, int]] = None) -> np.ndarray:
    """The pixel values `save_depth_map` writes: optional crop, optional bilinear restore to (width, height),
    min-max normalisation over the finite values (all non-finite -> range [0,1], non-finite pixels -> 0)."""
    if depth.ndim == 3:
        if depth.shape[0] != 1:
            raise ValueError(f"Example expects batch size of 1, got {depth.shape[0]}.")
        depth = depth[0]
    v = depth.astype(f32)
    if crop is not None:
        v = crop_depth_field(v, crop)
    if target_dims is not None and (target_dims[0] != v.shape[1] or target_dims[1] != v.shape[0]):
        v = resize_depth_field(v, target_dims[0], target_dims[1])
    fin = np.isfinite(v)
    lo, hi = (f32(v[fin].min()), f32(v[fin].max())) if fin.any() else (f32(0), f32(1))
    rng = max(f32(hi - lo), np.finfo(f32).eps)
    norm = np.where(fin, np.clip((v - lo) / rng, 0, 1), 0).astype(f32)
    return np.clip(np.floor(norm * f32(255) + f32(0.5)), 0, 255).astype(np.uint8)  # f32::round on non-negative values


def write_gray_png(path: str, pixels: np.ndarray) -> None:
    if pixels.dtype != np.uint8 or pixels.ndim != 2:
        raise ValueError("expected uint8 [H,W]")
    h, w = pixels.shape

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    raw = b"".join(b"\x00" + pixels[y].tobytes() for y in range(h))
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def read_gray_png(path: str) -> np.ndarray:
    """Reader for the files `write_gray_png` produces (8-bit grayscale, filter type 0 rows)."""
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", 0
    h = 0
    while pos < len(b):
        (n,), tag = struct.unpack(">I", b[pos:pos + 4]), b[pos + 4:pos + 8]
        data = b[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h = struct.unpack(">II", data[:8])
        elif tag == b"IDAT":
            idat += data
        pos += 12 + n
    raw = zlib.decompress(idat)
    rows = np.frombuffer(raw, np.uint8).reshape(h, w + 1)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].copy()


def save_depth_map(depth: np.ndarray, path: str, crop: Optional[ImageCropRegion] = None,
                   target_dims: Optional[Tuple[int, int]] = None) -> np.ndarray:
    px = depth_to_u8(depth, crop, target_dims)
    write_gray_png(path, px)
    return px


# ------------------------------------------------------------------------------------------------------------
# AnyDepthModel (src/model/mod.rs:40-142)
# ------------------------------------------------------------------------------------------------------------
class AnyDepthModel:
    def __init__(self, kind: DepthModelKind, model):
        self.kind, self.model = kind, model

    @staticmethod
    def load(kind: DepthModelKind, device, checkpoint: str, precision=None) -> "AnyDepthModel":
        """`AnyDepthModel::load`: Depth Pro loads directly; Depth-Anything-v3 tries metric_large then small, small
        first when the file name contains "small" (mod.rs:62-100). Errors carry the reference's message prefix."""
        from . import _lib
        from .config import DepthAnything3Config, DepthProConfig
        from .depth_anything3 import DepthAnything3
        from .depth_pro import DepthPro
        if kind == DepthModelKind.DEPTH_PRO:
            cfg = DepthProConfig()
            if precision is not None:
                cfg.precision = precision
            try:
                return AnyDepthModel(kind, DepthPro.load_with_config(device, cfg, checkpoint))
            except _lib.MdError as e:
                raise RuntimeError(f"Failed to load DepthPro checkpoint: {e}") from e
        configs = [DepthAnything3Config.metric_large(), DepthAnything3Config.small()]
        if "small" in os.path.basename(checkpoint).lower():
            configs.reverse()
        last = None
        for cfg in configs:
            if precision is not None:
                cfg.precision = precision
            try:
                return AnyDepthModel(kind, DepthAnything3.load_file(device, cfg, checkpoint))
            except _lib.MdError as e:
                last = e
        raise RuntimeError(f"Failed to load Depth Anything 3 checkpoint `{checkpoint}`: {last}")

    def preferred_input_resolution(self) -> Optional[int]:
        return None if self.kind == DepthModelKind.DEPTH_PRO else self.model.img_size()

    def prepare_input_image(self, rgb: np.ndarray) -> PreparedModelImage:
        if self.kind == DepthModelKind.DEPTH_PRO:
            return PreparedModelImage(rgb.shape[1], rgb.shape[0], rgb.copy(), None)
        return prepare_depth_anything3_image(rgb, self.model.img_size())

    def infer_from_rgb(self, prepared: PreparedModelImage):
        return self.model.infer_from_rgb(prepared.rgb.tobytes(), prepared.width, prepared.height)
