"""CPU oracle for Depth-Anything-v3 `metric_large` (mono DPT head) -- TEST INFRASTRUCTURE ONLY.

fp32 PyTorch-CPU restatement of `DepthAnything3::infer` for the mono-head variant
(/root/reference/src/model/depth_anything3/mod.rs:288-291,495-624 and dpt.rs:515-731,784-932,
1227-1301; interpolate.rs:7-47).

PARITY STATUS: the backbone (`burn_dino` 0.6.0 `forward_with_intermediate_tokens_ext`) is un-vendored and
no value-level test of the reference touches it => **parity unpinned** for the ViT and for the exact
contents of `DinoIntermediate.patches` (restated here as: final-LayerNorm'ed block output with the cls
token dropped, burn_dino's default `normalize_intermediate_tokens = true` -- the Depth Pro path sets it to
false explicitly, layers/vit.rs:63). The head follows the reference line by line; Burn's
`Interpolate2d(Linear)` = align_corners=True is pinned by depth_pro/interpolate.rs:193-202,231.
Only square inputs at the configured image size are covered (no pos-embed interpolation yet).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from burn_depth_amd.config import DepthAnything3Config
from oracle.depth_pro_ref import identity, resize_align_corners_true, vit_forward

Tensor = torch.Tensor
TOKEN_NORM_EPS = 1e-5     # dpt.rs:768
POS_EMBED_RATIO = 0.1     # dpt.rs:769
POS_EMBED_OMEGA0 = 100.0  # dpt.rs:770


def resize_bilinear(x: Tensor, out_hw: Sequence[int]) -> Tensor:
    """depth_anything3/interpolate.rs:7-47: identity if same size, else Burn Linear (align_corners=True)."""
    if tuple(x.shape[2:]) == tuple(out_hw):
        return x
    return resize_align_corners_true(x, out_hw)


def make_sincos_embedding(dim: int, position: np.float32) -> np.ndarray:
    """dpt.rs:900-932 (f32 arithmetic)."""
    if dim == 0:
        return np.zeros(0, np.float32)
    half = dim // 2
    out = []
    for i in range(half):
        exponent = np.float32(i) / np.float32(half) if half > 0 else np.float32(0)
        omega = np.float32(POS_EMBED_OMEGA0) ** np.float32(-exponent)
        out.append(np.sin(np.float32(position * omega), dtype=np.float32))
    remaining = dim - half
    for i in range(remaining):
        exponent = np.float32(i) / np.float32(remaining) if remaining > 0 else np.float32(0)
        omega = np.float32(POS_EMBED_OMEGA0) ** np.float32(-exponent)
        out.append(np.cos(np.float32(position * omega), dtype=np.float32))
    return np.array(out, dtype=np.float32)


def linspace(start, end, steps):
    """dpt.rs:892-898."""
    if steps <= 1:
        return [np.float32(start)]
    step = np.float32(np.float32(end) - np.float32(start)) / np.float32(steps - 1)
    return [np.float32(np.float32(start) + np.float32(step * np.float32(i))) for i in range(steps)]


def build_positional_embedding(channels: int, height: int, width: int, image_width: int, image_height: int) -> np.ndarray:
    """dpt.rs:835-890, including the transposed pixel index `x_idx*height + y_idx` (dpt.rs:879). Returns the
    flat buffer the reference reshapes to [1, C, height, width]."""
    f = np.float32
    aspect = f(image_width) / f(image_height)
    diag = f(np.sqrt(f(aspect * aspect + f(1.0))))
    span_x, span_y = f(aspect / diag), f(f(1.0) / diag)
    left_x = f(-span_x * f(f(width) - f(1.0)) / f(width))
    right_x = f(span_x * f(f(width) - f(1.0)) / f(width))
    top_y = f(-span_y * f(f(height) - f(1.0)) / f(height))
    bottom_y = f(span_y * f(f(height) - f(1.0)) / f(height))
    xs, ys = linspace(left_x, right_x, width), linspace(top_y, bottom_y, height)
    x_ch = channels // 2
    y_ch = channels - x_ch
    ex = np.stack([make_sincos_embedding(x_ch, x) for x in xs])  # [width, x_ch]
    ey = np.stack([make_sincos_embedding(y_ch, y) for y in ys])  # [height, y_ch]
    chw = np.zeros((channels, height * width), np.float32)
    xi, yi = np.meshgrid(np.arange(width), np.arange(height), indexing="ij")
    pix = (xi * height + yi).reshape(-1)
    chw[:x_ch, pix] = ex[xi.reshape(-1)].T
    chw[x_ch:, pix] = ey[yi.reshape(-1)].T
    return chw.reshape(-1)


def pos_embed_add(x: Tensor, image_width: int, image_height: int) -> Tensor:
    """PosEmbedCache::add (dpt.rs:799-828)."""
    _, c, h, w = x.shape
    table = torch.from_numpy(build_positional_embedding(c, h, w, image_width, image_height)).reshape(1, c, h, w)
    return x + table * POS_EMBED_RATIO


def _rcu(x: Tensor, W, name: str, q, extra=None) -> Tensor:
    """ResidualConvUnit::forward (dpt.rs:1248-1252) (+ the fusion add folded in, as the engine does)."""
    t = q(F.relu(F.conv2d(q(F.relu(x)), q(W[f"{name}.conv1.weight"]), W[f"{name}.conv1.bias"], padding=1)))
    t = F.conv2d(t, q(W[f"{name}.conv2.weight"]), W[f"{name}.conv2.bias"], padding=1) + x
    if extra is not None:
        t = extra + t
    return q(t)


def _ffb(top: Tensor, lateral, size, W, name: str, q) -> Tensor:
    """FeatureFusionBlock::forward (dpt.rs:1206-1222)."""
    y = top
    if lateral is not None and f"{name}.residual1.conv1.weight" in W:
        y = _rcu(lateral, W, f"{name}.residual1", q, extra=top)
    y = _rcu(y, W, f"{name}.residual2", q)
    target = size if size is not None else (y.shape[2] * 2, y.shape[3] * 2)
    y = q(resize_bilinear(y, target))
    return q(F.conv2d(y, q(W[f"{name}.out_conv.weight"]), W[f"{name}.out_conv.bias"]))


def head_forward_raw(hooks: List[Tensor], height: int, width: int, W, cfg: DepthAnything3Config, q=identity, debug=None) -> Tensor:
    """DepthAnything3Head::forward_raw (dpt.rs:587-631) with patch_start_idx = 0 (mod.rs:540)."""
    ps = cfg.patch_size
    ph, pw = height // ps, width // ps
    feats = []
    for s in range(4):
        x = hooks[s][:, :ph * pw]
        var, mean = torch.var_mean(x, dim=2, unbiased=False, keepdim=True)     # apply_token_norm, dpt.rs:761-766
        x = q((x - mean) / torch.sqrt(var + TOKEN_NORM_EPS))
        x = x.permute(0, 2, 1).reshape(x.shape[0], -1, ph, pw)
        x = F.conv2d(x, q(W[f"head_mono.projects.{s}.weight"]), W[f"head_mono.projects.{s}.bias"])
        if cfg.pos_embed:
            x = pos_embed_add(x, width, height)
        x = q(x)
        if s == 0:
            x = q(F.conv_transpose2d(x, q(W["head_mono.resize_layers.0.conv_t.weight"]), W["head_mono.resize_layers.0.conv_t.bias"], stride=4))
        elif s == 1:
            x = q(F.conv_transpose2d(x, q(W["head_mono.resize_layers.1.conv_t.weight"]), W["head_mono.resize_layers.1.conv_t.bias"], stride=2))
        elif s == 3:
            x = q(F.conv2d(x, q(W["head_mono.resize_layers.3.conv.weight"]), W["head_mono.resize_layers.3.conv.bias"], stride=2, padding=1))
        feats.append(x)
    rn = [q(F.conv2d(feats[i], q(W[f"head_mono.scratch.layer{i + 1}_rn.weight"]), padding=1)) for i in range(4)]
    sc = "head_mono.scratch"
    out = _ffb(rn[3], None, rn[2].shape[2:], W, f"{sc}.refinenet4", q)
    out = _ffb(out, rn[2], rn[1].shape[2:], W, f"{sc}.refinenet3", q)
    out = _ffb(out, rn[1], rn[0].shape[2:], W, f"{sc}.refinenet2", q)
    out = _ffb(out, rn[0], None, W, f"{sc}.refinenet1", q)
    fused = q(F.conv2d(out, q(W[f"{sc}.output_conv1.weight"]), W[f"{sc}.output_conv1.bias"], padding=1))
    fused = resize_bilinear(fused, (ph * ps, pw * ps))
    if cfg.pos_embed:
        fused = pos_embed_add(fused, width, height)
    fused = q(fused)
    # ConvStack (dpt.rs:1287-1290); the engine fuses conv1+relu+conv2+exp in one fp32 epilogue
    t = F.relu(F.conv2d(fused, q(W[f"{sc}.output_conv2.conv1.weight"]), W[f"{sc}.output_conv2.conv1.bias"], padding=1))
    logits = F.conv2d(t, W[f"{sc}.output_conv2.conv2.weight"], W[f"{sc}.output_conv2.conv2.bias"])
    if debug is not None:
        debug.update(stage_feats=feats, rn=rn, fused=fused, logits=logits)
    return torch.exp(logits)  # HeadActivation::Exp (dpt.rs:700)


def backbone_hooks(x: Tensor, W, cfg: DepthAnything3Config, q=identity) -> List[Tensor]:
    """Backbone::forward_with_hooks (mod.rs:202-215): per hook block the final-norm'ed patch tokens."""
    v = cfg.vit()
    _, raw = vit_forward(x, W, "backbone.pretrained", v, cfg.hook_block_ids, q)
    g, b = W["backbone.pretrained.norm.gamma"], W["backbone.pretrained.norm.beta"]
    return [F.layer_norm(h, (v.embed_dim,), g, b, v.ln_eps)[:, 1:] for h in raw]


def infer(x: Tensor, W, cfg: DepthAnything3Config, q=identity, debug: bool = False):
    """DepthAnything3::infer (mod.rs:288-291 -> 495-564 -> 587-609): depth [B,H,W]."""
    B, _, H, Wd = x.shape
    ps = cfg.patch_size
    if H % ps or Wd % ps:  # mod.rs:509-520 (panic)
        raise ValueError(f"Input {H}x{Wd} must be divisible by patch size {ps}")
    hooks = backbone_hooks(x, W, cfg, q)
    dbg = {} if debug else None
    act = head_forward_raw(hooks, H, Wd, W, cfg, q, dbg)
    out = dict(depth=act[:, 0])  # select_depth_channel (dpt.rs:633-647)
    if debug:
        dbg["hooks"] = hooks
        out["debug"] = dbg
    return out
