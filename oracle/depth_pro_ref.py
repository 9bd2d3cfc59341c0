"""CPU oracle for the Depth Pro hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A readable fp32 restatement (PyTorch-CPU functional ops) of what the reference computes in
``DepthPro::infer``.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package; the product path (``burn_depth_amd``) never does.

PARITY STATUS:
* The reference (Rust/Burn) can be neither compiled nor imported here, so this oracle is pinned
  by the reference's own known-answer tests, transcribed in ``tests/test_oracle_kats.py``:
  interpolate.rs:166-248, inference.rs:145-181, encoder.rs:501-586, lib.rs:179-195, vit.rs:76-96.
* The ViT arithmetic lives in the un-vendored crate ``burn_dino`` 0.6.0 (Cargo.lock:2130-2133);
  no value-level test in the reference touches it, so ``vit_forward`` restates the public DINOv2
  definition with the switches the reference sets (vit.rs:60-63) and is **parity unpinned**.

Every function cites the reference lines it follows.  ``q`` arguments are operand-quantisers:
identity for the fp32 oracle; ``bf16_round`` / ``f16_round`` reproduce the points at which the bf16 / f16
engine stores MFMA operands, so those GPU paths can be checked against a matched CPU emulation.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from oracle import ref_config
from oracle.ref_config import RefViT as ViTConfig


class InterpolationMethod:
    """depth_pro/interpolate.rs:11-22 (the oracle's own copy: oracle/ref_config.py)"""
    CUSTOM = ref_config.INTERP_CUSTOM
    BURN = ref_config.INTERP_BURN


def _cfg_vits(cfg):
    """The three ViT configurations of a DepthProConfig-like object, resolved BY PRESET NAME through the oracle's own
    tables (oracle/ref_config.py) -- never through the product's `cfg.patch_vit()`."""
    eps = getattr(cfg, "ln_eps", None)
    return (ref_config.vit_for(cfg.patch_encoder_preset, eps), ref_config.vit_for(cfg.image_encoder_preset, eps),
            ref_config.vit_for(cfg.fov_encoder_preset, eps))


Tensor = torch.Tensor
Quant = Callable[[Tensor], Tensor]


def identity(x: Tensor) -> Tensor:
    return x


def bf16_round(x: Tensor) -> Tensor:
    """Round-to-nearest-even to bfloat16 and back (what ``v_cvt_pk_bf16_f32`` does)."""
    return x.to(torch.bfloat16).to(torch.float32)


def f16_round(x: Tensor) -> Tensor:
    """Saturate at +-65504, round-to-nearest-even to IEEE half and back (the engine's MD_PREC_F16 stores:
    ``v_med3_f32`` + ``v_cvt_pk_f16_f32``). The reference's checkpoints are f16 (mod.rs:206), so real weights pass through
    unchanged."""
    return x.clamp(-65504.0, 65504.0).to(torch.float16).to(torch.float32)


def f16x2_round(x: Tensor) -> Tensor:
    """The engine's MD_PREC_F16X2 operand: two IEEE-half planes, hi = f16(x), lo = f16(x - hi); value hi + lo (22 bits)."""
    hi = f16_round(x)
    return hi + f16_round(x - hi)


# ---------------------------------------------------------------------------------------------
# a1  rgb_to_input_tensor  (src/inference.rs:79-121)
# ---------------------------------------------------------------------------------------------
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def rgb_to_input_tensor(rgb: bytes, width: int, height: int) -> Tensor:
    expected = width * height * 3
    if len(rgb) != expected:  # inference.rs:90-95
        raise ValueError(f"expected {expected} RGB bytes for {width}x{height}, got {len(rgb)}")
    px = torch.frombuffer(bytearray(rgb), dtype=torch.uint8).reshape(height * width, 3)
    out = torch.empty(3, height * width, dtype=torch.float32)
    for c in range(3):  # inference.rs:103-111: v/255 then (v-mean)/std, all in f32
        v = px[:, c].to(torch.float32) / torch.tensor(255.0)
        out[c] = (v - torch.tensor(MEAN[c])) / torch.tensor(STD[c])
    return out.reshape(1, 3, height, width)


# ---------------------------------------------------------------------------------------------
# a2  bilinear resize  (depth_pro/interpolate.rs:24-145)
# ---------------------------------------------------------------------------------------------
def compute_output_size(inp: int, scale: float) -> int:
    """interpolate.rs:24-27 (f32 multiply, floor, max 1)."""
    scaled = int(math.floor(float(np.float32(inp) * np.float32(scale))))
    return max(scaled, 1)


def _axis_taps(in_size: int, out_size: int):
    """interpolate.rs:29-41,78-89 for one axis: indices idx0/idx1 and fractional weight d (f32)."""
    scale = torch.tensor(in_size, dtype=torch.float32) / torch.tensor(out_size, dtype=torch.float32)
    o = torch.arange(out_size, dtype=torch.float32)
    src = (o + 0.5) * scale - 0.5
    f0 = torch.floor(src)
    f1 = torch.minimum(f0 + 1.0, torch.tensor(float(in_size - 1)))
    idx0 = torch.clamp(f0, min=0.0).to(torch.int64)
    idx1 = f1.to(torch.int64)
    d = src - f0
    return idx0, idx1, d


def resize_align_corners_false(x: Tensor, out_hw: Sequence[int]) -> Tensor:
    """``InterpolationMethod::Custom`` (interpolate.rs:54-110): PyTorch align_corners=False."""
    _, _, ih, iw = x.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    if ih == oh and iw == ow:
        return x
    assert oh > 0 and ow > 0, "output size must be positive"
    y0, y1, dy = _axis_taps(ih, oh)
    x0, x1, dx = _axis_taps(iw, ow)
    dx = dx.view(1, 1, 1, ow)
    dy = dy.view(1, 1, oh, 1)
    r0 = x.index_select(2, y0)
    r1 = x.index_select(2, y1)
    tl, tr = r0.index_select(3, x0), r0.index_select(3, x1)
    bl, br = r1.index_select(3, x0), r1.index_select(3, x1)
    top = tl * (1.0 - dx) + tr * dx          # interpolate.rs:48
    bottom = bl * (1.0 - dx) + br * dx       # interpolate.rs:49
    return top * (1.0 - dy) + bottom * dy    # interpolate.rs:51


def resize_align_corners_true(x: Tensor, out_hw: Sequence[int]) -> Tensor:
    """``InterpolationMethod::Burn`` = Burn ``module::interpolate`` Bilinear (interpolate.rs:112-121),
    pinned by the KATs at interpolate.rs:193-202,231: src = o*(in-1)/(out-1), out==1 -> src 0."""
    _, _, ih, iw = x.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])

    def taps(i, o):
        oo = torch.arange(o, dtype=torch.float32)
        if o > 1:
            src = oo * (torch.tensor(float(i - 1)) / torch.tensor(float(o - 1)))
        else:
            src = torch.zeros(o)
        f0 = torch.floor(src)
        i0 = f0.to(torch.int64).clamp(0, i - 1)
        i1 = (i0 + 1).clamp(max=i - 1)
        return i0, i1, src - f0

    y0, y1, dy = taps(ih, oh)
    x0, x1, dx = taps(iw, ow)
    dx = dx.view(1, 1, 1, ow)
    dy = dy.view(1, 1, oh, 1)
    r0, r1 = x.index_select(2, y0), x.index_select(2, y1)
    top = r0.index_select(3, x0) * (1.0 - dx) + r0.index_select(3, x1) * dx
    bottom = r1.index_select(3, x0) * (1.0 - dx) + r1.index_select(3, x1) * dx
    return top * (1.0 - dy) + bottom * dy


def resize_bilinear(x: Tensor, out_hw: Sequence[int], method: int = InterpolationMethod.CUSTOM) -> Tensor:
    """interpolate.rs:123-134."""
    if method == InterpolationMethod.CUSTOM:
        return resize_align_corners_false(x, out_hw)
    return resize_align_corners_true(x, out_hw)


def resize_bilinear_scale(x: Tensor, scale: Sequence[float], method: int = InterpolationMethod.CUSTOM) -> Tensor:
    """interpolate.rs:136-145."""
    th = compute_output_size(x.shape[2], scale[0])
    tw = compute_output_size(x.shape[3], scale[1])
    return resize_bilinear(x, (th, tw), method)


# ---------------------------------------------------------------------------------------------
# a3/a5/a6  split / reshape_feature / merge  (layers/encoder.rs:28-38,190-319)
# ---------------------------------------------------------------------------------------------
def split_geometry(image_size: int, patch_size: int, overlap_ratio: float) -> Tuple[int, int]:
    """encoder.rs:196-206 -> (stride, steps)."""
    stride = max(int(math.floor(float(np.float32(patch_size) * (np.float32(1.0) - np.float32(overlap_ratio))))), 1)
    stride = min(stride, patch_size)
    if patch_size >= image_size:
        steps = 1
    else:
        steps = 1 + -(-(image_size - patch_size) // stride)
    return stride, steps


def split(x: Tensor, patch_size: int, overlap_ratio: float) -> Tuple[Tensor, int, int]:
    """encoder.rs:190-232. Tile order: index (j*steps+i)*B + b (cat on dim 0 of [B,...] slices)."""
    image_size = x.shape[3]
    stride, steps = split_geometry(image_size, patch_size, overlap_ratio)
    tiles = []
    for j in range(steps):
        j0 = j * stride
        for i in range(steps):
            i0 = i * stride
            tiles.append(x[:, :, j0:j0 + patch_size, i0:i0 + patch_size])
    return torch.cat(tiles, 0), steps, stride


def feature_padding(patch_size: int, stride: int, feature_patch_size: int) -> int:
    """encoder.rs:28-38."""
    if feature_patch_size == 0 or patch_size == 0:
        return 0
    denom = max(patch_size, 1)
    feature_stride = (stride * feature_patch_size + denom // 2) // denom
    return max(feature_patch_size - feature_stride, 0) // 2


def merge(x: Tensor, batch_size: int, padding: int) -> Tensor:
    """encoder.rs:234-282."""
    n, c, h, w = x.shape
    steps = int(round(math.sqrt(n // batch_size)))
    if steps == 0:
        return x.new_zeros(batch_size, c, 0, 0)
    rows = []
    for j in range(steps):
        row = []
        for i in range(steps):
            idx = j * steps + i
            patch = x[batch_size * idx: batch_size * (idx + 1)]
            top = 0 if j == 0 else padding
            bottom = 0 if j == steps - 1 else padding
            left = 0 if i == 0 else padding
            right = 0 if i == steps - 1 else padding
            row.append(patch[:, :, top:h - bottom, left:w - right])
        rows.append(torch.cat(row, 3))
    return torch.cat(rows, 2)


def reshape_feature(emb: Tensor, width: int, height: int, cls_token_offset: int) -> Tensor:
    """encoder.rs:284-319."""
    b, tokens, dim = emb.shape
    spatial = width * height
    assert spatial <= tokens, f"cannot reshape {tokens} tokens into {width}x{height}"
    offset = cls_token_offset if tokens - cls_token_offset >= spatial else tokens - spatial
    emb = emb[:, offset:offset + spatial]
    return emb.reshape(b, height, width, dim).permute(0, 3, 1, 2)


# ---------------------------------------------------------------------------------------------
# a4  DINOv2 ViT (burn_dino 0.6.0 -- un-vendored; public DINOv2 definition; PARITY UNPINNED)
#     call sites: layers/vit.rs:45-68, encoder.rs:346-348,409, fov.rs:203
# ---------------------------------------------------------------------------------------------
FP8_ACT_SCALE = 8.0 / 448.0   # engine: Da3State::kActScale (LayerNorm / attention outputs)
FP8_HID_SCALE = 16.0 / 448.0  # engine: Da3State::kHidScale (GELU output)


def fp8_static(x: Tensor, scale: float) -> Tensor:
    """e4m3 MFMA-operand emulation on a static scale, with the engine's arithmetic: fp32 multiply by the fp32
    reciprocal, saturate at +-448, OCP e4m3fn cast (bit-identical to v_cvt_pk_fp8_f32, checked on the GPU)."""
    s32 = torch.tensor(scale, dtype=torch.float32)
    return (x.float() * (1.0 / s32)).clamp(-448, 448).to(torch.float8_e4m3fn).float() * s32


def fp8_rows(w: Tensor) -> Tensor:
    """per-output-channel e4m3 weights as pack_fp8_rows_kernel makes them: scale = amax * (1/448), inv = 1/scale."""
    amax = w.abs().amax(1, keepdim=True).float()
    sc = torch.where(amax > 0, amax * torch.tensor(1.0 / 448.0, dtype=torch.float32), torch.ones_like(amax))
    return (w.float() * (1.0 / sc)).clamp(-448, 448).to(torch.float8_e4m3fn).float() * sc


def linear_quantisers(q: "Quant", fp8: bool):
    """(LayerNorm-output, attention-output, GELU-output, weight) operand quantisers of the four ViT linear layers:
    `q` everywhere (bf16 emulation or identity), or the MD_PREC_FP8 scheme."""
    if not fp8:
        return q, q, q, q
    return (lambda t: fp8_static(t, FP8_ACT_SCALE), lambda t: fp8_static(t, FP8_ACT_SCALE),
            lambda t: fp8_static(t, FP8_HID_SCALE), fp8_rows)


def vit_forward(x: Tensor, W: Dict[str, Tensor], prefix: str, v: ViTConfig, hook_ids: Sequence[int],
                q: Quant = identity, chunk: int = 8, fp8: bool = False) -> Tuple[Tensor, List[Tensor]]:
    """Returns (x_norm_patchtokens [B, g*g, D], hooks: un-normalised tokens incl. cls after the
    0-based block indices ``hook_ids``; vit.rs:63 ``normalize_intermediate_tokens=false``).

    Switches set by the reference (vit.rs:60-63): plain softmax (not quiet), no register tokens,
    no mask token.  qkv bias, exact-erf GELU, LayerScale, LN eps = v.ln_eps."""
    outs, hooks_acc = [], None
    for s in range(0, x.shape[0], chunk):  # chunk only bounds peak memory
        o, h = _vit_forward_chunk(x[s:s + chunk], W, prefix, v, hook_ids, q, fp8)
        outs.append(o)
        hooks_acc = [[t] for t in h] if hooks_acc is None else [a + [t] for a, t in zip(hooks_acc, h)]
    return torch.cat(outs, 0), [torch.cat(a, 0) for a in (hooks_acc or [])]


def interpolate_pos_encoding(pos: Tensor, gh: int, gw: int, offset: float = 0.1) -> Tensor:
    """Public DINOv2 `interpolate_pos_encoding` (bicubic, no antialias, the upstream default `interpolate_offset` = 0.1:
    scale factors (g + 0.1) / M): used when the input grid differs from the pos_embed grid (Depth-Anything-v3 at sizes
    other than 518). burn_dino's own version is not visible -> parity unpinned. `offset` = 0 is the other published form
    (an output SIZE instead of scale factors: upstream with `interpolate_offset = 0`, Hugging Face's `Dinov2Embeddings`);
    tests/test_oracle_vs_hf.py holds that form to Hugging Face's and measures what the 0.1 moves."""
    n = pos.shape[1] - 1
    M = int(round(math.sqrt(n)))
    if gh == M and gw == M:
        return pos
    D = pos.shape[2]
    patch = pos[:, 1:].reshape(1, M, M, D).permute(0, 3, 1, 2)
    if offset:
        patch = F.interpolate(patch, scale_factor=((gh + offset) / M, (gw + offset) / M), mode="bicubic", align_corners=False)
    else:
        patch = F.interpolate(patch, size=(gh, gw), mode="bicubic", align_corners=False)
    assert patch.shape[-2:] == (gh, gw)
    return torch.cat([pos[:, :1], patch.permute(0, 2, 3, 1).reshape(1, gh * gw, D)], 1)


# Operand-rounding emulation only (q is not identity): restate the engine's LayerNorm fold (every norm2, norm1 of blocks 1..) so that a
# test can hold a 16-bit mode to an oracle that rounds WHERE the engine rounds. False = the reference's order of operations with the
# operand roundings of the unfolded schedule. The fp32 oracle (q = identity) never looks at it.
LN_FOLD_EMULATION = False

ATTN_QSCALE = float(np.float32(0.125) * np.float32(1.4426950408889634))  # head_dim^-0.5 * log2(e), head_dim = 64


def round_q_prescaled(qq: Tensor, q: Quant) -> Tensor:
    """Operand-rounding emulation of the engine's q: the softmax scale is folded into q BEFORE its one rounding to the
    operand type (csrc/kernels/ops.h kAttnQScale), so the rounded value is q * scale; returned un-scaled again so that the
    caller's softmax(q k^T * head_dim^-0.5) stays as the reference writes it. Identity for the fp32 oracle."""
    if q is identity:
        return qq
    return q(qq * ATTN_QSCALE) / ATTN_QSCALE


def _vit_forward_chunk(x, W, prefix, v, hook_ids, q, fp8=False):
    B = x.shape[0]
    qn, qo, qh, qw = linear_quantisers(q, fp8)
    D, Hn, hd = v.embed_dim, v.num_heads, v.head_dim
    p = lambda n: W[f"{prefix}.{n}"]
    tok = F.conv2d(q(x), q(p("patch_embed.proj.weight")), p("patch_embed.proj.bias"), stride=v.patch_size)
    tok = tok.flatten(2).transpose(1, 2)                      # [B, g*g, D], row-major (h, w)
    pos = interpolate_pos_encoding(p("pos_embed"), x.shape[2] // v.patch_size, x.shape[3] // v.patch_size)
    xs = torch.cat([p("cls_token").expand(B, 1, D), tok], 1) + pos
    N = xs.shape[1]
    scale = hd ** -0.5
    hooks: List[Tensor] = []

    def norm_linear(xs_, g_, b_, w_, bias_, folded):
        """LN(x) W^T + bias. `folded` (operand-rounding emulation only): the engine's LayerNorm fold -- the MFMA operand is
        round(gamma . x), not round(LN x), and the row statistics finish the product: rstd (acc - mu c) + d with c = Wq gamma,
        d = Wq beta + bias (csrc/kernels/gemm.h GemmParams::ln_*). Same value in exact arithmetic; the rounding falls elsewhere."""
        if not folded:
            return F.linear(qn(F.layer_norm(xs_, (D,), g_, b_, v.ln_eps)), qw(w_), bias_)
        wq_ = qw(w_)
        mu = xs_.mean(-1, keepdim=True)
        rstd = (xs_.var(-1, unbiased=False, keepdim=True) + v.ln_eps).rsqrt()
        return rstd * (F.linear(qn(xs_ * g_), wq_) - mu * (wq_ @ g_)) + (wq_ @ b_ + bias_)

    fold = LN_FOLD_EMULATION and q is not identity and not fp8
    for i in range(v.depth):
        b = f"blocks.{i}"
        qkv = norm_linear(xs, p(f"{b}.norm1.gamma"), p(f"{b}.norm1.beta"), p(f"{b}.attn.qkv.weight"), p(f"{b}.attn.qkv.bias"), fold and i > 0)
        qkv = qkv.reshape(B, N, 3, Hn, hd).permute(2, 0, 3, 1, 4)
        qq, kk, vv = round_q_prescaled(qkv[0], q), q(qkv[1]), q(qkv[2])
        s = (qq @ kk.transpose(-2, -1)) * scale
        pu = torch.exp(s - s.amax(-1, keepdim=True))
        o = (q(pu) @ vv) / pu.sum(-1, keepdim=True)
        o = qo(o.transpose(1, 2).reshape(B, N, D))
        xs = xs + p(f"{b}.ls1.gamma") * F.linear(o, qw(p(f"{b}.attn.proj.weight")), p(f"{b}.attn.proj.bias"))
        h = qh(F.gelu(norm_linear(xs, p(f"{b}.norm2.gamma"), p(f"{b}.norm2.beta"), p(f"{b}.mlp.fc1.weight"), p(f"{b}.mlp.fc1.bias"), fold)))
        xs = xs + p(f"{b}.ls2.gamma") * F.linear(h, qw(p(f"{b}.mlp.fc2.weight")), p(f"{b}.mlp.fc2.bias"))
        for hid in hook_ids:
            if hid == i:
                hooks.append(xs.clone())
    xn = F.layer_norm(xs, (D,), p("norm.gamma"), p("norm.beta"), v.ln_eps)
    return xn[:, 1:], hooks


# ---------------------------------------------------------------------------------------------
# a8  encoder  (layers/encoder.rs:41-84,321-459)
# ---------------------------------------------------------------------------------------------
def _project_upsample(x: Tensor, W, name: str, q: Quant) -> Tensor:
    """ProjectUpsampleBlock::forward (encoder.rs:77-83): 1x1 conv (no bias) then k2s2 deconvs."""
    x = q(F.conv2d(q(x), q(W[f"{name}.projection.weight"])))
    l = 0
    while f"{name}.upsample.{l}.weight" in W:
        x = q(F.conv_transpose2d(x, q(W[f"{name}.upsample.{l}.weight"]), stride=2))
        l += 1
    return x


def encoder_forward_debug(x: Tensor, W, cfg, q: Quant = identity) -> Dict[str, object]:
    """DepthProEncoder::forward_with_debug (encoder.rs:321-454). Keys follow ``EncoderDebug``."""
    pv, iv, _ = _cfg_vits(cfg)
    B = x.shape[0]
    win, out_size = pv.img_size, pv.grid_size()
    m = cfg.interpolation
    x0 = x
    x1 = resize_bilinear_scale(x, (0.5, 0.5), m)      # encoder.rs:326
    x2 = resize_bilinear_scale(x, (0.25, 0.25), m)    # encoder.rs:327
    s0, steps0, stride0 = split(x0, win, 0.25)
    s1, steps1, stride1 = split(x1, win, 0.5)
    s2 = x2
    pyramid = torch.cat([s0, s1, s2], 0)
    tokens, hooks = vit_forward(pyramid, W, "encoder.patch_encoder", pv, pv.encoder_feature_layer_ids, q)
    assert len(hooks) >= 2, "DepthPro encoder expects at least two hook tokens"
    enc = reshape_feature(q(tokens), out_size, out_size, 0)
    len0, len1, len2 = s0.shape[0], s1.shape[0], s2.shape[0]
    x0_enc, x1_enc, x2_enc = torch.split(enc, [len0, len1, len2], 0)
    high_count = B * steps0 * steps0
    lat0_in = reshape_feature(q(hooks[0]), out_size, out_size, 1)
    lat1_in = reshape_feature(q(hooks[1]), out_size, out_size, 1)
    lat0, lat1 = lat0_in[:high_count], lat1_in[:high_count]
    high_pad = feature_padding(win, stride0, out_size)
    mid_pad = feature_padding(win, stride1, out_size)
    merged_lat0 = merge(lat0, B, high_pad)
    merged_lat1 = merge(lat1, B, high_pad)
    merged_x0 = merge(x0_enc, B, high_pad)
    merged_x1 = merge(x1_enc, B, mid_pad)
    merged_x2 = x2_enc

    g_tokens, _ = vit_forward(s2, W, "encoder.image_encoder", iv, (), q)       # encoder.rs:409
    g = reshape_feature(q(g_tokens), out_size, out_size, 0)
    g = q(F.conv_transpose2d(g, q(W["encoder.upsample_lowres.weight"]), W["encoder.upsample_lowres.bias"], stride=2))
    up_x2 = _project_upsample(merged_x2, W, "encoder.upsample2", q)
    g = q(F.conv2d(torch.cat([up_x2, g], 1), q(W["encoder.fuse_lowres.weight"]), W["encoder.fuse_lowres.bias"]))
    feats = [
        _project_upsample(merged_lat0, W, "encoder.upsample_latent0", q),
        _project_upsample(merged_lat1, W, "encoder.upsample_latent1", q),
        _project_upsample(merged_x0, W, "encoder.upsample0", q),
        _project_upsample(merged_x1, W, "encoder.upsample1", q),
        g,
    ]
    return dict(features=feats, latent0=merged_lat0, latent1=merged_lat1, latent0_tokens=lat0,
                latent1_tokens=lat1, latent0_merge_input=lat0_in, latent1_merge_input=lat1_in,
                x0_tokens=x0_enc, x1_tokens=x1_enc, x2_tokens=x2_enc, split_x0=s0, split_x1=s1,
                split_x2=s2, merged_x0=merged_x0, merged_x1=merged_x1, merged_x2=merged_x2)


# ---------------------------------------------------------------------------------------------
# a9  decoder  (layers/decoder.rs:47-234)
# ---------------------------------------------------------------------------------------------
def _residual_block(x: Tensor, W, name: str, q: Quant, extra: Optional[Tensor] = None) -> Tensor:
    """ResidualBlock::forward (decoder.rs:74-87), batch_norm=false (decoder.rs:183).
    ``extra`` is the fusion skip added right after (decoder.rs:122-125); it is folded in before the
    single storage rounding because the engine adds it in the same epilogue."""
    out = F.relu(x)
    out = q(F.relu(F.conv2d(q(out), q(W[f"{name}.conv1.weight"]), W[f"{name}.conv1.bias"], padding=1)))
    out = F.conv2d(out, q(W[f"{name}.conv2.weight"]), W[f"{name}.conv2.bias"], padding=1)
    out = out + x
    if extra is not None:
        out = extra + out
    return q(out)


def _fusion(x0: Tensor, x1: Optional[Tensor], W, name: str, q: Quant) -> Tensor:
    """FeatureFusionBlock2d::forward (decoder.rs:119-134)."""
    x = x0
    if x1 is not None:
        x = _residual_block(x1, W, f"{name}.resnet1", q, extra=x0)
    x = _residual_block(x, W, f"{name}.resnet2", q)
    if f"{name}.deconv.weight" in W:
        x = q(F.conv_transpose2d(x, q(W[f"{name}.deconv.weight"]), stride=2))
    return q(F.conv2d(x, q(W[f"{name}.out_conv.weight"]), W[f"{name}.out_conv.bias"]))


def decoder_forward_debug(enc: List[Tensor], W, q: Quant = identity):
    """MultiresConvDecoder::forward_with_debug (decoder.rs:195-222) -> (features, lowres, fusions)."""
    n = len(enc)

    def proj(l, t):
        key = f"decoder.convs.{l}.conv.weight"
        if key not in W:
            return t  # identity (decoder.rs:163-165)
        k = W[key].shape[-1]
        return q(F.conv2d(q(t), q(W[key]), padding=k // 2))

    feats = proj(n - 1, enc[n - 1])
    lowres = feats
    fusions = []
    feats = _fusion(feats, None, W, f"decoder.fusions.{n - 1}", q)
    fusions.append(feats)
    for l in range(n - 2, -1, -1):
        feats = _fusion(feats, proj(l, enc[l]), W, f"decoder.fusions.{l}", q)
        fusions.append(feats)
    fusions.reverse()
    return feats, lowres, fusions


# ---------------------------------------------------------------------------------------------
# a10  depth head  (depth_pro/mod.rs:68-117)
# ---------------------------------------------------------------------------------------------
def head_debug(feat: Tensor, W, q: Quant = identity) -> Dict[str, Tensor]:
    """DepthPro::head_debug (mod.rs:262-278). In bf16 mode the engine fuses conv1+relu+conv_out+relu
    in one epilogue and keeps that chain in fp32, hence no q() after conv1."""
    conv0 = q(F.conv2d(q(feat), q(W["head.conv0.weight"]), W["head.conv0.bias"], padding=1))
    deconv = q(F.conv_transpose2d(conv0, q(W["head.deconv.weight"]), W["head.deconv.bias"], stride=2))
    conv1 = F.conv2d(deconv, q(W["head.conv1.weight"]), W["head.conv1.bias"], padding=1)
    relu = F.relu(conv1)
    pre_out = F.conv2d(relu, W["head.conv_out.weight"], W["head.conv_out.bias"])
    return dict(conv0=conv0, deconv=deconv, conv1=conv1, relu=relu, pre_out=pre_out, canonical=F.relu(pre_out))


# ---------------------------------------------------------------------------------------------
# a11  FOV network  (layers/fov.rs:16-247)
# ---------------------------------------------------------------------------------------------
def _conv_act(x: Tensor, W, name: str, stride: int, padding: int, relu: bool, method: int) -> Tensor:
    """ConvActivation + ensure_min_spatial (fov.rs:40-43,229-246). The FOV head is computed in
    fp32 by the engine in both precision modes (0.5 GFLOP), so no operand rounding here."""
    w = W[f"{name}.weight"]
    kh, kw = w.shape[2], w.shape[3]
    if x.shape[2] < kh or x.shape[3] < kw:
        x = resize_bilinear(x, (max(x.shape[2], kh), max(x.shape[3], kw)), InterpolationMethod.CUSTOM
                            if method == InterpolationMethod.CUSTOM else method)
    out = F.conv2d(x, w, W[f"{name}.bias"], stride=stride, padding=padding)
    return F.relu(out) if relu else out


def fov_forward(x: Tensor, lowres: Tensor, W, cfg, q: Quant = identity) -> Tensor:
    """FOVNetwork::forward (fov.rs:168-227) -> [B,1,h,w] (h=w=1 for the shipped presets)."""
    m = cfg.interpolation
    fv = _cfg_vits(cfg)[2]
    if fv is None:  # fov.rs:119-154 branch
        t = _conv_act(lowres, W, "fov.head_blocks.0.conv", 2, 1, True, m)
        t = _conv_act(t, W, "fov.head_blocks.1.conv", 2, 1, True, m)
        t = _conv_act(t, W, "fov.head_blocks.2.conv", 2, 1, True, m)
        return _conv_act(t, W, "fov.head_blocks.3.conv", 1, 0, False, m)
    feats = _conv_act(lowres, W, "fov.downsample_blocks.0.conv", 2, 1, True, m)
    xs = resize_bilinear_scale(x, (0.25, 0.25), m)                       # fov.rs:202
    tokens, _ = vit_forward(xs, W, "fov.encoder", fv, (), q)             # fov.rs:203
    proj = F.linear(q(tokens), q(W["fov.encoder_proj.weight"]), W["fov.encoder_proj.bias"])
    b, n, c = proj.shape
    enc = proj.permute(0, 2, 1).reshape(feats.shape)                     # fov.rs:219-226
    t = feats + enc
    t = _conv_act(t, W, "fov.head_blocks.0.conv", 2, 1, True, m)
    t = _conv_act(t, W, "fov.head_blocks.1.conv", 2, 1, True, m)
    return _conv_act(t, W, "fov.head_blocks.2.conv", 1, 0, False, m)


# ---------------------------------------------------------------------------------------------
# a12/a13  DepthPro::forward / infer  (depth_pro/mod.rs:210-252,312-414)
# ---------------------------------------------------------------------------------------------
def fovy_from_fovx_rad(fovx_rad: Tensor, h: int, w: int) -> Tensor:
    """mod.rs:370-414: 2*atan((H/W)*tan(fovx/2)) with the rational atan approximation; the
    approximation error is part of the reference result. Scalar constants are f64 in the source and
    are converted to the f32 element type by Burn's scalar ops."""
    k = torch.tensor(0.273, dtype=torch.float32)
    pi4 = torch.tensor(math.pi / 4, dtype=torch.float32)
    pi2 = torch.tensor(math.pi / 2, dtype=torch.float32)
    aspect = torch.tensor(float(h) / float(w), dtype=torch.float32)
    t = torch.tan(fovx_rad * 0.5) * aspect
    s = torch.sign(t)
    ax = torch.abs(t)
    use_inv = (ax > 1.0).to(torch.float32)
    inv = 1.0 / ax
    xr = ax * (1.0 - use_inv) + inv * use_inv
    inner = (1.0 - xr) * k + pi4
    atan_reduced = xr * inner
    delta = pi2 - atan_reduced * 2.0
    atan_ax = atan_reduced + delta * use_inv
    return atan_ax * s * 2.0


def forward_debug(x: Tensor, W, cfg, q: Quant = identity) -> Dict[str, object]:
    """DepthPro::forward_internal (mod.rs:210-252) with every debug tap."""
    enc = encoder_forward_debug(x, W, cfg, q)
    feats, lowres, fusions = decoder_forward_debug(enc["features"], W, q)
    hd = head_debug(feats, W, q)
    fov = None
    if cfg.use_fov_head:
        fov = fov_forward(x, lowres, W, cfg, q).reshape(x.shape[0])      # mod.rs:238-243
    return dict(encoder=enc, decoder_features=feats, decoder_lowres=lowres, fusions=fusions,
                head=hd, canonical=hd["canonical"], fov_deg=fov)


def infer(x: Tensor, W, cfg, q: Quant = identity, debug: bool = False):
    """DepthPro::infer (mod.rs:312-364) -> dict(depth[B,H,W], focallength_px[B], fovx_deg[B], fovy_rad[B])."""
    B, _, H, Wd = x.shape
    S = ref_config.img_size_for(cfg.patch_encoder_preset)
    resize_needed = (S != H) or (S != Wd)
    xin = resize_bilinear(x, (S, S), cfg.interpolation) if resize_needed else x
    dbg = forward_debug(xin, W, cfg, q)
    if dbg["fov_deg"] is None:
        raise RuntimeError("FOV head required for focal length")        # mod.rs:329
    fovx_deg = dbg["fov_deg"]
    fovx_rad = fovx_deg * torch.tensor(math.pi / 180.0, dtype=torch.float32)
    focal_px = (torch.ones_like(fovx_deg) * (float(Wd) * 0.5)) / torch.tan(fovx_rad * 0.5)
    ratio = (torch.ones_like(focal_px) * float(Wd)) / focal_px
    inv = dbg["canonical"] * ratio.reshape(B, 1, 1, 1)
    if resize_needed:
        inv = resize_bilinear(inv, (H, Wd), cfg.interpolation)
    depth = (1.0 / inv.clamp(1e-4, 1e4)).squeeze(1)
    out = dict(depth=depth, focallength_px=focal_px.reshape(B), fovx_deg=fovx_deg,
               fovy_rad=fovy_from_fovx_rad(fovx_rad, H, Wd), inverse_depth=inv)
    if debug:
        out["debug"] = dbg
    return out


def weights_to_torch(weights_np: Dict[str, np.ndarray]) -> Dict[str, Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights_np.items()}
