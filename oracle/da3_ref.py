"""CPU oracle for Depth-Anything-v3 (`metric_large` mono head; `small` dual head + camera decoder) -- TEST INFRASTRUCTURE ONLY.

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

from oracle import ref_config

DepthAnything3Config = object  # annotation only: the oracle checks a product config against its own table (ref_config.check_da3)
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


def backbone_hooks(x: Tensor, W, cfg: DepthAnything3Config, q=identity, fp8: bool = False) -> List[Tensor]:
    """Backbone::forward_with_hooks (mod.rs:202-215): per hook block the final-norm'ed patch tokens."""
    v = cfg.vit()
    _, raw = vit_forward(x, W, "backbone.pretrained", v, cfg.hook_block_ids, q, fp8=fp8)
    g, b = W["backbone.pretrained.norm.gamma"], W["backbone.pretrained.norm.beta"]
    return [F.layer_norm(h, (v.embed_dim,), g, b, v.ln_eps)[:, 1:] for h in raw]


# ---------------------------------------------------------------------------------------------
# `small` variant: burn_dino backbone extras + dual head + camera decoder
#
# PARITY STATUS of the backbone extras: **unpinned and under-specified** in the reference tree -- they live in
# burn_dino 0.6.0 and the reference only sets the switches (mod.rs:190-196: alt_block_start = qk_norm_block_start =
# rope_block_start = 4, cat_token, use_camera_tokens). Restated here from the public Depth-Anything-3 model
# definition those switches are named after (single view, S = 1):
#   * from block `ext_block_start` on, q and k get a per-head affine LayerNorm (head_dim) and then a 2-D rotary
#     embedding: the first half of head_dim rotates with the token's row, the second half with its column;
#     within a half, pairs (j, j + n/2) rotate by pos * base^(-2j/n), base = 100; patch positions are 1-based,
#     the cls slot sits at (0, 0);
#   * blocks alternate local / global attention (global = odd block index >= start). With one view the two
#     attend over the same tokens; global blocks use "no-diff" positions (every patch at (1, 1));
#   * entering block `ext_block_start` the cls slot is overwritten with the learned reference camera token;
#   * a hook is cat(x after the last LOCAL block, x after the hook block) (dim 2D) and only its second half
#     goes through the final LayerNorm; the camera feature is token 0 of the raw (un-normalised) concat.
# ---------------------------------------------------------------------------------------------
def _rope_half(t: Tensor, pos: Tensor, base: float) -> Tensor:
    """t [B, heads, N, n], pos [N] -> rotated t (pairs (j, j + n/2))."""
    n = t.shape[-1]
    inv = 1.0 / (base ** (torch.arange(0, n, 2, dtype=torch.float32) / n))   # [n/2]
    ang = pos.to(torch.float32)[:, None] * inv[None, :]                         # [N, n/2]
    ang = torch.cat([ang, ang], -1)                                             # [N, n]
    rot = torch.cat([-t[..., n // 2:], t[..., :n // 2]], -1)
    return t * ang.cos() + rot * ang.sin()


def rope2d(t: Tensor, pos: Tensor, base: float) -> Tensor:
    """t [B, heads, N, head_dim], pos [N, 2] = (row, col)."""
    n = t.shape[-1] // 2
    return torch.cat([_rope_half(t[..., :n], pos[:, 0], base), _rope_half(t[..., n:], pos[:, 1], base)], -1)


def backbone_hooks_ext(x: Tensor, W, cfg: DepthAnything3Config, q=identity, fp8: bool = False, camera_token=None):
    """Returns (hooks: 4 x [B, P, 2D] with the second half final-norm'ed, camera feature [B, 2D] of the last hook).
    `camera_token` [B, D]: the camera encoder's output (mod.rs:522-531); it takes the place of the learned reference token."""
    from oracle.depth_pro_ref import interpolate_pos_encoding, linear_quantisers, round_q_prescaled
    qn, qo, qh, qw = linear_quantisers(q, fp8)
    v = cfg.vit()
    bp = "backbone.pretrained"
    p = lambda n: W[f"{bp}.{n}"]
    B = x.shape[0]
    D, Hn, hd = v.embed_dim, v.num_heads, v.head_dim
    gh, gw = x.shape[2] // v.patch_size, x.shape[3] // v.patch_size
    tok = F.conv2d(q(x), q(p("patch_embed.proj.weight")), p("patch_embed.proj.bias"), stride=v.patch_size).flatten(2).transpose(1, 2)
    xs = torch.cat([p("cls_token").expand(B, 1, D), tok], 1) + interpolate_pos_encoding(p("pos_embed"), gh, gw)
    N = xs.shape[1]
    yy, xx = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    pos_l = torch.cat([torch.zeros(1, 2, dtype=torch.long), torch.stack([yy.reshape(-1), xx.reshape(-1)], 1) + 1], 0)
    pos_g = torch.cat([torch.zeros(1, 2, dtype=torch.long), torch.ones(gh * gw, 2, dtype=torch.long)], 0)
    start = cfg.ext_block_start
    local_x = xs
    raw = {}
    for i in range(v.depth):
        b = f"blocks.{i}"
        ext = start >= 0 and i >= start
        if ext and i == start:
            cam_tok = p("camera_token")[:, :1].expand(B, 1, D) if camera_token is None else camera_token.reshape(B, 1, D)
            xs = torch.cat([cam_tok, xs[:, 1:]], 1)
        is_global = ext and i % 2 == 1
        xn = qn(F.layer_norm(xs, (D,), p(f"{b}.norm1.gamma"), p(f"{b}.norm1.beta"), v.ln_eps))
        qkv = F.linear(xn, qw(p(f"{b}.attn.qkv.weight")), p(f"{b}.attn.qkv.bias"))
        qkv = qkv.reshape(B, N, 3, Hn, hd).permute(2, 0, 3, 1, 4)
        qq, kk, vv = round_q_prescaled(qkv[0], q), q(qkv[1]), q(qkv[2])
        if ext:
            pos = pos_g if is_global else pos_l
            qq = F.layer_norm(qq, (hd,), p(f"{b}.attn.q_norm.gamma"), p(f"{b}.attn.q_norm.beta"), cfg.qk_norm_eps)
            kk = F.layer_norm(kk, (hd,), p(f"{b}.attn.k_norm.gamma"), p(f"{b}.attn.k_norm.beta"), cfg.qk_norm_eps)
            qq, kk = round_q_prescaled(rope2d(qq, pos, cfg.rope_frequency), q), q(rope2d(kk, pos, cfg.rope_frequency))
        sc = (qq @ kk.transpose(-2, -1)) * hd ** -0.5
        pu = torch.exp(sc - sc.amax(-1, keepdim=True))
        o = (q(pu) @ vv) / pu.sum(-1, keepdim=True)
        o = qo(o.transpose(1, 2).reshape(B, N, D))
        xs = xs + p(f"{b}.ls1.gamma") * F.linear(o, qw(p(f"{b}.attn.proj.weight")), p(f"{b}.attn.proj.bias"))
        xn = qn(F.layer_norm(xs, (D,), p(f"{b}.norm2.gamma"), p(f"{b}.norm2.beta"), v.ln_eps))
        h = qh(F.gelu(F.linear(xn, qw(p(f"{b}.mlp.fc1.weight")), p(f"{b}.mlp.fc1.bias"))))
        xs = xs + p(f"{b}.ls2.gamma") * F.linear(h, qw(p(f"{b}.mlp.fc2.weight")), p(f"{b}.mlp.fc2.bias"))
        if not is_global:
            local_x = xs
        if i in cfg.hook_block_ids:
            raw[i] = torch.cat([local_x, xs], -1)
    hooks, cam = [], None
    for i in cfg.hook_block_ids:
        r = raw[i]
        hooks.append(torch.cat([r[..., :D], F.layer_norm(r[..., D:], (D,), p("norm.gamma"), p("norm.beta"), v.ln_eps)], -1)[:, 1:])
        cam = r[:, 0]
    return hooks, cam


def dual_head_forward(hooks: List[Tensor], height: int, width: int, W, cfg: DepthAnything3Config, q=identity, debug=None):
    """DualDepthAnything3Head::forward_dual (dpt.rs:227-280): returns depth, depth_confidence, aux, aux_confidence."""
    hp, sc = "head_dual", "head_dual.scratch"
    ps = cfg.patch_size
    ph, pw = height // ps, width // ps
    feats = []
    for s in range(4):  # prepare_stage (dpt.rs:282-317): affine LayerNorm (Burn default eps 1e-5), 1x1, + UV, resize layer
        x = q(F.layer_norm(hooks[s][:, :ph * pw], (cfg.dim_in,), W[f"{hp}.norm.gamma"], W[f"{hp}.norm.beta"], 1e-5))
        x = x.permute(0, 2, 1).reshape(x.shape[0], -1, ph, pw)
        x = F.conv2d(x, q(W[f"{hp}.projects.{s}.weight"]), W[f"{hp}.projects.{s}.bias"])
        if cfg.pos_embed:
            x = pos_embed_add(x, width, height)
        x = q(x)
        if s == 0:
            x = q(F.conv_transpose2d(x, q(W[f"{hp}.resize_layers.0.conv_t.weight"]), W[f"{hp}.resize_layers.0.conv_t.bias"], stride=4))
        elif s == 1:
            x = q(F.conv_transpose2d(x, q(W[f"{hp}.resize_layers.1.conv_t.weight"]), W[f"{hp}.resize_layers.1.conv_t.bias"], stride=2))
        elif s == 3:
            x = q(F.conv2d(x, q(W[f"{hp}.resize_layers.3.conv.weight"]), W[f"{hp}.resize_layers.3.conv.bias"], stride=2, padding=1))
        feats.append(x)
    rn = [q(F.conv2d(feats[i], q(W[f"{sc}.layer{i + 1}_rn.weight"]), padding=1)) for i in range(4)]

    def pyramid(suffix):
        out = _ffb(rn[3], None, rn[2].shape[2:], W, f"{sc}.refinenet4{suffix}", q)
        out = _ffb(out, rn[2], rn[1].shape[2:], W, f"{sc}.refinenet3{suffix}", q)
        out = _ffb(out, rn[1], rn[0].shape[2:], W, f"{sc}.refinenet2{suffix}", q)
        return _ffb(out, rn[0], None, W, f"{sc}.refinenet1{suffix}", q)

    # main branch (fuse_main + build_main_logits, dpt.rs:319-353)
    fused = q(F.conv2d(pyramid(""), q(W[f"{sc}.output_conv1.weight"]), W[f"{sc}.output_conv1.bias"], padding=1))
    fused = resize_bilinear(fused, (height, width))
    if cfg.pos_embed:
        fused = pos_embed_add(fused, width, height)
    fused = q(fused)
    t = F.relu(F.conv2d(fused, q(W[f"{sc}.output_conv2.conv1.weight"]), W[f"{sc}.output_conv2.conv1.bias"], padding=1))
    main = F.conv2d(t, W[f"{sc}.output_conv2.conv2.weight"], W[f"{sc}.output_conv2.conv2.bias"])
    # aux branch (build_aux_logits, dpt.rs:356-441): only the last level's neck and output head reach the result
    lvl = cfg.aux_levels - 1
    y = pyramid("_aux")
    for j in range(cfg.aux_out1_conv_num):
        n = f"{sc}.output_conv1_aux.{lvl}.layers.{j}"
        y = q(F.conv2d(y, q(W[f"{n}.weight"]), W[f"{n}.bias"], padding=1))
    neck = y
    if cfg.pos_embed:  # added twice (dpt.rs:428-435)
        y = pos_embed_add(pos_embed_add(y, width, height), width, height)
    y = q(y)
    o = f"{sc}.output_conv2_aux.{lvl}"
    t = F.conv2d(y, q(W[f"{o}.reduce.weight"]), W[f"{o}.reduce.bias"], padding=1)
    if f"{o}.norm.layer_norm.gamma" in W:
        t = F.layer_norm(t.permute(0, 2, 3, 1), (t.shape[1],), W[f"{o}.norm.layer_norm.gamma"], W[f"{o}.norm.layer_norm.beta"], 1e-5).permute(0, 3, 1, 2)
    aux_logits = F.conv2d(F.relu(t), W[f"{o}.project.weight"], W[f"{o}.project.bias"])
    if debug is not None:
        debug.update(stage_feats=feats, rn=rn, fused=fused, main_logits=main, aux_neck=neck, aux_head_input=y, aux_logits=aux_logits)
    k = cfg.aux_output_dim
    return dict(depth=torch.exp(main[:, 0]), depth_confidence=torch.exp(main[:, -1]) + 1.0,
                aux=aux_logits[:, :k - 1], aux_confidence=torch.exp(aux_logits[:, k - 1]) + 1.0)


def camera_decode(cam: Tensor, W, height: int, width: int):
    """CameraDecoder::forward + pose_encoding_to_extri_intri (camera.rs:143-199, 281-416), one view."""
    lin = lambda n, t: F.linear(t, W[f"camera_decoder.{n}.weight"], W[f"camera_decoder.{n}.bias"])
    h = F.relu(lin("backbone_2", F.relu(lin("backbone_1", cam))))
    pose = torch.cat([lin("fc_t", h), lin("fc_qvec", h), F.relu(lin("fc_fov", h))], 1)  # [B, 9]
    t, (qx, qy, qz, qw), fov = pose[:, :3], pose[:, 3:7].unbind(1), pose[:, 7:9]
    R = torch.stack([torch.stack([1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)], 1),
                     torch.stack([2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)], 1),
                     torch.stack([2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)], 1)], 1)
    Rt = R.transpose(1, 2)
    extr = torch.cat([Rt, -(Rt @ t[:, :, None])], 2)                                       # [B, 3, 4] world-to-camera
    tan_h = torch.sin(fov[:, 0] * 0.5) / torch.cos(fov[:, 0] * 0.5)
    tan_w = torch.sin(fov[:, 1] * 0.5) / torch.cos(fov[:, 1] * 0.5)
    intr = torch.zeros(cam.shape[0], 3, 3)
    intr[:, 0, 0], intr[:, 0, 2] = (width / 2.0) / tan_w, width / 2.0
    intr[:, 1, 1], intr[:, 1, 2] = (height / 2.0) / tan_h, height / 2.0
    intr[:, 2, 2] = 1.0
    return dict(pose_encoding=pose[:, None], extrinsics=extr[:, None], intrinsics=intr[:, None])


def matrix_to_quaternion(R: Tensor) -> Tensor:
    """camera.rs:418-514: four candidate quaternions (x, y, z, w), one picked per matrix by the branch masks
    (trace > 0; else m00 largest; else m11 > m22; else z). Clamps and the 1e-6 added to the divisors as in the reference."""
    m = lambda i, j: R[:, i, j]
    one, eps = torch.ones(R.shape[0]), 1e-6
    trace = m(0, 0) + m(1, 1) + m(2, 2)
    s = torch.sqrt(torch.clamp_min(trace + one, 1e-6)) * 2.0
    q_t = torch.stack([(m(2, 1) - m(1, 2)) / s, (m(0, 2) - m(2, 0)) / s, (m(1, 0) - m(0, 1)) / s, 0.25 * s], 1)
    s = torch.sqrt(torch.clamp_min(one + m(0, 0) - m(1, 1) - m(2, 2), 1e-6)) * 2.0
    q_x = torch.stack([0.25 * s, (m(0, 1) + m(1, 0)) / (s + eps), (m(0, 2) + m(2, 0)) / (s + eps), (m(2, 1) - m(1, 2)) / (s + eps)], 1)
    s = torch.sqrt(torch.clamp_min(one + m(1, 1) - m(0, 0) - m(2, 2), 1e-6)) * 2.0
    q_y = torch.stack([(m(0, 1) + m(1, 0)) / (s + eps), 0.25 * s, (m(1, 2) + m(2, 1)) / (s + eps), (m(0, 2) - m(2, 0)) / (s + eps)], 1)
    s = torch.sqrt(torch.clamp_min(one + m(2, 2) - m(0, 0) - m(1, 1), 1e-6)) * 2.0
    q_z = torch.stack([(m(0, 2) + m(2, 0)) / (s + eps), (m(1, 2) + m(2, 1)) / (s + eps), 0.25 * s, (m(1, 0) - m(0, 1)) / (s + eps)], 1)
    mk_t = (trace > 0).float()
    mk_x = (one - mk_t) * (m(0, 0) > m(1, 1)).float() * (m(0, 0) > m(2, 2)).float()
    mk_y = (one - mk_t - mk_x) * (m(1, 1) > m(2, 2)).float()
    mk_z = one - mk_t - mk_x - mk_y
    return q_t * mk_t[:, None] + q_x * mk_x[:, None] + q_y * mk_y[:, None] + q_z * mk_z[:, None]


def approx_atan_positive(x: Tensor) -> Tensor:
    """camera.rs:516-536: pi/4 v - v (v - 1)(0.2447 + 0.0663 v) for v <= 1, pi/2 - f(1/v) above."""
    f = lambda v: np.float32(math.pi / 4) * v - v * (v - 1.0) * (0.2447 + 0.0663 * v)
    small, large = f(x), np.float32(math.pi / 2) - f(1.0 / torch.clamp_min(x, 1e-6))
    mk = (x <= 1.0).float()
    return small * mk + large * (1.0 - mk)


def pose_encoding(extrinsics: Tensor, intrinsics: Tensor, height: int, width: int) -> Tensor:
    """extri_intri_to_pose_encoding (camera.rs:236-279): [B, V, 3, 4] world-to-camera + [B, V, 3, 3] ->
    [B, V, 9] = (camera-to-world translation, quaternion xyzw of the camera-to-world rotation, fov_h, fov_w)."""
    B, V = extrinsics.shape[:2]
    w2c = extrinsics.reshape(B * V, 3, 4).float()
    Rc2w = w2c[:, :, :3].transpose(1, 2)
    t = -(Rc2w @ w2c[:, :, 3:4])[:, :, 0]
    intr = intrinsics.reshape(B * V, 3, 3).float()
    fov_w = approx_atan_positive(np.float32(width / 2.0) / intr[:, 0, 0]) * 2.0
    fov_h = approx_atan_positive(np.float32(height / 2.0) / intr[:, 1, 1]) * 2.0
    return torch.cat([t, matrix_to_quaternion(Rc2w), fov_h[:, None], fov_w[:, None]], 1).reshape(B, V, 9)


def camera_encode(extrinsics: Tensor, intrinsics: Tensor, W, cfg: DepthAnything3Config, height: int, width: int, debug=None) -> Tensor:
    """CameraEncoder::forward (camera.rs:89-110): pose encoding -> PoseBranch (fc1, erf-GELU, fc2; :206-234) -> token_norm ->
    `trunk_depth` burn_dino Blocks over the V view tokens (qkv bias, LayerScale, plain softmax, no RoPE / qk-norm: :63-79) ->
    trunk_norm -> mean over views. Returns [B, D]. fp32 throughout. The Block itself is burn_dino's (un-vendored): restated as the
    standard pre-norm block the backbone uses, with the backbone's LayerNorm eps."""
    v = cfg.vit()
    D, Hn = v.embed_dim, cfg.cam_heads
    hd = D // Hn
    p = lambda n: W[f"camera_encoder.{n}"]
    lin = lambda n, t: F.linear(t, p(f"{n}.weight"), p(f"{n}.bias"))
    pe = pose_encoding(extrinsics, intrinsics, height, width)
    B, V = pe.shape[:2]
    x = lin("pose_branch.fc2", F.gelu(lin("pose_branch.fc1", pe)))
    x = F.layer_norm(x, (D,), p("token_norm.gamma"), p("token_norm.beta"), cfg.cam_ln_eps)
    if debug is not None:
        debug["pose_encoding_in"], debug["cam_tokens0"] = pe, x
    for i in range(cfg.cam_trunk_depth):
        b = f"trunk.{i}"
        qkv = lin(f"{b}.attn.qkv", F.layer_norm(x, (D,), p(f"{b}.norm1.gamma"), p(f"{b}.norm1.beta"), v.ln_eps))
        qkv = qkv.reshape(B, V, 3, Hn, hd).permute(2, 0, 3, 1, 4)
        a = torch.softmax((qkv[0] * hd ** -0.5) @ qkv[1].transpose(-1, -2), -1) @ qkv[2]
        x = x + p(f"{b}.ls1.gamma") * lin(f"{b}.attn.proj", a.transpose(1, 2).reshape(B, V, D))
        h = lin(f"{b}.mlp.fc2", F.gelu(lin(f"{b}.mlp.fc1", F.layer_norm(x, (D,), p(f"{b}.norm2.gamma"), p(f"{b}.norm2.beta"), v.ln_eps))))
        x = x + p(f"{b}.ls2.gamma") * h
    x = F.layer_norm(x, (D,), p("trunk_norm.gamma"), p("trunk_norm.beta"), cfg.cam_ln_eps)
    return x.mean(1)


def infer_raw(x: Tensor, W, cfg: DepthAnything3Config, q=identity) -> Tensor:
    """DepthAnything3::infer_raw (mod.rs:364-380): the dual head's `depth_logits` [B, 2, H, W] (output_conv2's result, dpt.rs:271,
    337-354), the mono head's `forward_raw` result [B, 1, H, W] (activation applied, dpt.rs:700)."""
    ref_config.check_da3(cfg)
    H, Wd = x.shape[2:]
    if cfg.dual_head:
        hooks, _ = backbone_hooks_ext(x, W, cfg, q)
        dbg = {}
        dual_head_forward(hooks, H, Wd, W, cfg, q, dbg)
        return dbg["main_logits"]
    return head_forward_raw(backbone_hooks(x, W, cfg, q), H, Wd, W, cfg, q)


def infer_from_tokens(patches: List[Tensor], W, cfg: DepthAnything3Config, height: int, width: int, q=identity, debug: bool = False):
    """DepthAnything3::infer_from_tokens (mod.rs:389-469): the head on caller-supplied hook tokens [B, T, din]; rows before
    `patch_start` are dropped (0 when T is the patch count, else `patch_token_start` = 1, mod.rs:419-424); no camera prediction."""
    ref_config.check_da3(cfg)
    ps = cfg.patch_size
    expected = max(height // ps, 1) * max(width // ps, 1)
    start = 0 if patches[0].shape[1] == expected else 1
    hooks = [t[:, start:] for t in patches]
    if any(h.shape[1] != expected for h in hooks):
        raise ValueError(f"{patches[0].shape[1]} tokens per image for a {height}x{width} input")
    dbg = {} if debug else None
    if cfg.dual_head:
        out = dual_head_forward(hooks, height, width, W, cfg, q, dbg)
    else:
        out = dict(depth=head_forward_raw(hooks, height, width, W, cfg, q, dbg)[:, 0])
    if debug:
        out["debug"] = dbg
    return out


def infer(x: Tensor, W, cfg: DepthAnything3Config, q=identity, debug: bool = False, fp8: bool = False, extrinsics=None, intrinsics=None):
    """DepthAnything3::infer (mod.rs:288-291 -> 495-564 -> 587-624): depth [B,H,W] (+ confidence, aux rays,
    aux confidence, pose encoding, extrinsics, intrinsics for the dual-head variant). With `extrinsics` [B, V, 3, 4] and
    `intrinsics` [B, V, 3, 3] it is `infer_with_camera` (mod.rs:301-309): the camera encoder's token conditions the backbone."""
    ref_config.check_da3(cfg)  # the product's variant tables against the oracle's own restatement of the reference's
    B, _, H, Wd = x.shape
    ps = cfg.patch_size
    if H % ps or Wd % ps:  # mod.rs:509-520 (panic)
        raise ValueError(f"Input {H}x{Wd} must be divisible by patch size {ps}")
    dbg = {} if debug else None
    if cfg.dual_head:
        cam_tok = None
        if extrinsics is not None and intrinsics is not None and cfg.camera_encoder:  # mod.rs:522-527
            cam_tok = camera_encode(extrinsics, intrinsics, W, cfg, H, Wd, dbg)
            if debug:
                dbg["camera_token"] = cam_tok
        hooks, cam = backbone_hooks_ext(x, W, cfg, q, fp8, cam_tok)
        out = dual_head_forward(hooks, H, Wd, W, cfg, q, dbg)
        out.update(camera_decode(cam, W, H, Wd))
        if debug:
            dbg["camera_feature"] = cam
    else:
        hooks = backbone_hooks(x, W, cfg, q, fp8)
        act = head_forward_raw(hooks, H, Wd, W, cfg, q, dbg)
        out = dict(depth=act[:, 0])  # select_depth_channel (dpt.rs:633-647)
    if debug:
        dbg["hooks"] = hooks
        out["debug"] = dbg
    return out
