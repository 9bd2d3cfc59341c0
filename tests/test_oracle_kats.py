"""Pins the CPU oracle against every known-answer test the reference holds for the Depth Pro
path. The expected numbers are the reference tests' own data."""
import math

import pytest
import torch

from burn_depth_amd.config import DepthProConfig, InterpolationMethod, vit_config_from_preset
from oracle import depth_pro_ref as R


def t4(vals, shape):
    return torch.tensor(vals, dtype=torch.float32).reshape(shape)


def test_align_corners_false_outputs_match_expected():
    # reference: depth_pro/interpolate.rs:166-219
    x = t4([1.0, 2.0, 3.0, 4.0], (1, 1, 2, 2))
    custom = R.resize_bilinear(x, (4, 4), InterpolationMethod.CUSTOM)
    burn = R.resize_bilinear(x, (4, 4), InterpolationMethod.BURN)
    exp_custom = t4([1.0, 1.25, 1.75, 2.0, 1.5, 1.75, 2.25, 2.5,
                     2.5, 2.75, 3.25, 3.5, 3.0, 3.25, 3.75, 4.0], (1, 1, 4, 4))
    exp_burn = t4([1.0, 1.3333334, 1.6666666, 2.0, 1.6666666, 2.0, 2.3333333, 2.6666667,
                   2.3333333, 2.6666667, 3.0, 3.3333333, 3.0, 3.3333333, 3.6666667, 4.0], (1, 1, 4, 4))
    assert torch.allclose(custom, exp_custom, rtol=1e-5, atol=1e-5)
    assert torch.allclose(burn, exp_burn, rtol=1e-5, atol=1e-5)
    assert not torch.allclose(custom, burn, rtol=1e-5, atol=1e-5)


def test_scale_resize_outputs_match_expected():
    # reference: depth_pro/interpolate.rs:221-248
    x = t4([4.0, 1.0, 0.0, 2.0], (1, 1, 2, 2))
    custom = R.resize_bilinear_scale(x, (1.5, 0.5), InterpolationMethod.CUSTOM)
    burn = R.resize_bilinear_scale(x, (1.5, 0.5), InterpolationMethod.BURN)
    assert custom.shape == (1, 1, 3, 1) and burn.shape == (1, 1, 3, 1)
    assert torch.allclose(custom.flatten(), torch.tensor([2.5, 1.75, 1.0]), rtol=1e-5, atol=1e-5)
    assert torch.allclose(burn.flatten(), torch.tensor([4.0, 2.0, 0.0]), rtol=1e-5, atol=1e-5)


def test_compute_output_size():
    # reference: depth_pro/interpolate.rs:24-27
    assert R.compute_output_size(2, 1.5) == 3
    assert R.compute_output_size(2, 0.5) == 1
    assert R.compute_output_size(1, 0.25) == 1  # max(.., 1)
    assert R.compute_output_size(1536, 0.5) == 768
    assert R.compute_output_size(1536, 0.25) == 384


def test_resize_identity_when_same_size():
    # reference: depth_pro/interpolate.rs:61-63
    x = torch.rand(2, 3, 5, 7)
    assert R.resize_bilinear(x, (5, 7)) is x


def test_rgb_to_input_tensor_normalizes_channels():
    # reference: src/inference.rs:145-173
    t = R.rgb_to_input_tensor(bytes([0, 255, 128, 255, 0, 128]), 1, 2)
    assert tuple(t.shape) == (1, 3, 2, 1)
    exp = [-2.1179039, 2.2489083, 2.4285715, -2.0357141, 0.42649257, 0.42649257]
    for v, e in zip(t.flatten().tolist(), exp):
        assert abs(v - e) < 1e-2
        assert abs(v - e) < 1e-6  # the oracle is in fact exact to f32 rounding


def test_rgb_to_input_tensor_rejects_invalid_length():
    # reference: src/inference.rs:175-181
    with pytest.raises(ValueError):
        R.rgb_to_input_tensor(bytes(5), 1, 2)


def test_split_merge_roundtrip_without_overlap():
    # reference: layers/encoder.rs:501-519 (128-window preset -> img 512, 16 tiles, padding 0)
    v = vit_config_from_preset("dinov2l16_128")
    size = v.img_size * 4
    x = torch.arange(3 * size * size, dtype=torch.float32).reshape(1, 3, size, size)
    tiles, steps, stride = R.split(x, v.img_size, 0.0)
    assert steps * steps == 16
    pad = R.feature_padding(v.img_size, stride, v.grid_size())
    merged = R.merge(tiles, 1, pad)
    assert torch.allclose(merged, x, rtol=1e-5, atol=1e-5)


def test_merge_overlapping_layout_matches_expected():
    # reference: layers/encoder.rs:521-586
    B, C, fs, steps, pad = 1, 2, 8, 5, 1
    n = B * steps * steps
    patches = torch.arange(n, dtype=torch.float32).reshape(n, 1, 1, 1).expand(n, C, fs, fs).contiguous()
    merged = R.merge(patches, B, pad)
    oh, ow = merged.shape[2], merged.shape[3]
    exp = torch.full((B, C, oh, ow), -1.0)
    for b in range(B):
        for j in range(steps):
            for i in range(steps):
                idx = B * (j * steps + i) + b
                top = 0 if j == 0 else pad
                bottom = 0 if j == steps - 1 else pad
                left = 0 if i == 0 else pad
                right = 0 if i == steps - 1 else pad
                sh, sw = fs - top - bottom, fs - left - right
                by = j * (fs - 2 * pad) + (0 if j == 0 else pad)
                bx = i * (fs - 2 * pad) + (0 if i == 0 else pad)
                exp[b, :, by:by + sh, bx:bx + sw] = float(idx)
    assert merged.shape == exp.shape == (1, 2, 32, 32)
    assert torch.equal(merged, exp)


def test_split_geometry_default_config():
    # 384 window at 1536 -> stride 288, 5 steps; at 768 -> stride 192, 3 steps
    assert R.split_geometry(1536, 384, 0.25) == (288, 5)
    assert R.split_geometry(768, 384, 0.5) == (192, 3)
    assert R.feature_padding(384, 288, 24) == 3
    assert R.feature_padding(384, 192, 24) == 6
    # CI-size preset: paddings 1 and 2
    assert R.split_geometry(512, 128, 0.25) == (96, 5)
    assert R.feature_padding(128, 96, 8) == 1
    assert R.feature_padding(128, 64, 8) == 2


def test_vit_patch_count_matches_grid():
    # reference: layers/vit.rs:76-96 (shape-only pin), run on the tiny preset to stay fast
    from burn_depth_amd import weights as Wt
    cfg = DepthProConfig.tiny_test()
    W = R.weights_to_torch(Wt.generate_depth_pro_weights(cfg, 0))
    v = cfg.patch_vit()
    out, hooks = R.vit_forward(torch.ones(1, 3, v.img_size, v.img_size), W, "encoder.patch_encoder", v,
                               v.encoder_feature_layer_ids)
    assert out.shape[1] == v.grid_size() ** 2
    assert len(hooks) == 4 and hooks[0].shape[1] == v.num_tokens


def test_fovy_from_fovx():
    # reference: depth_pro/mod.rs:370-414 -- approximation stays within 1e-2 rad of the exact value
    for deg, h, w in [(56.0, 360, 540), (30.0, 1536, 1536), (100.0, 540, 360)]:
        fx = torch.tensor([math.radians(deg)])
        got = R.fovy_from_fovx_rad(fx, h, w).item()
        exact = 2 * math.atan(h / w * math.tan(math.radians(deg) / 2))
        assert abs(got - exact) < 1e-2  # atan approx err <= ~3.8e-3, doubled
    # square image, |x|<=1 branch: value of the rational formula itself
    fx = torch.tensor([1.0])
    t = math.tan(0.5)
    assert abs(R.fovy_from_fovx_rad(fx, 8, 8).item() - 2 * t * (math.pi / 4 + 0.273 * (1 - t))) < 1e-6


def test_infer_shapes_zeros_input():
    # reference: src/lib.rs:179-195 (depth [1,S,S], focal [1]) on the tiny preset
    from burn_depth_amd import weights as Wt
    cfg = DepthProConfig.tiny_test()
    W = R.weights_to_torch(Wt.generate_depth_pro_weights(cfg, 0))
    S = cfg.img_size()
    out = R.infer(torch.zeros(1, 3, S, S), W, cfg)
    assert tuple(out["depth"].shape) == (1, S, S)
    assert tuple(out["focallength_px"].shape) == (1,)
    assert torch.isfinite(out["depth"]).all()


def test_layernorm_fold_restatement_is_the_same_function():
    """oracle/depth_pro_ref.py::LN_FOLD_EMULATION restates the engine's LayerNorm fold (csrc/kernels/gemm.h GemmParams::ln_*:
    LN(x) W^T + b = rstd (round(gamma x) W^T - mu c) + d) for the operand-rounding oracle only. It must be the SAME function of x:
    ignored by the fp32 oracle (q = identity), equal to the unfolded form when no operand is rounded (fp64 here, to 1e-12), and within
    the rounding of the operand type when operands are rounded (bf16: both forms sit at the same distance from the fp32 oracle).
    Block order: burn_dino (called from /root/reference/src/model/depth_pro/layers/encoder.rs:346-348)."""
    import torch
    from burn_depth_amd import weights as Wt
    from burn_depth_amd.config import DepthProConfig
    from oracle import depth_pro_ref as R
    cfg = DepthProConfig.tiny_test()
    v = cfg.patch_vit()
    W = R.weights_to_torch(Wt.generate_depth_pro_weights(cfg, 0, Wt.INIT_PARITY))
    g = torch.Generator().manual_seed(1)
    for k in list(W):  # non-trivial gamma / beta so that c, d matter
        if k.endswith((".norm1.gamma", ".norm2.gamma")):
            W[k] = W[k] * (0.5 + torch.rand(W[k].shape, generator=g))
        if k.endswith((".norm1.beta", ".norm2.beta")):
            W[k] = W[k] + 0.3 * torch.randn(W[k].shape, generator=g)
    x = torch.randn(2, 3, v.img_size, v.img_size, generator=g)
    with torch.no_grad():
        ref, ref_h = R.vit_forward(x, W, "encoder.patch_encoder", v, (1, 2))
        prev = R.LN_FOLD_EMULATION
        try:
            R.LN_FOLD_EMULATION = True
            same, _ = R.vit_forward(x, W, "encoder.patch_encoder", v, (1, 2))              # identity quantiser: the flag is ignored
            noround = lambda t: t                                                          # noqa: E731  (a quantiser that rounds nothing)
            Wd = {k: t.double() for k, t in W.items()}
            fold64, _ = R.vit_forward(x.double(), Wd, "encoder.patch_encoder", v, (1, 2), q=noround)
            fold_bf, _ = R.vit_forward(x, W, "encoder.patch_encoder", v, (1, 2), q=R.bf16_round)
            R.LN_FOLD_EMULATION = False
            plain64, _ = R.vit_forward(x.double(), Wd, "encoder.patch_encoder", v, (1, 2), q=noround)
            plain_bf, _ = R.vit_forward(x, W, "encoder.patch_encoder", v, (1, 2), q=R.bf16_round)
        finally:
            R.LN_FOLD_EMULATION = prev
    assert torch.equal(same, ref)
    assert (fold64 - plain64).abs().max().item() <= 1e-11 * plain64.abs().max().item()
    scale = ref.abs().max().item()
    e_fold, e_plain = (fold_bf - ref).abs().max().item() / scale, (plain_bf - ref).abs().max().item() / scale
    assert 0 < e_fold <= 3 * e_plain + 1e-3 and e_plain <= 3 * e_fold + 1e-3, (e_fold, e_plain)
    assert not torch.equal(fold_bf, plain_bf)
