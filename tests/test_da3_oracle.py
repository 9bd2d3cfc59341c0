"""Oracle checks for the Depth-Anything-v3 mono head: the reference's shape smoke test
(depth_anything3/mod.rs:634-642) on the reduced variant, and properties of the UV position table
(dpt.rs:835-932) that the engine's C++ twin must reproduce."""
import numpy as np
import torch

from burn_depth_amd import weights as Wt
from burn_depth_amd.config import DepthAnything3Config
from oracle import da3_ref as D
from oracle import depth_pro_ref as R


def test_depth_anything3_emits_depth_tensor():
    # reference: depth_anything3/mod.rs:634-642 ([1,3,518,518] zeros -> depth [1,518,518]); tiny variant here
    cfg = DepthAnything3Config.tiny_test()
    W = R.weights_to_torch(Wt.generate_da3_weights(cfg, 0))
    out = D.infer(torch.zeros(1, 3, cfg.image_size, cfg.image_size), W, cfg)
    assert tuple(out["depth"].shape) == (1, cfg.image_size, cfg.image_size)
    assert torch.isfinite(out["depth"]).all() and (out["depth"] > 0).all()  # exp activation


def test_rejects_sizes_not_divisible_by_patch():
    # reference: depth_anything3/mod.rs:509-520 (assert)
    cfg = DepthAnything3Config.tiny_test()
    W = R.weights_to_torch(Wt.generate_da3_weights(cfg, 0))
    try:
        D.infer(torch.zeros(1, 3, 71, 70), W, cfg)
    except ValueError:
        return
    raise AssertionError("expected an error")


def test_positional_embedding_layout():
    # dpt.rs:835-890: first C/2 channels depend on the x coordinate, the rest on y; index x_idx*height + y_idx
    C, h, w = 8, 3, 5
    t = D.build_positional_embedding(C, h, w, 70, 42).reshape(C, h * w)
    aspect = np.float32(70) / np.float32(42)
    diag = np.sqrt(aspect * aspect + 1)
    xs = np.linspace(-aspect / diag * (w - 1) / w, aspect / diag * (w - 1) / w, w)
    ys = np.linspace(-1 / diag * (h - 1) / h, 1 / diag * (h - 1) / h, h)
    for xi in range(w):
        for yi in range(h):
            pix = xi * h + yi
            assert abs(t[0, pix] - np.sin(xs[xi])) < 1e-6          # omega = 100^0 = 1
            assert abs(t[C // 2, pix] - np.sin(ys[yi])) < 1e-6
            assert abs(t[C // 4, pix] - np.cos(xs[xi])) < 1e-6     # second half of the x block: cos
    # square map: the reference's flat index makes the table the transpose of the natural (y, x) layout
    s = D.build_positional_embedding(4, 4, 4, 64, 64).reshape(4, 4, 4)
    assert np.allclose(s[0], s[0][:, :1])        # rows = x index: constant along the width axis
    assert not np.allclose(s[0], s[0][:1, :])


def test_align_corners_true_resize_identity():
    x = torch.rand(1, 2, 5, 7)
    assert D.resize_bilinear(x, (5, 7)) is x
    y = D.resize_bilinear(x, (9, 13))
    assert torch.allclose(y[..., 0, 0], x[..., 0, 0]) and torch.allclose(y[..., -1, -1], x[..., -1, -1])


def test_fp8_operand_emulation_properties():
    """The MD_PREC_FP8 emulation used to check the engine: static-scale / per-row e4m3 quantisers."""
    import torch
    from oracle import depth_pro_ref as R
    x = torch.tensor([0.0, 1e-4, 0.5, 7.9, 8.1, 20.0, -9.0])
    q = R.fp8_static(x, R.FP8_ACT_SCALE)
    assert q[0] == 0 and abs(q[2] - 0.5) < 0.5 / 16 and q[4] == q[5] == 8.0 and q[6] == -8.0   # saturates at 448 * scale
    assert abs(q[3] - 7.9) <= 8.0 / 16                                                              # 3 mantissa bits
    w = torch.randn(5, 64)
    w[3] = 0
    wq = R.fp8_rows(w)
    assert torch.equal(wq[3], torch.zeros(64))
    assert torch.allclose(wq.abs().amax(1)[[0, 1, 2, 4]], w.abs().amax(1)[[0, 1, 2, 4]], rtol=1e-6)  # the row maximum is exact
    assert ((wq - w).abs() <= w.abs().amax(1, keepdim=True) / 16 + 1e-9).all()
    qn, qo, qh, qw = R.linear_quantisers(R.identity, False)
    assert qn is R.identity and qw is R.identity


def test_fp8_mode_stays_close_to_fp32_on_the_oracle():
    import torch
    from burn_depth_amd import weights as Wt
    from burn_depth_amd.config import DepthAnything3Config
    from oracle import da3_ref as D3, depth_pro_ref as R
    cfg = DepthAnything3Config.tiny_dual_test()
    W = R.weights_to_torch(Wt.generate_da3_weights(cfg, 0, Wt.INIT_PARITY))
    torch.manual_seed(1)
    x = torch.randn(1, 3, 70, 70)
    with torch.no_grad():
        a = D3.infer(x, W, cfg)["depth"]
        b = D3.infer(x, W, cfg, q=R.bf16_round, fp8=True)["depth"]
    rel = ((a - b).abs() / a.abs())
    assert rel.mean() < 3e-2 and rel.max() < 0.3


def test_infer_from_tokens_is_the_head_of_infer():
    # reference: DepthAnything3::infer_from_tokens (mod.rs:389-469) = forward_raw / forward_dual on given hook tokens; a leading row
    # (patch_token_start = 1) is dropped when the token count is not the patch count (mod.rs:419-424)
    for cfg in (DepthAnything3Config.tiny_test(), DepthAnything3Config.tiny_dual_test()):
        W = R.weights_to_torch(Wt.generate_da3_weights(cfg, 0, Wt.INIT_PARITY))
        x = torch.randn(2, 3, cfg.image_size, cfg.image_size, generator=torch.Generator().manual_seed(4))
        full = D.infer(x, W, cfg, debug=True)
        toks = full["debug"]["hooks"]
        assert torch.equal(D.infer_from_tokens(toks, W, cfg, 70, 70)["depth"], full["depth"])
        lead = [torch.cat([torch.full((2, 1, t.shape[2]), 7.0), t], 1) for t in toks]
        out = D.infer_from_tokens(lead, W, cfg, 70, 70)
        assert torch.equal(out["depth"], full["depth"]) and "pose_encoding" not in out
        try:
            D.infer_from_tokens([t[:, :-1] for t in toks], W, cfg, 70, 70)
        except ValueError:
            continue
        raise AssertionError("expected an error")


def test_camera_pose_encoding_round_trips_through_the_decoder_formulas():
    # extri_intri_to_pose_encoding (camera.rs:236-279) followed by pose_encoding_to_extri_intri (camera.rs:281-358) returns the
    # extrinsics (quaternion branches: trace > 0, x, y, z largest) and, up to the reference's polynomial atan, the focal lengths
    import math

    def rot(axis, ang):
        a = np.asarray(axis, float)
        a /= np.linalg.norm(a)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        return np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * K @ K
    Rs = [rot((1, 2, 3), 0.4), rot((1, .1, .1), 3.0), rot((.1, 1, .1), 3.0), rot((.1, .1, 1), 3.0)]
    E = torch.tensor(np.stack([np.concatenate([r, [[0.3], [-0.2], [0.5]]], 1) for r in Rs]), dtype=torch.float32).reshape(1, 4, 3, 4)
    K = torch.tensor([[100.0, 0, 35], [0, 20.0, 35], [0, 0, 1]]).expand(1, 4, 3, 3)   # fx > W/2 (small branch), fy < H/2 (reciprocal branch)
    pe = D.pose_encoding(E, K, 70, 70).reshape(4, 9)
    assert [int(r.abs().argmax()) for r in pe[:, 3:7]] == [3, 0, 1, 2]
    t, (qx, qy, qz, qw) = pe[:, :3], pe[:, 3:7].unbind(1)
    Rm = torch.stack([torch.stack([1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)], 1),
                      torch.stack([2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)], 1),
                      torch.stack([2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)], 1)], 1)
    Rt = Rm.transpose(1, 2)
    back = torch.cat([Rt, -(Rt @ t[:, :, None])], 2)
    assert (back - E[0]).abs().max() < 2e-5
    fy = 35.0 / torch.tan(pe[:, 7] * 0.5)
    fx = 35.0 / torch.tan(pe[:, 8] * 0.5)
    assert ((fx - 100.0).abs() / 100.0).max() < 1e-2 and ((fy - 20.0).abs() / 20.0).max() < 1e-2   # approx_atan_positive, camera.rs:516-536
