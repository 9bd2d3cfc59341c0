"""The two commit-time layer compositions of the depth head, restated in PyTorch on the CPU and checked against the
uncomposed layers (Burn = PyTorch conv semantics). The index conventions below are the ones of `compose_head_kernel`,
`compose_head_bias_kernel` and `compose_c1c3_kernel` (burn_depth_amd/csrc/md_engine.hip); the GPU parity tests then check
the kernels themselves end to end.

  * head.deconv (ConvTranspose2d k2 s2, bias) -> head.conv1 (Conv2d 3x3 pad 1, bias), mod.rs:106-107: one 3x3 convolution
    with 4 x Cout columns (one group per output parity) on the deconv's input grid + nine position-class bias vectors;
  * decoder.fusions.0.out_conv (Conv2d 1x1, bias) -> head.conv0 (Conv2d 3x3 pad 1, bias), decoder.rs:137 -> mod.rs:105: one
    3x3 convolution + the same nine bias classes on its own grid.
"""
import torch
import torch.nn.functional as F


def compose_head(wd, w1):
    """wd [Cin, Cmid, 2, 2] (ConvTranspose2d), w1 [Cout, Cmid, 3, 3] -> wc [4*Cout, Cin, 3, 3]; column group q = 2*py + px."""
    cin, cmid = wd.shape[:2]
    cout = w1.shape[0]
    wc = torch.zeros(4 * cout, cin, 3, 3, dtype=wd.dtype)
    for py in range(2):
        for px in range(2):
            q = 2 * py + px
            for u in range(3):
                a, dy = (py + u + 1) // 2 - 1, (py + u + 1) & 1
                for v in range(3):
                    b, dx = (px + v + 1) // 2 - 1, (px + v + 1) & 1
                    # sum over mid of w1[co, mid, u, v] * wd[ci, mid, dy, dx]
                    wc[q * cout:(q + 1) * cout, :, a + 1, b + 1] += torch.einsum("om,im->oi", w1[:, :, u, v], wd[:, :, dy, dx])
    return wc


def bias_classes(w3, b_in, b_out):
    """w3 [Cout, Cmid, 3, 3], b_in [Cmid] (bias of the layer in front), b_out [Cout] -> [9, Cout], class = 3*ry + rx."""
    out = torch.zeros(9, w3.shape[0], dtype=w3.dtype)
    for ry in range(3):
        for rx in range(3):
            acc = b_out.clone()
            for u in range(3):
                if (ry == 0 and u == 0) or (ry == 2 and u == 2):
                    continue
                for v in range(3):
                    if (rx == 0 and v == 0) or (rx == 2 and v == 2):
                        continue
                    acc = acc + w3[:, :, u, v] @ b_in
            out[3 * ry + rx] = acc
    return out


def class_map(H, W):
    ry = torch.ones(H, dtype=torch.long)
    ry[0], ry[-1] = 0, 2
    rx = torch.ones(W, dtype=torch.long)
    rx[0], rx[-1] = 0, 2
    return 3 * ry[:, None] + rx[None, :]


def test_deconv_then_conv3x3_is_one_conv_with_parity_columns_and_position_class_biases():
    g = torch.Generator().manual_seed(0)
    B, cin, cmid, cout, H, W = 2, 5, 6, 4, 7, 5
    x = torch.randn(B, cin, H, W, generator=g, dtype=torch.float64)
    wd = torch.randn(cin, cmid, 2, 2, generator=g, dtype=torch.float64)
    bd = torch.randn(cmid, generator=g, dtype=torch.float64)
    w1 = torch.randn(cout, cmid, 3, 3, generator=g, dtype=torch.float64)
    b1 = torch.randn(cout, generator=g, dtype=torch.float64)
    want = F.conv2d(F.conv_transpose2d(x, wd, bd, stride=2), w1, b1, padding=1)  # [B, cout, 2H, 2W]

    y = F.conv2d(x, compose_head(wd, w1), None, padding=1)  # [B, 4*cout, H, W]
    got = torch.zeros_like(want)
    for py in range(2):
        for px in range(2):
            q = 2 * py + px
            got[:, :, py::2, px::2] = y[:, q * cout:(q + 1) * cout]
    b9 = bias_classes(w1, bd, b1)                       # classes on the OUTPUT (2H x 2W) grid
    got = got + b9[class_map(2 * H, 2 * W)].permute(2, 0, 1)[None]
    assert torch.allclose(got, want, rtol=0, atol=1e-11)


def test_conv1x1_then_conv3x3_is_one_conv_with_position_class_biases():
    g = torch.Generator().manual_seed(1)
    B, cin, cmid, cout, H, W = 2, 6, 5, 3, 6, 9
    x = torch.randn(B, cin, H, W, generator=g, dtype=torch.float64)
    w1 = torch.randn(cmid, cin, 1, 1, generator=g, dtype=torch.float64)
    b1 = torch.randn(cmid, generator=g, dtype=torch.float64)
    w3 = torch.randn(cout, cmid, 3, 3, generator=g, dtype=torch.float64)
    b3 = torch.randn(cout, generator=g, dtype=torch.float64)
    want = F.conv2d(F.conv2d(x, w1, b1), w3, b3, padding=1)

    wc = torch.einsum("omuv,mi->oiuv", w3, w1[:, :, 0, 0])  # compose_c1c3_kernel
    b9 = bias_classes(w3, b1, b3)
    interior = F.conv2d(x, wc, b9[4], padding=1)             # what the convolution itself adds
    fix = (b9 - b9[4])[class_map(H, W)].permute(2, 0, 1)     # border_bias_fix_kernel
    assert torch.allclose(interior + fix[None], want, rtol=0, atol=1e-11)
    assert fix[:, 1:-1, 1:-1].abs().max() == 0               # interior pixels untouched


def test_two_k2s2_deconvolutions_are_one_k4s4():
    """encoder.rs:146-152: ConvTranspose2d(k2, s2, no bias) twice = ConvTranspose2d(k4, s4) on the weight product
    W''[ci][co][2 dy1 + dy2][2 dx1 + dx2] = sum_m Wa[ci][m][dy1][dx1] * Wb[m][co][dy2][dx2] (compose_deconv_pair_kernel)."""
    g = torch.Generator().manual_seed(2)
    B, cin, cmid, cout, H, W = 2, 4, 5, 3, 6, 7
    x = torch.randn(B, cin, H, W, generator=g, dtype=torch.float64)
    wa = torch.randn(cin, cmid, 2, 2, generator=g, dtype=torch.float64)
    wb = torch.randn(cmid, cout, 2, 2, generator=g, dtype=torch.float64)
    want = F.conv_transpose2d(F.conv_transpose2d(x, wa, None, stride=2), wb, None, stride=2)
    wc = torch.zeros(cin, cout, 4, 4, dtype=torch.float64)
    for ty in range(4):
        for tx in range(4):
            wc[:, :, ty, tx] = wa[:, :, ty >> 1, tx >> 1] @ wb[:, :, ty & 1, tx & 1]
    got = F.conv_transpose2d(x, wc, None, stride=4)
    assert torch.allclose(got, want, rtol=0, atol=1e-11)
