"""CPU study (oracle only, no GPU): which operand roundings the accurate fast mode can afford.

Runs the fp32 oracle of Depth Pro on the ViT-L CI preset (512^2, `DepthProConfig.small_test`, 24 blocks of width 1024 --
the same depth of rounding as the full-size model) with a quantiser per operand SITE and reports the depth error against
the un-quantised run. Weights are rounded to f16 first (the reference's checkpoints are f16, mod.rs:206), so weights
are exact MFMA operands in every variant.

Sites: patch (patch-embed input), ln (LayerNorm outputs -> qkv / fc1), q, k, v, p (softmax probabilities), ao (attention
output -> proj), h (GELU output -> fc2), conv (every operand of the encoder tail / decoder / head convolutions).
Quantisers: f16 = one IEEE half; split = hi + lo with hi = f16(x), lo = f16(x - hi) (two MFMAs per product).

usage: precision_study.py                 -> all-f16, all-split, then each site alone at f16 with the rest split
       precision_study.py p=f16 h=f16     -> one run with the named sites at f16, the rest split
"""
from __future__ import annotations

import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from burn_depth_amd import weights as Wt  # noqa: E402
from burn_depth_amd.config import DepthProConfig  # noqa: E402
from oracle import depth_pro_ref as R  # noqa: E402

SITES = ("patch", "ln", "q", "k", "v", "p", "ao", "h", "conv")


def split_round(x):
    hi = R.f16_round(x)
    return hi + R.f16_round(x - hi)


QUANT = {"f16": R.f16_round, "split": split_round, "exact": R.identity}


class SiteQ:
    """Callable used as the oracle's `q` (the convolution sites) that also carries the ViT's per-site quantisers."""

    def __init__(self, modes):
        self.m = {s: QUANT[modes.get(s, "split")] for s in SITES}

    def __call__(self, x):
        return self.m["conv"](x)


def vit_chunk_sites(x, W, prefix, v, hook_ids, q, fp8=False):
    """oracle/depth_pro_ref.py::_vit_forward_chunk with one quantiser per operand site (same arithmetic otherwise)."""
    if not isinstance(q, SiteQ):
        return ORIG_CHUNK(x, W, prefix, v, hook_ids, q, fp8)
    m = q.m
    B = x.shape[0]
    D, Hn, hd = v.embed_dim, v.num_heads, v.head_dim
    p = lambda n: W[f"{prefix}.{n}"]
    tok = F.conv2d(m["patch"](x), p("patch_embed.proj.weight"), p("patch_embed.proj.bias"), stride=v.patch_size)
    tok = tok.flatten(2).transpose(1, 2)
    xs = torch.cat([p("cls_token").expand(B, 1, D), tok], 1) + p("pos_embed")
    N = xs.shape[1]
    scale = hd ** -0.5
    hooks = []
    for i in range(v.depth):
        b = f"blocks.{i}"
        xn = m["ln"](F.layer_norm(xs, (D,), p(f"{b}.norm1.gamma"), p(f"{b}.norm1.beta"), v.ln_eps))
        qkv = F.linear(xn, p(f"{b}.attn.qkv.weight"), p(f"{b}.attn.qkv.bias")).reshape(B, N, 3, Hn, hd).permute(2, 0, 3, 1, 4)
        qq = m["q"](qkv[0] * R.ATTN_QSCALE) / R.ATTN_QSCALE
        kk, vv = m["k"](qkv[1]), m["v"](qkv[2])
        s = (qq @ kk.transpose(-2, -1)) * scale
        pu = torch.exp(s - s.amax(-1, keepdim=True))
        o = (m["p"](pu) @ vv) / pu.sum(-1, keepdim=True)
        o = m["ao"](o.transpose(1, 2).reshape(B, N, D))
        xs = xs + p(f"{b}.ls1.gamma") * F.linear(o, p(f"{b}.attn.proj.weight"), p(f"{b}.attn.proj.bias"))
        xn = m["ln"](F.layer_norm(xs, (D,), p(f"{b}.norm2.gamma"), p(f"{b}.norm2.beta"), v.ln_eps))
        h = m["h"](F.gelu(F.linear(xn, p(f"{b}.mlp.fc1.weight"), p(f"{b}.mlp.fc1.bias"))))
        xs = xs + p(f"{b}.ls2.gamma") * F.linear(h, p(f"{b}.mlp.fc2.weight"), p(f"{b}.mlp.fc2.bias"))
        for hid in hook_ids:
            if hid == i:
                hooks.append(xs.clone())
    xn = F.layer_norm(xs, (D,), p("norm.gamma"), p("norm.beta"), v.ln_eps)
    return xn[:, 1:], hooks


ORIG_CHUNK = R._vit_forward_chunk
R._vit_forward_chunk = vit_chunk_sites


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = DepthProConfig.small_test()
    Wn = Wt.generate_depth_pro_weights(cfg, 0, Wt.INIT_PARITY)
    W = {k: R.f16_round(v) for k, v in R.weights_to_torch(Wn).items()}
    torch.manual_seed(0)
    x = (torch.rand(1, 3, 512, 512) - 0.45) / 0.225
    t0 = time.time()
    ref = R.infer(x, W, cfg)
    print(f"fp32 oracle: {time.time() - t0:.1f} s, depth in [{ref['depth'].min():.3f}, {ref['depth'].max():.3f}]", flush=True)

    def run(label, modes):
        out = R.infer(x, W, cfg, q=SiteQ(modes))
        d, rd = out["depth"], ref["depth"]
        err = (d - rd).abs()
        rel = err / rd.abs()
        print(f"{label:14s} depth L_inf {err.max():.3e}  max-rel {rel.max():.3e}  p99.9 {rel.flatten().kthvalue(int(rel.numel() * 0.999)).values:.3e}"
              f"  mean-rel {rel.mean():.3e}  fovx abs {abs(out['fovx_deg'].item() - ref['fovx_deg'].item()):.2e}", flush=True)

    args = [a for a in sys.argv[1:] if "=" in a]
    if args:
        modes = dict(a.split("=") for a in args)
        run(" ".join(args), modes)
        return
    run("all f16", {s: "f16" for s in SITES})
    run("all split", {})
    for s in SITES:
        run(f"{s}=f16", {s: "f16"})


if __name__ == "__main__":
    main()
