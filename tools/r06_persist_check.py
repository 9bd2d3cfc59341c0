"""Persistent fc1 kernel (gemm256p_kernel; md_debug_gemm_persistent): bit-identity against the one-tile kernel on a stand-alone GELU linear
layer with >= 1024 tiles (a partial last m-tile included) and on whole DepthPro::infer calls (LayerNorm fold on), bf16 / f16 / f16x2."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from burn_depth_amd import _lib, ops  # noqa: E402
from burn_depth_amd import weights as Wt  # noqa: E402
from burn_depth_amd.config import DepthProConfig  # noqa: E402
from burn_depth_amd.depth_pro import DepthPro, Device  # noqa: E402


def main():
    dev = Device(0)
    lib = _lib.load()
    ok = True
    g = torch.Generator().manual_seed(5)
    M, N, K = 256 * 64 + 100, 4096, 1024
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).cuda()
    b = (torch.randn(N, generator=g) * 0.1).cuda()
    for prec in (0, 3, 4):
        xx, ww = (x.half().float(), w.half().float()) if prec == 4 else (x, w)
        lib.md_debug_gemm_persistent(0)
        ref = ops.linear(dev, xx, ww, b, act=2, precision=prec, tile=_lib.TILE_256x256, storage_out=True)
        lib.md_debug_gemm_persistent(15)
        got = ops.linear(dev, xx, ww, b, act=2, precision=prec, tile=_lib.TILE_256x256, storage_out=True)
        got2 = ops.linear(dev, xx, ww, b, act=2, precision=prec, tile=_lib.TILE_256x256, storage_out=True)
        lib.md_debug_gemm_persistent(0)
        same = torch.equal(ref, got) and torch.equal(got, got2)
        print(f"linear prec {prec}: persistent == one-tile: {same}  (max |diff| {(ref - got).abs().max().item():.3e})", flush=True)
        ok = ok and same
    for prec, fold in ((0, 1), (0, 0), (3, 1), (4, 1)):   # (the QKV projection runs the tile loop in the one-plane types, folded and unfolded)
        cfg = DepthProConfig()
        cfg.precision = prec
        cfg.max_batch = 2
        m = DepthPro.new(dev, cfg, seed=0, init_scheme=Wt.INIT_PARITY)
        m.set_option("ln_fold", fold)
        if prec == 4:
            m.round_weights_to_f16()
        torch.manual_seed(1)
        xi = torch.randn(2, 3, 1536, 1536, device="cuda")
        lib.md_debug_gemm_persistent(0)
        a = m.infer(xi).depth.clone()
        lib.md_debug_gemm_persistent(15)
        bb = m.infer(xi).depth.clone()
        cc = m.infer(xi).depth.clone()
        lib.md_debug_gemm_persistent(0)
        same = torch.equal(a, bb) and torch.equal(bb, cc)
        print(f"DepthPro::infer [2,3,1536,1536] prec {prec} (fold {m.query('ln_fold_active')}): persistent == one-tile: {same}  max |diff| {(a - bb).abs().max().item():.3e}", flush=True)
        ok = ok and same
        m.destroy()
    # proj / fc2 reach 1024 tiles (gemm256r_kernel) from B = 4 and 2048 (its start offset between the halves of an XCD's workgroups) from B = 7
    for prec in (0, 3, 4):
        cfg = DepthProConfig()
        cfg.precision = prec
        cfg.max_batch = 8
        m = DepthPro.new(dev, cfg, seed=0, init_scheme=Wt.INIT_PARITY)
        if prec == 4:
            m.round_weights_to_f16()
        torch.manual_seed(2)
        xi = torch.randn(8, 3, 1536, 1536, device="cuda")
        lib.md_debug_gemm_persistent(3)
        a = m.infer(xi).depth.clone()
        lib.md_debug_gemm_persistent(7)
        a7 = m.infer(xi).depth.clone()  # (15 adds the decoder's lean 3 x 3 convolutions)
        lib.md_debug_gemm_persistent(15)
        bb = m.infer(xi).depth.clone()
        lib.md_debug_gemm_stagger(0, 0)
        cc = m.infer(xi).depth.clone()
        lib.md_debug_gemm_stagger(0, 2000)
        lib.md_debug_gemm_persistent(0)
        dd = m.infer(xi).depth.clone()
        lib.md_debug_gemm_persistent(15)
        same = torch.equal(a, bb) and torch.equal(bb, cc) and torch.equal(cc, dd) and torch.equal(a, a7)
        print(f"DepthPro::infer [8,3,1536,1536] prec {prec}: read-modify-write tile loop (with / without its start offset) == one-tile kernels: {same}  "
              f"max |diff| {(a - bb).abs().max().item():.3e} {(a - cc).abs().max().item():.3e} {(a - dd).abs().max().item():.3e}", flush=True)
        ok = ok and same
        m.destroy()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
