"""Determinism / exactness soak of the MFMA GEMM family (256x256 and 128x128 tiles; bf16, f16 and fp32 operands; dense,
3x3 convolution, k2s2 deconvolution): small-integer operands (every product and partial sum exact in fp32), repeated launches per shape, fp32-store and bf16-store epilogues,
bit-exact against the CPU result."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from burn_depth_amd import _lib, ops  # noqa: E402
from burn_depth_amd.depth_pro import Device  # noqa: E402

SHAPES = [(21349, 1024, 1024), (5000, 3072, 1024), (3000, 1024, 4096), (2500, 512, 128), (4096, 256, 2304), (700, 264, 64)]


def run(dev, iters=25, verbose=False):
    g = torch.Generator().manual_seed(1)
    bad = []
    for (M, N, K) in SHAPES:
        x = torch.randint(-3, 4, (M, K), generator=g).float()
        w = torch.randint(-2, 3, (N, K), generator=g).float()
        b = torch.randint(-5, 6, (N,), generator=g).float()
        want = (x.double() @ w.double().t() + b.double()).float().cuda()  # exact small integers
        want16 = want.to(torch.bfloat16).float()  # the engine's bf16 store epilogue rounds to nearest even as well
        xc, wc, bc = x.cuda(), w.cuda(), b.cuda()
        for it in range(iters):
            d = (ops.linear(dev, xc, wc, bc, 0, 0, _lib.TILE_256x256) - want).abs().max().item()
            d16 = (ops.linear(dev, xc, wc, bc, 0, 0, _lib.TILE_256x256, storage_out=True) - want16).abs().max().item()
            if d != 0.0 or d16 != 0.0:
                bad.append((M, N, K, it, d, d16))
        # the two-stage kernel (128x128 tiles, 16x16 MFMA) on the same operands, and the f16 / fp32 operand forms of both
        for it in range(max(1, iters // 4)):
            for tile in (_lib.TILE_128x128, _lib.TILE_256x256):
                for prec in (0, 3, 1):
                    if tile == _lib.TILE_256x256 and prec == 0:
                        continue
                    d = (ops.linear(dev, xc, wc, bc, 0, prec, tile) - want).abs().max().item()
                    if d != 0.0:
                        bad.append((M, N, K, it, "tile", tile, "prec", prec, d))
        if verbose:
            print("shape", M, N, K, "ok" if not bad else "MISMATCH", flush=True)
    # the 64 x 64 kernel with its contraction split over two / four wave groups of the workgroup (small launches behind a long k-loop:
    # gemm_kernel's KSPLIT): the groups' partial accumulators are summed through LDS -- exact on small integers, bf16 and f16 operands
    split_before = _lib.load().md_gemm_ksplit_launches()
    for (M, N, K) in [(1370, 384, 1536), (361, 384, 3456), (300, 128, 1024), (65, 68, 2048), (1369, 64, 1728), (1370, 96, 768)]:
        x = torch.randint(-3, 4, (M, K), generator=g).float()
        w = torch.randint(-2, 3, (N, K), generator=g).float()
        b = torch.randint(-5, 6, (N,), generator=g).float()
        want = (x.double() @ w.double().t() + b.double()).float().cuda()
        xc, wc, bc = x.cuda(), w.cuda(), b.cuda()
        for it in range(max(1, iters // 4)):
            for prec in (0, 3):
                d = (ops.linear(dev, xc, wc, bc, 0, prec, _lib.TILE_64x64) - want).abs().max().item()
                if d != 0.0:
                    bad.append((M, N, K, it, "k-split", "prec", prec, d))
    if _lib.load().md_gemm_ksplit_launches() - split_before < 12 * max(1, iters // 4):  # every launch above must have taken the split form
        bad.append(("k-split form did not run", _lib.load().md_gemm_ksplit_launches() - split_before))
    # the implicit-GEMM convolution (tap masks, 32-bit pixel index) and the one-division pixel-shuffle epilogue
    import torch.nn.functional as F
    for (B, Cin, H, W, Cout) in [(2, 128, 160, 120, 256), (1, 64, 33, 17, 32)]:
        x = torch.randint(-2, 3, (B, Cin, H, W), generator=g).float()
        w = torch.randint(-1, 2, (Cout, Cin, 3, 3), generator=g).float()
        b = torch.randint(-5, 6, (Cout,), generator=g).float()
        want = F.conv2d(x.double(), w.double(), b.double(), padding=1).float().cuda()
        want16 = want.to(torch.bfloat16).float()
        for it in range(max(1, iters // 4)):
            d = (ops.conv3x3(dev, x.cuda(), w.cuda(), b.cuda(), False, 0) - want).abs().max().item()
            d16 = (ops.conv3x3(dev, x.cuda(), w.cuda(), b.cuda(), False, 0, storage_out=True) - want16).abs().max().item()
            if d != 0.0 or d16 != 0.0:
                bad.append(("conv3x3", B, Cin, H, W, Cout, it, d, d16))
        if verbose:
            print("conv3x3", B, Cin, H, W, Cout, "ok" if not bad else "MISMATCH", flush=True)
    for (B, Cin, H, W, Cout) in [(2, 256, 97, 88, 128), (1, 128, 150, 130, 64)]:
        x = torch.randint(-2, 3, (B, Cin, H, W), generator=g).float()
        w = torch.randint(-1, 2, (Cin, Cout, 2, 2), generator=g).float()
        b = torch.randint(-5, 6, (Cout,), generator=g).float()
        want = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2).float().cuda()
        want16 = want.to(torch.bfloat16).float()
        for it in range(max(1, iters // 4)):
            d = (ops.deconv2x2(dev, x.cuda(), w.cuda(), b.cuda(), 0) - want).abs().max().item()
            d16 = (ops.deconv2x2(dev, x.cuda(), w.cuda(), b.cuda(), 0, storage_out=True) - want16).abs().max().item()
            if d != 0.0 or d16 != 0.0:
                bad.append(("deconv2x2", B, Cin, H, W, Cout, it, d, d16))
        if verbose:
            print("deconv2x2", B, Cin, H, W, Cout, "ok" if not bad else "MISMATCH", flush=True)
    return bad


if __name__ == "__main__":
    bad = run(Device(0), 25, True)
    for r in bad:
        print("MISMATCH", *r)
    print("soak done, mismatches:", len(bad))
    sys.exit(1 if bad else 0)
