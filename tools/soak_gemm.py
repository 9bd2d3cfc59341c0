"""Determinism / exactness soak of the 256x256 GEMM: small-integer operands (every product and partial sum exact in
fp32), `iters` launches per shape, fp32-store and bf16-store epilogues, bit-exact against the CPU product."""
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
        if verbose:
            print("shape", M, N, K, "ok" if not bad else "MISMATCH", flush=True)
    return bad


if __name__ == "__main__":
    bad = run(Device(0), 25, True)
    for r in bad:
        print("MISMATCH", *r)
    print("soak done, mismatches:", len(bad))
    sys.exit(1 if bad else 0)
