"""Determinism / exactness soak of the 256x256 GEMM: small-integer operands (every product and partial sum exact in
fp32), 25 launches per shape, fp32-store and bf16-store epilogues, bit-exact against the CPU product."""
import sys, math, torch
sys.path.insert(0, ".")
from burn_depth_amd import ops, _lib
from burn_depth_amd.depth_pro import Device
dev = Device(0)
g = torch.Generator().manual_seed(1)
bad = 0
for (M, N, K) in [(21349, 1024, 1024), (5000, 3072, 1024), (3000, 1024, 4096), (2500, 512, 128), (4096, 256, 2304)]:
    x = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    b = torch.randint(-5, 6, (N,), generator=g).float()
    want = (x.double() @ w.double().t() + b.double()).float().cuda()   # exact small integers
    xc, wc, bc = x.cuda(), w.cuda(), b.cuda()
    want16 = want.to(torch.bfloat16).float()  # the engine's bf16 store epilogue rounds to nearest even as well
    for it in range(25):
        got = ops.linear(dev, xc, wc, bc, 0, 0, _lib.TILE_256x256)
        d = (got - want).abs().max().item()
        got16 = ops.linear(dev, xc, wc, bc, 0, 0, _lib.TILE_256x256, storage_out=True)
        d16 = (got16 - want16).abs().max().item()
        if d != 0.0 or d16 != 0.0:
            bad += 1
            print("MISMATCH", M, N, K, it, d, d16)
    print("shape", M, N, K, "ok" if bad == 0 else "bad", flush=True)
print("soak done, mismatches:", bad)
