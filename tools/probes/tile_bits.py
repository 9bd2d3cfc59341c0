"""Are the GEMM tiles bit-consistent with each other? Same operands through every tile, plain / GELU epilogue, fp32 and
storage-type outputs."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from burn_depth_amd import _lib, ops
from burn_depth_amd.depth_pro import Device

dev = Device(0)
g = torch.Generator().manual_seed(1)
M, N, K = 2320, 1024, 1024
x = torch.randn(M, K, generator=g).bfloat16().float().cuda()
w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16().float().cuda()
b = torch.randn(N, generator=g).cuda()
for prec, pn in ((0, "bf16"), (1, "f32"), (4, "f16x2")):
    for act in (0, 2):
        for so in (False, True):
            ref = None
            for tile, tn in ((0, "256"), (1, "128"), (3, "128x64"), (4, "64")):
                o = ops.linear(dev, x, w, b, act, prec, tile, storage_out=so)
                if ref is None:
                    ref = o
                    continue
                d = (o - ref).abs().max().item()
                print(f"{pn} act{act} storage_out={so} tile {tn} vs 256: equal={torch.equal(o, ref)} max|d|={d:.3e}", flush=True)
