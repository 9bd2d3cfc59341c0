"""Config 2 (DA3 small, 518^2, eager): per-family kernel time (HIP events around every launch) with every output and with depth + confidence only."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from burn_depth_amd import _lib as L, weights as Wt  # noqa: E402
from burn_depth_amd.config import DepthAnything3Config, Precision  # noqa: E402
from burn_depth_amd.depth_anything3 import DepthAnything3  # noqa: E402
from burn_depth_amd.depth_pro import Device, _stream_ptr  # noqa: E402

dev = Device(0)
cfg = DepthAnything3Config.small()
cfg.precision = Precision.BF16
S, B = 518, 1
m = DepthAnything3.new(dev, cfg, seed=0, init_scheme=Wt.INIT_PARITY).round_weights_to_f16()
x = torch.randn(B, 3, S, S, device="cuda")
ah = 8 * (S // 14)
f = lambda *sh: torch.empty(sh, dtype=torch.float32, device="cuda")
depth, conf, aux, auxc, pose, ext, intr = f(B, S, S), f(B, S, S), f(B, 6, ah, ah), f(B, ah, ah), f(B, 1, 9), f(B, 1, 3, 4), f(B, 1, 3, 3)
p = lambda t: t.data_ptr() if t is not None else None
for name, outs in {"all outputs": (depth, conf, aux, auxc, pose, ext, intr), "depth + confidence": (depth, conf, None, None, None, None, None)}.items():
    o = L.MdDa3Outputs(*(p(t) for t in outs))
    step = lambda: L.check(L.load().md_da3_infer_ex(m._h, C.c_void_p(x.data_ptr()), B, S, S, L.MD_MEM_DEVICE, C.byref(o), L.MD_MEM_DEVICE, _stream_ptr(0)))
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    m.enable_timing(True)
    N = 20
    for _ in range(N):
        step()
    torch.cuda.synchronize()
    tm = m.read_timing()
    m.enable_timing(False)
    tot = sum(v[0] for v in tm.values()) / N
    print(f"== {name}: {tot * 1e3:.0f} us of kernels per frame, {sum(v[1] for v in tm.values()) // N} launches")
    for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0]):
        print(f"   {k:18s} {v[0] / N * 1e3:7.1f} us  {v[1] // N:3d} launches  {v[0] / v[1] * 1e3:6.1f} us each")
