"""Config 2 (DA3 small, 518^2, graph replay): what the dual head's side branches cost. Times md_da3_infer_ex with (a) every output,
(b) depth + confidence only (main pyramid), (c) depth + aux (no camera), (d) depth + camera (no aux pyramid)."""
import ctypes as C
import os
import sys
import time

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
cases = {"all outputs": (depth, conf, aux, auxc, pose, ext, intr), "depth + confidence (main pyramid only)": (depth, conf, None, None, None, None, None),
         "depth only": (depth, None, None, None, None, None, None), "main + aux pyramid, no camera": (depth, conf, aux, auxc, None, None, None),
         "main + camera, no aux pyramid": (depth, conf, None, None, pose, ext, intr)}
m.enable_graph(True)
for name, outs in cases.items():
    o = L.MdDa3Outputs(*(p(t) for t in outs))
    step = lambda: L.check(L.load().md_da3_infer_ex(m._h, C.c_void_p(x.data_ptr()), B, S, S, L.MD_MEM_DEVICE, C.byref(o), L.MD_MEM_DEVICE, _stream_ptr(0)))
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(300):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 300)
    print(f"{name:45s} {best * 1e3:.3f} ms  {1 / best:.1f} frames/s", flush=True)
