"""Per-stage error table at full size (first step): where does each precision mode spend its error?

Runs the default DepthProConfig on one seeded [1,3,1536,1536] frame in MD_PREC_F32 (the parity mode: max-rel 1.6e-5
against the CPU oracle at this size, tests/test_gpu_parity.py::test_full_size_default_config_against_the_oracle) with
the debug taps enabled, keeps its taps on the host, then runs every requested mode on the same frame and compares tap by
tap: encoder_feature_i, decoder_fusion_i, head_*, canonical_inverse_depth (names of example/correctness.rs:98-122) and
the final depth. Writes a JSON table (for profiles/) and prints it.

usage: python tools/stage_errors.py [--modes f16,bf16] [--out gpurun_out/stage_errors.json] [--f16-weights]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from burn_depth_amd import weights as Wt  # noqa: E402
from burn_depth_amd.config import DepthProConfig, Precision  # noqa: E402
from burn_depth_amd.depth_pro import DepthPro, Device  # noqa: E402

TAPS = ([f"encoder_feature_{i}" for i in range(5)] + [f"decoder_fusion_{i}" for i in (4, 3, 2, 1, 0)] +
        ["head_conv0", "head_deconv", "canonical_inverse_depth"])
MODES = {"f32": Precision.F32, "f16": Precision.F16, "bf16": Precision.BF16}
if hasattr(Precision, "F16X2"):
    MODES["f16x2"] = Precision.F16X2


def stats(got: np.ndarray, ref: np.ndarray) -> dict:
    g = torch.from_numpy(got).double().flatten()
    r = torch.from_numpy(ref).double().flatten()
    d = (g - r).abs()
    return {"max_abs": float(d.max()), "max_rel_to_peak": float(d.max() / (r.abs().max() + 1e-30)),
            "rms_rel": float(torch.sqrt((d * d).mean()) / (torch.sqrt((r * r).mean()) + 1e-30)), "ref_rms": float(torch.sqrt((r * r).mean()))}


def run(dev, precision, x, f16_weights, size):
    cfg = DepthProConfig() if size == 1536 else DepthProConfig.small_test()
    cfg.precision = precision
    cfg.max_batch = 1
    model = DepthPro.new(dev, cfg, seed=0, init_scheme=Wt.INIT_PARITY)
    if f16_weights:
        model.round_weights_to_f16()
    model.enable_taps(True)
    out = model.infer(x)
    torch.cuda.synchronize()
    taps = {n: model.read_tap(n) for n in TAPS}
    taps["depth"] = out.depth.cpu().numpy()
    taps["fovx_deg"] = out.fovx_deg.cpu().numpy()
    model.destroy()
    return taps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="f16,bf16")
    ap.add_argument("--out", default="gpurun_out/stage_errors.json")
    ap.add_argument("--f16-weights", action="store_true", help="round the seeded weights to f16 first (an f16 checkpoint, mod.rs:206)")
    ap.add_argument("--size", type=int, default=1536, choices=(1536, 512))
    a = ap.parse_args()
    dev = Device(0)
    torch.manual_seed(0)
    S = a.size
    mean = torch.tensor((0.485, 0.456, 0.406)).view(1, 3, 1, 1)
    std = torch.tensor((0.229, 0.224, 0.225)).view(1, 3, 1, 1)
    x = ((torch.rand(1, 3, S, S) - mean) / std).cuda()
    t0 = time.time()
    ref = run(dev, Precision.F32, x, a.f16_weights, S)
    print(f"fp32 mode: {time.time() - t0:.1f} s; depth in [{ref['depth'].min():.3f}, {ref['depth'].max():.3f}]", flush=True)
    table = {"size": S, "f16_weights": bool(a.f16_weights), "reference": "MD_PREC_F32 engine taps", "modes": {}}
    for name in a.modes.split(","):
        got = run(dev, MODES[name], x, a.f16_weights, S)
        rows = {n: stats(got[n], ref[n]) for n in TAPS}
        d, rd = got["depth"].astype(np.float64), ref["depth"].astype(np.float64)
        rel = np.abs(d - rd) / np.abs(rd)
        rows["depth"] = {"L_inf": float(np.abs(d - rd).max()), "max_rel": float(rel.max()), "p999_rel": float(np.quantile(rel, 0.999)),
                         "mean_rel": float(rel.mean())}
        rows["fovx_deg"] = {"abs": float(np.abs(got["fovx_deg"] - ref["fovx_deg"]).max())}
        table["modes"][name] = rows
        print(f"--- {name} vs fp32 mode (size {S}) ---")
        print(f"{'tap':28s} {'rms-rel':>10s} {'max/peak':>10s} {'max-abs':>10s}")
        for n in TAPS:
            r = rows[n]
            print(f"{n:28s} {r['rms_rel']:10.3e} {r['max_rel_to_peak']:10.3e} {r['max_abs']:10.3e}")
        r = rows["depth"]
        print(f"{'depth':28s} L_inf {r['L_inf']:.3e}  max-rel {r['max_rel']:.3e}  p99.9 {r['p999_rel']:.3e}  mean-rel {r['mean_rel']:.3e}   fovx abs {rows['fovx_deg']['abs']:.2e}", flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(table, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
