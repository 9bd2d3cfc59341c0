"""CPU-oracle frames of the full-size parity tests, computable in a child process -- TEST INFRASTRUCTURE (imports `oracle/`).

`tools/gpu_diag.py` compares the engine's full-size results (Depth Pro at 1536^2, Depth-Anything-v3 `metric_large` at 1036^2)
with fp32 CPU-oracle frames of 7-19 TFLOP each. Computed one after the other inside the test process they were 60 % of the GPU
suite's wall time (profiles/r04_pytest_gpu.log: 621 s of the driver's 900-s limit) -- and one process on all host cores runs
these shapes at a fraction of what several processes on a share of the cores each reach together. This module holds the frame
functions (no GPU, no libmi_depth.so) and a command line that computes ONE frame and saves it:

    python tools/oracle_frames.py --job full:seeded:f16w:0:fp32 --out /tmp/frame.pt --threads 16

`gpu_diag.prefetch_processes` starts one such child per frame when the first full-size test begins, waits for all of them and
loads the results into its oracle cache. What is compared, and against what, is unchanged."""
from __future__ import annotations

import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def host_cpus() -> int:
    """Cores this process may really use: the affinity mask, cut by a cgroup CPU quota when there is one. (The GPU boxes of this
    pool show 256 CPUs and run under a 16-core quota: torch's default of 128 threads then runs the oracle's GEMMs at 0.67 of
    the rate 16-32 threads reach, profiles/r05_host_probe.txt.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
    except (OSError, ValueError, IndexError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except (OSError, ValueError):
            pass
    return max(1, n)


def full_size_frame(frame="seeded", f16_weights=False, scheme=1, part="fp32"):
    """Depth Pro default config at [1,3,1536,1536]: the input (`x`, and `rgb` for the test.jpg frame) and the oracle's result --
    part "fp32": `R.infer` (dict), part "q": the depth of the oracle that rounds every MFMA operand to bf16 where the engine does."""
    import numpy as np
    import torch
    from burn_depth_amd import weights as Wt
    from burn_depth_amd.config import DepthProConfig
    from oracle import depth_pro_ref as R
    cfg = DepthProConfig()
    W = R.weights_to_torch(Wt.generate_depth_pro_weights(cfg, 0, scheme))
    if f16_weights:
        W = {k: R.f16_round(v) for k, v in W.items()}
    S = cfg.img_size()
    rgb = None
    if frame == "seeded":
        g = torch.Generator().manual_seed(0)  # = torch.manual_seed(0); torch.rand(...), without touching the global generator
        x = (torch.rand(1, 3, S, S, generator=g) - torch.tensor(R.MEAN).view(1, 3, 1, 1)) / torch.tensor(R.STD).view(1, 3, 1, 1)
    elif frame == "zeros":
        x = torch.zeros(1, 3, S, S)
    else:
        rgb = np.load(os.path.join(ROOT, "tests", "golden", "test_jpg_rgb.npy"))
        x = R.rgb_to_input_tensor(rgb.tobytes(), rgb.shape[1], rgb.shape[0])
    t0 = time.time()
    with torch.no_grad():
        if part == "q":
            out = R.infer(x, W, cfg, q=R.bf16_round)["depth"]
        else:
            out = R.infer(x, W, cfg)
    print(f"      full-size oracle {'with bf16 operand rounding' if part == 'q' else 'fp32'} {time.time() - t0:.1f}s ({frame} frame "
          f"{tuple(x.shape)}, {torch.get_num_threads()} threads)", flush=True)
    return dict(x=x, rgb=rgb, out=out)


def da3_frame(variant="metric_large", image_size=1036, B=1, scheme=1):
    """Depth-Anything-v3 at `image_size`^2, seeded input (= torch.manual_seed(1); torch.randn of gpu_diag.run_da3), fp32 oracle."""
    import torch
    from burn_depth_amd import weights as Wt
    from burn_depth_amd.config import DepthAnything3Config
    from oracle import da3_ref as D3
    from oracle import depth_pro_ref as R
    cfg = {"metric_large": DepthAnything3Config.metric_large, "small": DepthAnything3Config.small}[variant]()
    cfg.image_size = image_size
    W = R.weights_to_torch(Wt.generate_da3_weights(cfg, 0, scheme))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 3, cfg.image_size, cfg.image_width or cfg.image_size, generator=g)
    t0 = time.time()
    with torch.no_grad():
        out = D3.infer(x, W, cfg, debug=False)
    print(f"      da3 oracle fp32 {time.time() - t0:.1f}s ({variant} {image_size}^2, {torch.get_num_threads()} threads)", flush=True)
    return dict(x=x, out=out)


def run_job(job: str):
    """job = "full:<frame>:<f16w|f32w>:<scheme>:<fp32|q>" or "da3:<variant>:<size>:<B>:<scheme>"."""
    f = job.split(":")
    if f[0] == "full":
        return full_size_frame(f[1], f[2] == "f16w", int(f[3]), f[4])
    if f[0] == "da3":
        return da3_frame(f[1], int(f[2]), int(f[3]), int(f[4]))
    raise ValueError(f"unknown job `{job}`")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--job", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    import torch
    if a.threads > 0:
        torch.set_num_threads(a.threads)
    res = run_job(a.job)
    torch.save(res, a.out + ".tmp")
    os.replace(a.out + ".tmp", a.out)


if __name__ == "__main__":
    main()
