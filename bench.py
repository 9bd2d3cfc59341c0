"""bench.py -- frames/s of DepthPro::infer on synthetic [B,3,1536,1536] (BASELINE.json metric).

`python bench.py --gpus N --steps K --warmup W`; for N > 1 launched under torch.distributed.run with
one rank per GPU.  A step = one DepthPro::infer over one batch of `--batch` synthetic images that are
already resident in HBM.  Independent images shard over ranks (data parallel, weak scaling: every
rank runs the same per-GPU batch); the only collectives are the one-time weight broadcast from rank 0
(outside the timed region) and, per step, the gather of the depth maps to rank 0 over RCCL/xGMI
(inside the timed region).

The JSON line also carries
  * "roofline": MFMA roofline of the dominant kernel family, from HIP events recorded on the launch
    stream around every launch of that family during the timed steps (md_model_enable_timing);
  * "kernels": the same for every kernel family (ms per step, achieved TFLOP/s or GB/s);
  * "cpu_baseline": the CPU oracle (a port of the reference's NdArray path; the Rust reference
    cannot be built here) timed on this host's cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from burn_depth_amd import weights as Wt  # noqa: E402
from burn_depth_amd.config import DepthProConfig, Precision  # noqa: E402
from burn_depth_amd.depth_pro import DepthPro, Device  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def work_model(cfg: DepthProConfig, B: int):
    """Algorithmic FLOPs (2*MAC) / bytes per kernel family for one batch (SURVEY 8d derivation)."""
    v = cfg.patch_vit()
    D, P, NT, depth, heads = v.embed_dim, v.grid_size() ** 2, v.num_tokens, v.depth, v.num_heads
    S = cfg.img_size()
    nseq = (25 + 9 + 1 + 1 + (1 if cfg.fov_encoder_preset else 0)) * B
    rows = nseq * NT
    F = cfg.decoder_features
    dims = v.encoder_feature_dims
    g = v.grid_size()
    hi, mid = 4 * g, 2 * g
    fl = {}
    fl["patch_embed"] = 2.0 * nseq * P * D * (3 * v.patch_size ** 2)
    fl["qkv_gemm"] = 2.0 * rows * D * 3 * D * depth
    fl["attention"] = 4.0 * nseq * heads * NT * NT * 64 * depth
    fl["proj_gemm"] = 2.0 * rows * D * D * depth
    fl["fc1_gemm"] = 2.0 * rows * D * 4 * D * depth
    fl["fc2_gemm"] = fl["fc1_gemm"]
    px = lambda s: B * s * s  # noqa: E731
    enc_proj = 2.0 * D * (px(hi) * (dims[0] * 2 + dims[1]) + px(mid) * dims[2] + px(g) * dims[3])
    enc_dec = 2.0 * 4 * (px(hi) * dims[0] * F + px(2 * hi) * F * F + px(4 * hi) * F * F + px(hi) * dims[0] * dims[0] +
                         px(2 * hi) * dims[0] * dims[0] + px(hi) * dims[1] * dims[1] + px(mid) * dims[2] * dims[2] +
                         px(g) * dims[3] * dims[3] + px(g) * D * dims[3])
    fl["enc_proj"] = enc_proj
    fl["enc_deconv"] = enc_dec
    fl["enc_fuse"] = 2.0 * px(2 * g) * 2 * dims[3] * dims[3]
    hw = [8 * hi, 4 * hi, 2 * hi, 2 * mid, 2 * g]
    ddims = [F] + list(dims)
    c3 = 0.0
    for l in range(5):
        if l:
            c3 += 2.0 * 9 * ddims[l] * F * px(hw[l])
        c3 += 2.0 * 9 * F * F * px(hw[l]) * (2 if l == 4 else 4)
    fl["dec_conv3x3"] = c3
    # deconv + 1x1 out_conv run as one GEMM on the weight product (executed flops, not the unfused count)
    fl["dec_deconv_out"] = sum(2.0 * 4 * F * F * px(hw[l]) for l in range(1, 5))
    fl["dec_out_conv"] = 2.0 * F * F * px(hw[0])
    fl["head_conv0"] = 2.0 * 9 * F * (F // 2) * px(hw[0])
    fl["head_deconv"] = 2.0 * 4 * (F // 2) ** 2 * px(hw[0])
    fl["head_conv1_fused"] = 2.0 * (9 * (F // 2) * 32 + 32) * px(2 * hw[0])
    by = {}
    by["pyramid_patchify"] = B * 3 * S * S * 4.0 + (35 * B) * P * 3 * v.patch_size ** 2 * 2.0
    by["layernorm"] = (2 * depth + 1) * nseq * NT * D * (4.0 + 2.0)
    by["depth_post"] = B * S * S * 8.0
    by["hook_copy"] = 2 * 25 * B * NT * D * 6.0
    return fl, by


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (B of DepthPro::infer([B,3,S,S])); default 8 for depth_pro (BASELINE config 4's 8 images/GPU), 1 for da3_* (single-image configs 2 / 5)")
    ap.add_argument("--precision", choices=["bf16", "f32", "fp8"], default="bf16",
                    help="fp8 (da3_* only, BASELINE config 5): e4m3 operands for the four ViT linear layers, bf16 elsewhere")
    ap.add_argument("--preset", choices=["full", "small", "tiny"], default="full")
    ap.add_argument("--model", choices=["depth_pro", "da3_large", "da3_small"], default="depth_pro",
                    help="depth_pro = the BASELINE headline; da3_large / da3_small = Depth-Anything-v3 (BASELINE configs 5 / 2)")
    ap.add_argument("--image-size", type=int, default=0, help="da3_* only: square input side (multiple of 14), default 518")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent in-flight batches per GPU, each on its own HIP stream with its own workspace")
    ap.add_argument("--graph", action="store_true",
                    help="replay the launch schedule from a hipGraph and drop the per-kernel HIP events from the timed region "
                         "(no `kernels` / `roofline` in the line: latency mode for the single-image configurations)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--dump-launch-order", default="", help="write the per-launch kernel-family list of one infer (json)")
    args = ap.parse_args()
    if args.batch <= 0:
        args.batch = 8 if args.model == "depth_pro" else 1

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world
    if args.gpus != world and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    dev = Device(local_rank)
    tdev = torch.device("cuda", local_rank)
    if args.model in ("da3_large", "da3_small"):
        return bench_da3(args, dev, tdev, world, rank)
    if args.precision == "fp8":
        print("fp8 operands are built for the Depth-Anything-v3 models only (BASELINE config 5); the Depth Pro headline is bf16",
              file=sys.stderr)
        return 2
    cfg = {"full": DepthProConfig(), "small": DepthProConfig.small_test(), "tiny": DepthProConfig.tiny_test()}[args.preset]
    cfg.precision = Precision.BF16 if args.precision == "bf16" else Precision.F32
    cfg.max_batch = args.batch
    S, B = cfg.img_size(), args.batch
    # weights: random init (DepthPro::new, bench/inference.rs:25). Rank 0 generates, the others receive
    # the fp32 weight arena over RCCL (one-time, outside the timed region).
    from burn_depth_amd.parallel import broadcast_weights, gather_depth
    model = DepthPro.new(dev, cfg, seed=0 if rank == 0 else 1 + rank, init_scheme=Wt.INIT_PARITY)
    t_bcast = 0.0
    if world > 1:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        broadcast_weights(model, src=0)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    img = torch.rand(B, 3, S, S, generator=g)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    x = ((img - mean) / std).to(tdev)  # resident in HBM before the timed region
    depth = torch.empty((B, S, S), dtype=torch.float32, device=tdev)
    focal = torch.empty((B,), dtype=torch.float32, device=tdev)
    fovx = torch.empty((B,), dtype=torch.float32, device=tdev)
    fovy = torch.empty((B,), dtype=torch.float32, device=tdev)
    gathered = None
    do_gather = world > 1 and not args.no_gather
    if do_gather and rank == 0:
        gathered = [torch.empty_like(depth) for _ in range(world)]

    extra = []  # additional in-flight batches: (model, stream, x, depth, focal, fovx, fovy)
    for si in range(1, args.streams):
        m2 = DepthPro.new(dev, cfg, seed=0 if rank == 0 else 1 + rank, init_scheme=Wt.INIT_PARITY)
        extra.append((m2, torch.cuda.Stream(device=tdev), x.clone(), torch.empty_like(depth), torch.empty_like(focal),
                      torch.empty_like(fovx), torch.empty_like(fovy)))
    main_stream = torch.cuda.Stream(device=tdev) if args.streams > 1 else None

    def step():
        if args.streams > 1:
            with torch.cuda.stream(main_stream):
                model.infer_into(x, depth, focal, fovx, fovy)
            for (m2, st2, x2, d2, f2, fx2, fy2) in extra:
                with torch.cuda.stream(st2):
                    m2.infer_into(x2, d2, f2, fx2, fy2)
        else:
            model.infer_into(x, depth, focal, fovx, fovy)
        if do_gather:
            gather_depth(depth, gathered, dst=0)

    if args.dump_launch_order and rank == 0:
        model.enable_timing(True)
        step()
        torch.cuda.synchronize()
        with open(args.dump_launch_order, "w") as f:
            json.dump({"families": model.read_launch_order(), "infers": args.steps}, f)
        model.read_timing()
        model.enable_timing(False)
    if args.graph:
        model.enable_graph(True)
    for _ in range(max(args.warmup, 3 if args.graph else 0)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    model.enable_timing(not args.graph)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timing = model.read_timing()
    model.enable_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ok = bool(torch.isfinite(depth).all().item())

    if rank == 0:
        frames = args.steps * B * world * args.streams
        fps = frames / elapsed
        fl, by = work_model(cfg, B)
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
        kernels = {}
        for name, (ms, calls) in timing.items():
            per_step = ms / args.steps
            e = {"ms_per_step": round(per_step, 4), "launches_per_step": calls // args.steps}
            if name in fl:
                e["tflops"] = round(fl[name] / (per_step * 1e-3) / 1e12, 2)
                e["frac_mfma_peak"] = round(e["tflops"] / peak, 4)
            elif name in by:
                e["gbs"] = round(by[name] / (per_step * 1e-3) / 1e9, 1)
                e["frac_hbm_peak"] = round(e["gbs"] / PEAK_HBM_GBS, 4)
            kernels[name] = e
        mfma = {k: v for k, v in kernels.items() if "tflops" in k or "tflops" in v}
        dom = max(mfma, key=lambda k: mfma[k]["ms_per_step"]) if mfma else None
        roofline = None
        if dom:
            e = kernels[dom]
            symbols = {"fc1_gemm": "md::gemm256_kernel<md::bf16_t, 0, 2, 4> (dense A, 16x16x32 ping-pong, fused bias+GELU store)"}
            roofline = {"kernel": dom, "kernel_symbol": symbols.get(dom) if args.precision == "bf16" else None,
                        "bound": "mfma", "achieved": e["tflops"], "peak": peak, "unit": "TFLOP/s",
                        "frac": e["frac_mfma_peak"], "traffic": pmc_traffic(dom, B, args),
                        "avg_launch_ms": round(e["ms_per_step"] / max(e["launches_per_step"], 1), 4),
                        "flops_per_launch": fl[dom] / max(e["launches_per_step"], 1)}
        gpu_ms = sum(v["ms_per_step"] for v in kernels.values())
        total_flops = sum(fl.values())
        out = {
            "metric": "frames/sec Depth Pro @1536^2 bf16" if args.preset == "full" and args.precision == "bf16"
            else f"frames/sec Depth Pro preset={args.preset} {args.precision}",
            "value": round(fps, 4), "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic (seeded U[0,1) images, ImageNet-normalised; random-init weights)",
            "config": {"workload": f"DepthPro::infer [{B},3,{S},{S}] per GPU, default DepthProConfig" if args.preset == "full"
                       else f"DepthPro::infer [{B},3,{S},{S}] preset {args.preset}",
                       "batch_per_gpu": B, "streams_per_gpu": args.streams, "global_batch": B * world * args.streams,
                       "parallelism": f"dp{world}",
                       "gather_depth_to_rank0": do_gather},
            "finite_output": ok,
            "frame_tflops_algorithmic": round(total_flops / B / 1e12, 3),
            "frame_mfma_frac": round((total_flops / B) * (fps / world) / 1e12 / peak, 4),
            "gpu_kernel_ms_per_step": round(gpu_ms, 3),
            "weight_broadcast_s": round(t_bcast, 4),
            "roofline": roofline,
            "kernels": kernels,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out))
    model.destroy()
    if world > 1:
        dist.destroy_process_group()
    return 0


def bench_da3(args, dev, tdev, world, rank) -> int:
    """frames/s of DepthAnything3::infer (metric_large, mono head) on synthetic [B,3,S,S]; same timing
    contract as the Depth Pro path (barrier + synchronize, max over ranks, weak scaling)."""
    from burn_depth_amd.config import DepthAnything3Config
    from burn_depth_amd.depth_anything3 import DepthAnything3
    small = args.model == "da3_small"
    cfg = DepthAnything3Config.small() if small else DepthAnything3Config.metric_large()
    if args.image_size:
        cfg.image_size = args.image_size
    cfg.precision = {"bf16": Precision.BF16, "f32": Precision.F32, "fp8": Precision.FP8}[args.precision]
    cfg.max_batch = args.batch
    S, B = cfg.image_size, args.batch
    model = DepthAnything3.new(dev, cfg, seed=0, init_scheme=Wt.INIT_PARITY)
    g = torch.Generator(device="cpu").manual_seed(99 + rank)
    x = torch.randn(B, 3, S, S, generator=g).to(tdev)
    depth = torch.empty((B, S, S), dtype=torch.float32, device=tdev)
    # the dual-head variant is timed with EVERY output of DepthAnything3Inference (confidence, aux rays, camera)
    step = (lambda: model.infer(x)) if small else (lambda: model.infer_into(x, depth))
    if args.graph:
        if small:  # fixed output buffers so that the graph key repeats
            from burn_depth_amd import _lib as L
            import ctypes as C
            ps = cfg.patch_size
            ah = 8 * (S // ps)
            bufs = [depth] + [torch.empty(sh, dtype=torch.float32, device=tdev) for sh in
                              ((B, S, S), (B, cfg.aux_output_dim - 1, ah, ah), (B, ah, ah), (B, 1, 9), (B, 1, 3, 4), (B, 1, 3, 3))]
            o = L.MdDa3Outputs(*(t.data_ptr() for t in bufs))
            from burn_depth_amd.depth_pro import _stream_ptr
            step = lambda: L.check(L.load().md_da3_infer_ex(model._h, C.c_void_p(x.data_ptr()), B, S, S, L.MD_MEM_DEVICE, C.byref(o),
                                                            L.MD_MEM_DEVICE, _stream_ptr(dev.ordinal)))
        model.enable_graph(True)
    for _ in range(max(args.warmup, 3 if args.graph else 0)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    model.enable_timing(not args.graph)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timing = model.read_timing()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        v = cfg.vit()
        ph = S // 14
        NT, D, depth_n = ph * ph + 1, v.embed_dim, v.depth
        vit_flops = B * (2.0 * NT * D * 3 * D + 4.0 * NT * NT * D + 2.0 * NT * D * D + 4.0 * NT * D * 4 * D) * depth_n
        fps = args.steps * B * world / elapsed
        kernels = {k: {"ms_per_step": round(ms / args.steps, 4), "launches_per_step": c // args.steps} for k, (ms, c) in timing.items()}
        attn_ms = kernels.get("attention", {}).get("ms_per_step")
        out = {"metric": f"frames/sec Depth-Anything-v3 {cfg.variant} @{S}^2 {args.precision}", "value": round(fps, 3), "unit": "frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
               "data": "synthetic (seeded normal images; random-init weights)",
               "config": {"workload": f"DepthAnything3::infer [{B},3,{S},{S}] per GPU, " + ("small (dual head, all outputs)" if small else "metric_large (mono head)"), "batch_per_gpu": B,
                          "global_batch": B * world, "parallelism": f"dp{world}"},
               "backbone_tflops_algorithmic": round(vit_flops / B / 1e12, 3),
               "attention_tflops": round(4.0 * B * v.num_heads * NT * NT * 64 * depth_n / (attn_ms * 1e-3) / 1e12, 1) if attn_ms else None,
               "kernels": kernels}
        print(json.dumps(out))
    model.destroy()
    if world > 1:
        dist.destroy_process_group()
    return 0


def pmc_traffic(kernel: str, B: int, args):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE are collected in separate runs of this same command; tools/pmc_traffic.py applies the
    gfx950 corrections of MI355X_MICROARCH.md and writes profiles/r01_traffic.json). None if the
    passes were made for another batch/precision."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("batch") == B and t.get("precision") == args.precision and t.get("preset") == args.preset:
            return t["kernels"].get(kernel, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_baseline(cfg: DepthProConfig):
    """Times the CPU oracle (a port: the Rust reference cannot be built here) on a bounded sample:
    the ViT-L patch encoder over 16 of the 37 tiles of one frame, scaled by algorithmic FLOPs.
    ViT work is 73 % of a frame and the oracle runs every part through the same oneDNN/MKL GEMMs."""
    from oracle import depth_pro_ref as R
    import numpy as np
    # use this process's CPU share (a 1-GPU box gets 16 cores of the host), not every core of the host
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))
    torch.set_num_threads(threads)
    v = cfg.patch_vit()
    torch.manual_seed(0)
    tiles = 16
    specs = [s for s in Wt.depth_pro_param_specs(cfg, Wt.INIT_PARITY) if s.name.startswith("encoder.patch_encoder.")]
    W = {s.name: torch.from_numpy(Wt.uniform_stream(s.name, 0, int(np.prod(s.shape)), s.lo, s.hi).reshape(s.shape)) for s in specs}
    x = torch.randn(tiles, 3, v.img_size, v.img_size)
    t0 = time.perf_counter()
    R.vit_forward(x, W, "encoder.patch_encoder", v, v.encoder_feature_layer_ids)
    dt = time.perf_counter() - t0
    fl, _ = work_model(cfg, 1)
    frame_flops = sum(fl.values())
    nseq = 25 + 9 + 1 + 1 + (1 if cfg.fov_encoder_preset else 0)
    vit_flops_per_tile = (fl["patch_embed"] + fl["qkv_gemm"] + fl["attention"] + fl["proj_gemm"] + fl["fc1_gemm"] + fl["fc2_gemm"]) / nseq
    sample_flops = tiles * vit_flops_per_tile
    est_frame_s = dt * frame_flops / sample_flops
    return {"value": round(1.0 / est_frame_s, 5), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle ViT-L/16 forward on {tiles} of {nseq} 384^2 tiles ({sample_flops / 1e12:.3f} of {frame_flops / 1e12:.2f} TFLOP/frame) "
                      f"in {dt:.2f} s, scaled by algorithmic FLOPs", "sample_seconds": round(dt, 3),
            "cpu_tflops": round(sample_flops / dt / 1e12, 3)}


if __name__ == "__main__":
    sys.exit(main())
