"""bench.py -- frames/s of DepthPro::infer on synthetic [B,3,1536,1536] (BASELINE.json metric).

`python bench.py --gpus N --steps K --warmup W`. With N > 1 and no WORLD_SIZE in the environment the
script starts its own N ranks (`python -m torch.distributed.run --nproc-per-node N ... bench.py`) as a
CHILD process before anything touches the GPU, relays the child's output and exits with its code; under
an external `torch.distributed.run` it reads RANK / LOCAL_RANK / WORLD_SIZE as usual.

A step = one DepthPro::infer over one batch of `--batch` synthetic images per GPU. Independent images
shard over ranks (data parallel, weak scaling: every rank runs the same per-GPU batch). N = 1: the
batch is resident in HBM when the timed region starts. N > 1 (BASELINE config 4): the whole global
batch is resident in rank 0's HBM; every step scatters the shards from rank 0 (RCCL over xGMI), runs
the engine and gathers the depth maps back to rank 0 -- scatter and gather are INSIDE the timed
region, double-buffered so that the transfers of step k+1 / k-1 overlap the
compute of step k; the one-time weight broadcast from rank 0 is outside it and reported separately.

The JSON line also carries
  * "roofline": MFMA roofline of the dominant kernel family, from HIP events recorded on the launch
    stream around every launch of that family during the timed steps (md_model_enable_timing +
    md_model_set_timing_filter: ONLY that family is timed inside the timed region -- two event records
    around each of the ~226 launches of a step cost 0.7 % of it);
  * "kernels": the same for every kernel family (ms per step, achieved TFLOP/s or GB/s), from a separate
    fully timed pass of up to 5 steps right after the timed region ("kernels_pass" says so);
  * "cpu_baseline": the CPU oracle (a port of the reference's NdArray path; the Rust reference
    cannot be built here) timed on this host's cores on ONE whole frame (BASELINE config 1: zeros
    [1,3,1536,1536], bench/inference.rs:21-48), or on a bounded sample when a frame would not fit
    the time budget.

`--dry-run` replaces the engine by a CPU stand-in and RCCL by gloo: it exercises the launcher, the
scatter / gather pipeline and the JSON contract without a GPU (tests/test_bench_launcher.py).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# torch, torch.distributed and the engine are imported inside the functions that need them: the parent of a
# self-spawned multi-rank run must not touch the GPU.

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 / f16 MFMA, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def work_model(cfg, B: int):
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
    # executed: the last two k2s2 deconvolutions of latent0 / latent1 run as one k4s4 on their weight product (16 taps from the
    # coarser grid = the FLOPs of the finer deconvolution alone; the middle one is gone)
    enc_dec = 2.0 * 4 * (px(hi) * dims[0] * F + px(4 * hi) * F * F +
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
    fl["dec_out_conv"] = 2.0 * F * F * px(hw[0])  # composed into head.conv0 at commit: only launched (for its tap) in debug runs
    fl["head_conv0"] = 2.0 * 9 * F * (F // 2) * px(hw[0])
    # deconv k2s2 -> conv1 3x3 -> conv_out 1x1 (mod.rs:106-111) run as one composed 3x3 convolution with 4 x 32 columns on
    # conv0's output: executed flops (the unfused pair is 1.45x that)
    fl["head_tail_fused"] = 2.0 * (9 * (F // 2) * 128 + 4 * 32) * px(hw[0])
    by = {}
    esz = 4.0 if int(cfg.precision) in (1, 4) else 2.0  # f16x2: two half planes per element
    by["pyramid_patchify"] = B * 3 * S * S * 4.0 + (35 * B) * P * 3 * v.patch_size ** 2 * esz
    by["layernorm"] = (2 * depth + 1) * nseq * NT * D * (4.0 + esz)
    by["depth_post"] = B * S * S * 8.0
    by["hook_copy"] = 2 * 25 * B * NT * D * (4.0 + esz)
    return fl, by


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int, argv) -> int:
    """Parent of a self-launched multi-rank run. Makes NO GPU call (never imports torch): starts the ranks as a child
    process tree through torch.distributed.run, relays stdout / stderr unchanged and returns the child's exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (B of DepthPro::infer([B,3,S,S])); default 8 for depth_pro (BASELINE config 4's 8 images/GPU), 1 for da3_* (single-image configs 2 / 5)")
    ap.add_argument("--precision", choices=["bf16", "f16", "f16x2", "f32", "fp8"], default="bf16",
                    help="MFMA operand type. bf16 = the BASELINE metric; f16 = same rate, 3 more mantissa bits; f16x2 = activations as hi + lo "
                         "half planes on f16-exact weights (the seeded weights are rounded to f16 first, as an f16 checkpoint holds them): the accurate fast mode; "
                         "f32 = parity mode; fp8 (da3_* only, BASELINE config 5): e4m3 operands for the four ViT linear layers")
    ap.add_argument("--preset", choices=["full", "small", "tiny"], default="full")
    ap.add_argument("--model", choices=["depth_pro", "da3_large", "da3_small"], default="depth_pro",
                    help="depth_pro = the BASELINE headline; da3_large / da3_small = Depth-Anything-v3 (BASELINE configs 5 / 2)")
    ap.add_argument("--image-size", type=int, default=0, help="da3_* only: square input side (multiple of 14), default 518")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent in-flight batches per GPU, each on its own HIP stream with its own workspace; the weights are shared (md_model_fork)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the launch schedule from a hipGraph and drop the per-kernel HIP events from the timed region "
                         "(no `kernels` / `roofline` in the line: latency mode for the single-image configurations)")
    ap.add_argument("--attention-form", choices=["asm", "hip"], default="asm",
                    help="bf16 Depth Pro attention (577 tokens): the assembly-owned gfx950 kernel (the product) or the HIP kernel every other shape runs -- an A/B switch (md_debug_attention_asm), recorded in config.attention_form when it is not the default")
    ap.add_argument("--ln-fold", choices=["auto", "off", "neutral", "finish-launch"], default="auto",
                    help="the LayerNorms between the ViT's GEMMs folded into those GEMMs (md_model_set_option(\"ln_fold\"): automatic = on for 16-bit models with 577-token sequences) or as stand-alone launches -- an A/B switch, recorded in config.layernorm")
    ap.add_argument("--direct-store", choices=["on", "off"], default="on",
                    help="lean 2-byte store epilogues of the 256 x 256 GEMM kernel straight from the accumulator layout (the product) or staged through LDS (md_debug_gemm_direct_store): an A/B switch, same bits")
    ap.add_argument("--stagger", default="", help="md_debug_gemm_stagger: 'proj,fc2,fc1,qkv' ticks (10 ns each; -1 = the default): a timing switch, same bits")
    ap.add_argument("--persistent-fc1", choices=["on", "off", "fc1", "fc1qkv", "noqkv", "noconv"], default="on",
                    help="the GEMMs that run as persistent tile loops (md_debug_gemm_persistent; on = fc1 + QKV + proj / fc2 + the lean 3x3 convolutions): an A/B switch, same bits")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="no per-launch HIP events in the timed region (the `kernels` / `roofline` objects are then empty): measures what the events themselves cost")
    ap.add_argument("--cpu-baseline-budget", type=float, default=150.0, help="seconds the whole-frame CPU baseline may take (predicted from a 2-tile probe); beyond it the sampled estimate is reported")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: no depth gather to rank 0")
    ap.add_argument("--no-scatter", action="store_true", help="N > 1: every rank synthesises its own batch instead of receiving it from rank 0")
    ap.add_argument("--accuracy", action="store_true", help="also report depth max-rel / mean-rel / L_inf of this precision against the fp32 CPU oracle on one seeded frame of the 512^2 ViT-L preset (`accuracy` object in the line)")
    ap.add_argument("--side-kernels", action="store_true", help="also time the stand-alone HBM-bound kernels (resize_bilinear / resize_nhwc at the shapes of bench/interpolate.rs) into `kernels`")
    ap.add_argument("--dry-run", action="store_true", help="CPU stand-in for the engine + gloo instead of RCCL: tests the launcher / scatter / gather / JSON contract without a GPU")
    ap.add_argument("--dump-launch-order", default="", help="write the per-launch kernel-family list of one infer (json)")
    ap.add_argument("--native-comm", action="store_true",
                    help="N > 1: weight broadcast, image scatter and depth gather through the library's own RCCL entry points (md_comm_*, grouped "
                         "ncclSend / ncclRecv on a side HIP stream) instead of torch.distributed; torch.distributed only carries the rendezvous id")
    ap.add_argument("--tile-parallel", action="store_true",
                    help="SURVEY 8(e) second mode: ONE image per step, its 37 ViT sequences split over the N ranks (md_comm_depth_pro_infer_tiles); "
                         "the line reports single-image throughput = 1 / latency, scaling 'strong'")
    ap.add_argument("--no-extras", action="store_true",
                    help="default N = 1 Depth Pro bf16 run only: skip the objects measured after the timed region (`configs`: the other BASELINE "
                         "configurations; `accurate_mode` / `fp32_mode_fps`; `accuracy` at 1536^2 against the CPU-baseline frame; `host_io`)")
    ap.add_argument("--host-io", action="store_true", help="report `host_io` (host NCHW in, host depth out through md_depth_pro_infer, as INTEGRATION.md section 2's shim calls it) for any precision / preset")
    return ap.parse_args(argv)


_JSON_OUT = None


def _claim_stdout() -> None:
    """The contract is ONE JSON line on stdout. Libraries under this process write to file descriptor 1 on their own (RCCL prints
    a five-line version banner at communicator creation, ROCm tools their notices), so a worker keeps the original descriptor for
    the JSON line and points fd 1 at stderr for everything else."""
    global _JSON_OUT
    if _JSON_OUT is not None:
        return
    sys.stdout.flush()
    _JSON_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)


def emit(obj) -> None:
    out = _JSON_OUT if _JSON_OUT is not None else sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.batch <= 0:
        args.batch = 8 if args.model == "depth_pro" else 1
    if args.gpus < 1:
        print("--gpus must be >= 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus, argv)  # before any torch / HIP import in this process
    _claim_stdout()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; running {world} ranks", file=sys.stderr)
    if args.dry_run:
        return bench_dry(args, world, rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from burn_depth_amd import weights as Wt
    from burn_depth_amd.config import DepthProConfig, Precision
    from burn_depth_amd.depth_pro import DepthPro, Device
    from burn_depth_amd.parallel import broadcast_weights

    dev = Device(local_rank)
    tdev = torch.device("cuda", local_rank)
    if args.persistent_fc1 != "on":
        from burn_depth_amd import _lib as _lps
        _lps.load().md_debug_gemm_persistent({"off": 0, "fc1": 1, "fc1qkv": 3, "noqkv": 5, "noconv": 7}[args.persistent_fc1])
    if args.stagger:
        from burn_depth_amd import _lib as _lst
        for which, ticks in enumerate(int(v) for v in args.stagger.split(",")):
            if ticks >= 0:
                _lst.load().md_debug_gemm_stagger(which, ticks)
    if args.direct_store == "off":
        from burn_depth_amd import _lib as _lds
        _lds.load().md_debug_gemm_direct_store(0)
    if args.attention_form == "hip":  # before the model exists: its graphs capture whichever form the first infer launches
        from burn_depth_amd import _lib as _l
        _l.load().md_debug_attention_asm(0)
    if args.model in ("da3_large", "da3_small"):
        return bench_da3(args, dev, tdev, world, rank)
    if args.tile_parallel:
        return bench_tile_parallel(args, dev, tdev, world, rank)
    if args.precision == "fp8":
        print("fp8 operands are built for the Depth-Anything-v3 models only (BASELINE config 5); the Depth Pro headline is bf16",
              file=sys.stderr)
        return 2
    cfg = {"full": DepthProConfig(), "small": DepthProConfig.small_test(), "tiny": DepthProConfig.tiny_test()}[args.preset]
    cfg.precision = {"bf16": Precision.BF16, "f16": Precision.F16, "f32": Precision.F32, "f16x2": Precision.F16X2}[args.precision]
    cfg.max_batch = args.batch
    S, B = cfg.img_size(), args.batch
    # weights: random init (DepthPro::new, bench/inference.rs:25). Rank 0 generates, the others receive
    # the fp32 weight arena over RCCL (one-time, outside the timed region).
    model = DepthPro.new(dev, cfg, seed=0 if rank == 0 else 1 + rank, init_scheme=Wt.INIT_PARITY)
    if args.ln_fold != "auto":  # "neutral": a diagnostic -- the unfolded schedule through the fold-form consumer kernels on neutral statistics
        model.set_option("ln_fold", {"off": 0, "neutral": 3, "finish-launch": 4}[args.ln_fold])
    t_bcast = 0.0
    ncomm = None
    if args.native_comm:
        from burn_depth_amd.parallel import NativeComm
        ncomm = NativeComm.from_torch_distributed(dev) if world > 1 else NativeComm(dev, NativeComm.unique_id(), 1, 0)
    if world > 1 or ncomm is not None:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if ncomm is not None:
            ncomm.broadcast_weights(model, root=0)
        else:
            broadcast_weights(model, src=0)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
    # the reference's checkpoints are f16 records (`HalfPrecisionSettings`, depth_pro/mod.rs:206): every mode is measured on
    # weights an f16 checkpoint can hold (in f16x2 they are exact MFMA operands: two terms per product)
    model.round_weights_to_f16()

    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)

    def synth(n, seed):
        g = torch.Generator(device="cpu").manual_seed(seed)
        return (torch.rand(n, 3, S, S, generator=g) - mean) / std

    do_scatter = (world > 1 or ncomm is not None) and not args.no_scatter
    do_gather = (world > 1 or ncomm is not None) and not args.no_gather
    nbuf = 2 if (do_scatter or do_gather) else 1  # double buffering: transfers of neighbouring steps overlap this step's compute
    if do_scatter:
        # the whole global batch lives in rank 0's HBM before the timed region; the other ranks own receive buffers only
        global_in = torch.cat([synth(B, 1234 + r) for r in range(world)], 0).to(tdev) if rank == 0 else None
        xs = [torch.empty(B, 3, S, S, dtype=torch.float32, device=tdev) for _ in range(nbuf)]
    else:
        global_in = None
        xs = [synth(B, 1234 + rank).to(tdev)] * nbuf
    depths = [torch.empty((B, S, S), dtype=torch.float32, device=tdev) for _ in range(nbuf)]
    focal = torch.empty((B,), dtype=torch.float32, device=tdev)
    fovx = torch.empty((B,), dtype=torch.float32, device=tdev)
    fovy = torch.empty((B,), dtype=torch.float32, device=tdev)
    gathered = [[torch.empty_like(depths[0]) for _ in range(world)] for _ in range(nbuf)] if (do_gather and rank == 0) else None
    scatter_chunks = list(global_in.split(B, 0)) if (do_scatter and rank == 0) else None
    pipe = None
    if ncomm is not None:
        # native path: RCCL point-to-point groups on a side stream, ordered against the compute stream by events
        # (burn_depth_amd/parallel.py::NativePipeline; the same bookkeeping runs under `--dry-run --native-comm` on the CPU)
        from burn_depth_amd.parallel import NativePipeline
        gathered_flat = [torch.empty((world * B, S, S), dtype=torch.float32, device=tdev) for _ in range(nbuf)] if (do_gather and rank == 0) else None
        if gathered_flat is not None:
            gathered = [list(g.split(B, 0)) for g in gathered_flat]
        pipe = NativePipeline(ncomm, lambda slot, cur: model.infer_into(xs[slot], depths[slot], focal, fovx, fovy), nbuf, rank, 0, do_scatter, do_gather,
                              global_in, xs, depths, gathered_flat, make_stream=lambda: torch.cuda.Stream(device=tdev), make_event=torch.cuda.Event,
                              current_stream=torch.cuda.current_stream, stream_handle=lambda st: st.cuda_stream)

    extra = []  # additional in-flight batches: (forked model sharing the weights, stream, x, depth, focal, fovx, fovy)
    for si in range(1, args.streams):
        m2 = model.fork()
        extra.append((m2, torch.cuda.Stream(device=tdev), xs[0].clone(), torch.empty_like(depths[0]), torch.empty_like(focal),
                      torch.empty_like(fovx), torch.empty_like(fovy)))
    main_stream = torch.cuda.Stream(device=tdev) if args.streams > 1 else None

    pending = {"scatter": None, "gather": [None] * nbuf, "k": 0}

    def issue_scatter(slot):
        pending["scatter"] = dist.scatter(xs[slot], scatter_chunks, src=0, async_op=True)

    def step():
        if pipe is not None:
            pipe.step()
            pending["k"] = pipe.k
            return
        k = pending["k"]
        slot = k % nbuf
        if do_scatter:
            if pending["scatter"] is None:
                issue_scatter(slot)          # first step of a run: nothing was prefetched
            pending["scatter"].wait()        # the compute stream waits for this step's shard
            issue_scatter((k + 1) % nbuf)    # the next step's shard travels while this step computes
        if do_gather and pending["gather"][slot] is not None:
            pending["gather"][slot].wait()   # depth buffer `slot` was handed to a gather two steps ago
            pending["gather"][slot] = None
        if args.streams > 1:
            with torch.cuda.stream(main_stream):
                model.infer_into(xs[slot], depths[slot], focal, fovx, fovy)
            for (m2, st2, x2, d2, f2, fx2, fy2) in extra:
                with torch.cuda.stream(st2):
                    m2.infer_into(x2, d2, f2, fx2, fy2)
        else:
            model.infer_into(xs[slot], depths[slot], focal, fovx, fovy)
        if do_gather:
            pending["gather"][slot] = dist.gather(depths[slot], gathered[slot] if rank == 0 else None, dst=0, async_op=True)
        pending["k"] = k + 1

    def drain():
        if pipe is not None:
            pipe.drain(torch.cuda.synchronize)
            return
        if pending["scatter"] is not None:
            pending["scatter"].wait()
            pending["scatter"] = None
        for i, w in enumerate(pending["gather"]):
            if w is not None:
                w.wait()
                pending["gather"][i] = None
        torch.cuda.synchronize()

    if args.dump_launch_order and rank == 0 and world == 1:
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
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # Per-launch HIP events on EVERY kernel cost 0.7 % of a step (226 launches x 2 records; measured with --no-kernel-timing),
    # so the timed region times only the family that is reported against its roofline (found by one untimed, fully timed step)
    # and the per-family table comes from a separate fully timed pass after the timed region.
    per_kernel = not args.graph and not args.no_kernel_timing
    dom_family = None
    if per_kernel:
        model.enable_timing(True)
        step()
        drain()
        torch.cuda.synchronize()
        probe = model.read_timing()
        model.enable_timing(False)
        fl_probe, _ = work_model(cfg, B)
        cand = {k: v[0] for k, v in probe.items() if k in fl_probe}
        dom_family = max(cand, key=cand.get) if cand else None
        model.set_timing_filter(dom_family)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    model.enable_timing(per_kernel)
    from burn_depth_amd import _lib as _l_asm
    _asm0 = int(_l_asm.load().md_debug_attention_asm_launches())  # what the timed region really launched, not what the flag asked for
    _l_asm.load().md_debug_attention_redo_units(dev.handle, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()  # every scatter / gather issued for the timed steps has completed (the one prefetched shard beyond them included)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timing_dom = model.read_timing()  # the dominant family's launches, measured inside the timed region
    asm_launches = int(_l_asm.load().md_debug_attention_asm_launches()) - _asm0  # 0 under graph replay (launches are recorded at capture)
    redo_units = int(_l_asm.load().md_debug_attention_redo_units(dev.handle, 0))
    model.enable_timing(False)
    model.set_timing_filter(None)
    table_steps = 0
    timing = {}
    if per_kernel:
        table_steps = min(args.steps, 5)
        model.enable_timing(True)
        for _ in range(table_steps):
            step()
        drain()
        torch.cuda.synchronize()
        timing = model.read_timing()
        model.enable_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ok = all(bool(torch.isfinite(d).all().item()) for d in depths)
    if do_gather and rank == 0:
        ok = ok and all(bool(torch.isfinite(t).all().item()) for t in gathered[(pending["k"] - 1) % nbuf])

    if rank == 0:
        frames = args.steps * B * world * args.streams
        fps = frames / elapsed
        fl, by = work_model(cfg, B)
        peak = PEAK_F32_TFLOPS if args.precision == "f32" else PEAK_BF16_TFLOPS
        kernels = {}
        for name, (ms, calls) in timing.items():  # the separate, fully timed pass of `table_steps` steps
            per_step = ms / max(table_steps, 1)
            e = {"ms_per_step": round(per_step, 4), "launches_per_step": calls // max(table_steps, 1)}
            if name in fl:
                e["tflops"] = round(fl[name] / (per_step * 1e-3) / 1e12, 2)
                e["frac_mfma_peak"] = round(e["tflops"] / peak, 4)
            elif name in by:
                e["gbs"] = round(by[name] / (per_step * 1e-3) / 1e9, 1)
                e["frac_hbm_peak"] = round(e["gbs"] / PEAK_HBM_GBS, 4)
            kernels[name] = e
        roofline = None
        dom = dom_family if (dom_family in timing_dom and dom_family in fl) else None
        if dom:  # measured live by HIP events on the launch stream INSIDE the timed region (only this family was timed there)
            ms, calls = timing_dom[dom]
            per_step = ms / args.steps
            launches = calls // args.steps
            tfl = fl[dom] / (per_step * 1e-3) / 1e12
            # the rocprofv3 row of the same command: EK 11 = the GELU store kind with the LayerNorm fold, direct stores (EK 9 without the fold)
            pf = args.persistent_fc1 != "off"
            symbols = {"fc1_gemm": ((("md::gemm256p_kernel<md::bf16_t, true, false, false> (persistent tile loop, " if pf else "md::gemm256_kernel<md::bf16_t, 0, 11, false> (")
                                     + "dense A, 16x16x32 two-group schedule, LayerNorm fold + bias + GELU, direct store)")
                                    if model.query("ln_fold_active") else
                                    (("md::gemm256p_kernel<md::bf16_t, false, false, false> (persistent tile loop, " if pf else "md::gemm256_kernel<md::bf16_t, 0, 9, false> (")
                                     + "dense A, 16x16x32 two-group schedule, fused bias + GELU, direct store)"))}
            roofline = {"kernel": dom, "kernel_symbol": symbols.get(dom) if args.precision == "bf16" else None,
                        "bound": "mfma", "achieved": round(tfl, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(tfl / peak, 4), "traffic": pmc_traffic(dom, B, args)[0], "traffic_source": pmc_traffic(dom, B, args)[1],
                        "avg_launch_ms": round(per_step / max(launches, 1), 4),
                        "flops_per_launch": fl[dom] / max(launches, 1),
                        "ms_per_step": round(per_step, 4), "launches_per_step": launches}
        # the form the timed region launched: the library's own launch counter (a captured graph replays what its capture launched)
        asm_ran = asm_launches > 0 or (args.graph and args.attention_form == "asm" and args.precision == "bf16" and redo_units >= 0 and cfg.patch_vit().num_tokens == 577)
        attention_form = ("attn577_gfx950.s (assembly-owned, one persistent workgroup per CU)" if asm_ran else "hip kernel")
        # north_star's named kernel target (>= 40 % of the bf16 MFMA peak in attention) as a first-class object of the line
        roofline_attention = None
        if "attention" in kernels and "tflops" in kernels["attention"]:
            ka = kernels["attention"]
            roofline_attention = {"kernel": "attention", "kernel_symbol": "md_attn577_bf16 (kernels/attn577_gfx950.s) + attention_redo_scan_kernel + attention_redo_kernel" if asm_ran
                                  else "md::attention_kernel<T, false, true>",
                                  "bound": "mfma", "achieved": ka["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": ka["frac_mfma_peak"],
                                  "avg_launch_ms": round(ka["ms_per_step"] / max(ka["launches_per_step"], 1), 4), "launches_per_step": ka["launches_per_step"],
                                  "flops": "useful: 4 * sequences * heads * N^2 * 64 per launch (the 577 -> 640 padding does not count)",
                                  "asm_launches_in_timed_region": asm_launches,
                                  "units_recomputed_by_the_safe_body": redo_units if redo_units >= 0 else None,
                                  "target": 0.40}
        gpu_ms = sum(v["ms_per_step"] for v in kernels.values())
        if args.side_kernels:
            kernels.update(side_kernels(dev, tdev))
        # executed FLOPs: a family the schedule no longer launches (a layer composed into its neighbour at commit) does not count
        total_flops = sum(v for k, v in fl.items() if k in kernels)
        out = {
            "metric": "frames/sec Depth Pro @1536^2 bf16" if args.preset == "full" and args.precision == "bf16"
            else f"frames/sec Depth Pro preset={args.preset} {args.precision}",
            "value": round(fps, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic (seeded U[0,1) images, ImageNet-normalised; random-init weights rounded to f16 like the reference's f16 checkpoint records)",
            "config": {"workload": f"DepthPro::infer [{B},3,{S},{S}] per GPU, default DepthProConfig" if args.preset == "full"
                       else f"DepthPro::infer [{B},3,{S},{S}] preset {args.preset}",
                       "batch_per_gpu": B, "batch_note": "8 images per GPU = BASELINE config 4's shard; config 3 as SURVEY 8(d) words it (B = 1) is configs[0]" if B == 8 else None,
                       "streams_per_gpu": args.streams, "global_batch": B * world * args.streams,
                       "parallelism": f"dp{world}",
                       "attention_form": attention_form,
                       "layernorm": ("folded into proj / fc2 (round(gamma x) + row statistics) and qkv / fc1 (rstd (acc - mu c) + d); block 0's norm1 and the final norm are launches"
                                     if model.query("ln_fold_active") else "stand-alone launches"),
                       "scatter_inputs_from_rank0": do_scatter, "gather_depth_to_rank0": do_gather,
                       "comm": "native md_comm_* (RCCL point-to-point groups on a side stream)" if ncomm is not None else ("torch.distributed (RCCL)" if world > 1 else None),
                       # what the communicator itself reports (ncclCommCount / the process group's size), beside WORLD_SIZE
                       "ranks_seen": (ncomm.ranks_seen() if ncomm is not None else (dist.get_world_size() if world > 1 else 1))},
            "finite_output": ok,
            # FLOPs the schedule EXECUTES per frame (layers composed at commit count once); null when no per-family pass ran (--graph)
            "frame_tflops_executed": round(total_flops / B / 1e12, 3) if kernels else None,
            "frame_mfma_frac": round((total_flops / B) * (fps / world) / 1e12 / peak, 4) if kernels else None,
            "gpu_kernel_ms_per_step": round(gpu_ms, 3),
            "weight_broadcast_s": round(t_bcast, 4),
            "roofline": roofline,
            "roofline_attention": roofline_attention,
            "kernels_pass": (f"separate fully timed pass of {table_steps} steps after the timed region: two HIP event records around each of "
                             "the ~226 launches cost 0.7 % of a step, so inside the timed region only the roofline family is timed") if table_steps else None,
            "kernels": kernels,
        }
        if world == 1:
            out["box"] = box_probe(tdev)
        if args.accuracy:
            out["accuracy"] = accuracy_report(dev, cfg.precision)
        ref_frame = None
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], ref_frame = cpu_baseline(cfg, args.cpu_baseline_budget)
        headline = world == 1 and args.preset == "full" and args.precision == "bf16" and args.streams == 1 and not args.graph
        if args.host_io or (headline and not args.no_extras):
            out["host_io"] = measure_host_io(model, B, S, fps)
        if headline and not args.no_extras:
            for e in extra:
                e[0].destroy()
            extra = []
            out.update(extra_measurements(dev, tdev, model, cfg, B, ref_frame))
            cands = out.pop("_tolerance_candidates", None)
            if cands:
                for c in cands:
                    if c["fps"] is None:
                        c["fps"] = out["value"]
                okc = [c for c in cands if c["depth_linf"] < 1e-3]
                best = max(okc, key=lambda c: c["fps"]) if okc else None
                out["value_at_tolerance"] = (dict(best, tolerance="depth L_inf < 1e-3 against the fp32 CPU oracle frame of `accuracy` (north_star)",
                                                  batch_per_gpu=(1 if best["precision"] == "f32" else B)) if best else None)
        emit(out)
    for e in extra:
        e[0].destroy()
    model.destroy()
    if ncomm is not None:
        ncomm.destroy()
    if world > 1:
        dist.destroy_process_group()
    return 0


def _timed_steps(step, warmup, steps):
    import torch
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def _dominant(model, step, fl, peak, steps=3):
    """Per-family pass (HIP events on the launch stream) of `steps` eager steps: the dominant MFMA family against its roofline."""
    import torch
    model.enable_timing(True)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    tm = model.read_timing()
    model.enable_timing(False)
    cand = {k: v[0] for k, v in tm.items() if k in fl}
    if not cand:
        return None
    dom = max(cand, key=cand.get)
    ms, calls = tm[dom]
    per_step = ms / steps
    tfl = fl[dom] / (per_step * 1e-3) / 1e12
    return {"kernel": dom, "bound": "mfma", "achieved": round(tfl, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tfl / peak, 4), "traffic": None,
            "avg_launch_ms": round(per_step / max(calls // steps, 1), 4), "ms_per_step": round(per_step, 4), "launches_per_step": calls // steps}


def box_probe(tdev):
    """Which class of box this run landed on (plumbing only: a torch device-to-device copy, no engine kernel). The MI355X boxes of the
    pool differ in what their memory side sustains -- the HBM-bound LayerNorm family takes 9.8 ms per step on some and 13.7 on others
    with byte-identical kernels (rounds 3-5) -- so a line carries the copy rate of ITS box: compare HBM-bound
    families across runs against it, not against each other."""
    import torch
    n = 1 << 28  # 1 GiB of floats read + 1 GiB written per copy
    a = torch.empty(n, dtype=torch.float32, device=tdev).normal_()
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(5):
        b.copy_(a)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 5
    del a, b
    return {"d2d_copy_tbs": round(2 * n * 4 / (ms * 1e-3) / 1e12, 3), "what": "torch copy_ of 1 GiB (read + written bytes over time): the box's achievable HBM rate",
            "device": torch.cuda.get_device_name(0)}


def measure_host_io(model, B, S, resident_fps, steps=4):
    """The call a reference-side host makes (`DepthModel::infer_depth` over the C ABI): pageable
    host NCHW in, pageable host depth out, one synchronous md_depth_pro_infer per step (PCIe inside the timed region)."""
    import ctypes as C
    import numpy as np
    from burn_depth_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    x = rng.standard_normal((B, 3, S, S), dtype=np.float32)
    depth = np.empty((B, S, S), np.float32)
    sc = [np.empty(B, np.float32) for _ in range(3)]
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731

    def step():
        _lib.check(lib.md_depth_pro_infer(model._h, p(x), B, S, S, _lib.MD_MEM_HOST, p(depth), p(sc[0]), p(sc[1]), p(sc[2]), _lib.MD_MEM_HOST, None))
    a0 = None
    step()
    step()
    a0 = model.query("allocs")
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    out = {"value": round(B / dt, 3), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 3), "batch": B,
           "path": "pageable host NCHW -> pinned bounce -> device; depth device -> pinned bounce -> pageable host; synchronous call",
           "bytes_per_step": int(B * (3 + 1) * S * S * 4), "vs_resident_inputs": round(B / dt / resident_fps, 4),
           "allocations_during_timed_steps": int(model.query("allocs") - a0), "finite_output": bool(np.isfinite(depth).all())}
    # The same synchronous call from TWO host threads, each on its own inference context over the SAME weights (md_model_fork: own
    # workspace, stream, staging and pinned buffers) -- what a reference-side host does with `model.clone()` per worker thread
    # (crates/bevy_burn_depth/src/lib.rs:18,29). One thread's PCIe transfers and host copies run under the other's kernels.
    try:
        import threading
        fork = model.fork()
        ctxs = [(model, x, depth, sc), (fork, x.copy(), np.empty_like(depth), [np.empty(B, np.float32) for _ in range(3)])]

        def worker(m, xx, dd, ss, n):
            for _ in range(n):
                _lib.check(lib.md_depth_pro_infer(m._h, p(xx), B, S, S, _lib.MD_MEM_HOST, p(dd), p(ss[0]), p(ss[1]), p(ss[2]), _lib.MD_MEM_HOST, None))
        worker(*ctxs[1], 1)  # the fork's first call allocates its staging
        ths = [threading.Thread(target=worker, args=(*c, steps)) for c in ctxs]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt2 = (time.perf_counter() - t0) / (2 * steps)
        same = bool(np.array_equal(ctxs[0][2], ctxs[1][2]))
        fork.destroy()
        out["two_host_threads"] = {"value": round(B / dt2, 3), "unit": "frames/s", "ms_per_step": round(dt2 * 1e3, 3),
                                   "vs_resident_inputs": round(B / dt2 / resident_fps, 4), "outputs_equal": same,
                                   "what": "two threads, each a synchronous md_depth_pro_infer on its own md_model_fork context (shared weights): transfers of one overlap the kernels of the other"}
    except Exception as ex:  # noqa: BLE001 -- an extra must never lose the headline line
        out["two_host_threads"] = {"error": f"{type(ex).__name__}: {ex}"}
    return out


# the reference's own acceptance bar for Depth-Anything-v3 (example/correctness.rs:1109-1111)
DA3_REF_BAR = {"max_abs": 5e-3, "mean_abs": 1e-3, "max_rel": 1e-2}


def _da3_cfg(variant, size):
    from burn_depth_amd.config import DepthAnything3Config
    cfg = DepthAnything3Config.small() if variant == "small" else DepthAnything3Config.metric_large()
    cfg.image_size = size
    cfg.max_batch = 1
    return cfg


def da3_reference_frame(variant, size):
    """One fp32 CPU-oracle frame of a BASELINE Depth-Anything-v3 configuration (oracle/da3_ref.py: test infrastructure, used
    here only to CHECK the measured modes): the seeded input of `measure_da3`, the seed-0 weights rounded to f16 like the
    reference's DA3 records (NamedMpkFileRecorder<HalfPrecisionSettings>, example/correctness.rs:977)."""
    import torch
    from burn_depth_amd import weights as Wt
    from oracle import da3_ref as D3, depth_pro_ref as R
    cfg = _da3_cfg(variant, size)
    W = {k: R.f16_round(t) for k, t in R.weights_to_torch(Wt.generate_da3_weights(cfg, 0, Wt.INIT_PARITY)).items()}
    x = torch.randn(1, 3, size, size, generator=torch.Generator().manual_seed(99))
    t0 = time.perf_counter()
    with torch.no_grad():
        depth = D3.infer(x, W, cfg)["depth"]
    return {"x": x, "depth": depth, "seconds": round(time.perf_counter() - t0, 2), "range": [float(depth.min()), float(depth.max())]}


def measure_da3(dev, tdev, variant, size, precision, steps=10, ref=None):
    """One BASELINE Depth-Anything-v3 configuration (B = 1, graph replay for the rate, one eager per-family pass for the roofline).
    Weights: seed 0, rounded to f16 (what the reference's f16 records hold). `ref` (da3_reference_frame): the depth of the timed
    input is compared with the fp32 CPU oracle -- max-abs / mean-abs / max-rel, the reference's own statistics."""
    import ctypes as C
    import torch
    from burn_depth_amd import _lib as L, weights as Wt
    from burn_depth_amd.config import Precision
    from burn_depth_amd.depth_anything3 import DepthAnything3
    from burn_depth_amd.depth_pro import _stream_ptr
    small = variant == "small"
    cfg = _da3_cfg(variant, size)
    cfg.precision = {"bf16": Precision.BF16, "fp8": Precision.FP8, "f16": Precision.F16, "f16x2": Precision.F16X2, "f32": Precision.F32}[precision]
    model = DepthAnything3.new(dev, cfg, seed=0, init_scheme=Wt.INIT_PARITY).round_weights_to_f16()
    S, B = size, 1
    x = (ref["x"] if ref is not None else torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(99))).to(tdev)
    depth = torch.empty((B, S, S), dtype=torch.float32, device=tdev)
    if small:  # every output of DepthAnything3Inference, fixed buffers so that the graph key repeats
        ah = 8 * (S // cfg.patch_size)
        bufs = [depth] + [torch.empty(sh, dtype=torch.float32, device=tdev) for sh in
                          ((B, S, S), (B, cfg.aux_output_dim - 1, ah, ah), (B, ah, ah), (B, 1, 9), (B, 1, 3, 4), (B, 1, 3, 3))]
        o = L.MdDa3Outputs(*(t.data_ptr() for t in bufs))
        step = lambda: L.check(L.load().md_da3_infer_ex(model._h, C.c_void_p(x.data_ptr()), B, S, S, L.MD_MEM_DEVICE, C.byref(o),  # noqa: E731
                                                        L.MD_MEM_DEVICE, _stream_ptr(dev.ordinal)))
    else:
        step = lambda: model.infer_into(x, depth)  # noqa: E731
    v = cfg.vit()
    NT, D, dn = (S // 14) ** 2 + 1, v.embed_dim, v.depth
    fl = {"qkv_gemm": 2.0 * NT * D * 3 * D * dn, "proj_gemm": 2.0 * NT * D * D * dn, "fc1_gemm": 2.0 * NT * D * 4 * D * dn,
          "fc2_gemm": 2.0 * NT * D * 4 * D * dn, "attention": 4.0 * v.num_heads * NT * NT * 64 * dn}
    peak = PEAK_F32_TFLOPS if precision == "f32" else PEAK_BF16_TFLOPS
    roof = _dominant(model, step, fl, peak) if precision != "f32" else None
    if roof and precision == "fp8" and roof["kernel"] != "attention":
        roof["peak"], roof["frac"] = 2 * peak, round(roof["achieved"] / (2 * peak), 4)  # e4m3 operands on the block-scaled MFMA: 5 PFLOP/s dense
    model.enable_graph(True)
    dt = _timed_steps(step, 4, steps)
    ok = bool(torch.isfinite(depth).all())
    out = {"value": round(B / dt, 2), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 3), "dtype": precision, "graph_replay": True,
           "workload": f"DepthAnything3::infer [1,3,{S},{S}] {cfg.variant}" + (" (dual head, every output)" if small else " (mono head)"),
           "roofline": roof, "finite_output": ok}
    if precision == "f16x2":
        out["weight_terms"] = model.query("weight_terms")
    if ref is not None:
        d, rd = depth.cpu(), ref["depth"]
        err = (d - rd).abs()
        rel = err / rd.abs()
        e = {"depth_max_abs": float(err.max()), "depth_mean_abs": float(err.mean()), "depth_max_rel": float(rel.max()), "depth_mean_rel": float(rel.mean())}
        e["within_reference_bar"] = bool(e["depth_max_abs"] <= DA3_REF_BAR["max_abs"] and e["depth_mean_abs"] <= DA3_REF_BAR["mean_abs"] and e["depth_max_rel"] <= DA3_REF_BAR["max_rel"])
        out["accuracy"] = e
    model.destroy()
    return out


def extra_measurements(dev, tdev, model, cfg, B, ref_frame):
    """What the default run reports beside the headline, each measured after the timed region on the same GPU:
    `configs` (the other BASELINE configurations), `accurate_mode` (MD_PREC_F16X2 at the headline batch), `fp32_mode_fps`,
    and `accuracy` of the three Depth Pro modes at 1536^2 against the CPU-baseline frame (one oracle frame serves both)."""
    import torch
    from burn_depth_amd import weights as Wt
    from burn_depth_amd.config import DepthProConfig, Precision
    from burn_depth_amd.depth_pro import DepthPro
    S = cfg.img_size()
    out = {}
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)

    def errors(m):
        if ref_frame is None:
            return None
        d = m.infer(ref_frame["x"].to(tdev))
        rd = ref_frame["depth"]
        err = (d.depth.cpu() - rd).abs()
        rel = err / rd.abs()
        return {"depth_linf": float(err.max()), "depth_max_rel": float(rel.max()), "depth_mean_rel": float(rel.mean()),
                "depth_mean_abs": float(err.mean()), "fovx_deg_abs": float((d.fovx_deg.cpu() - ref_frame["fovx_deg"]).abs().max())}

    def rate(m, b, steps):
        x = ((torch.rand(b, 3, S, S, generator=torch.Generator().manual_seed(1234)) - mean) / std).to(tdev)
        bufs = [torch.empty((b, S, S), dtype=torch.float32, device=tdev)] + [torch.empty((b,), dtype=torch.float32, device=tdev) for _ in range(3)]
        step = lambda: m.infer_into(x, *bufs)  # noqa: E731
        return _timed_steps(step, 2, steps), step

    acc = {"bf16": errors(model)}
    # config 3 as SURVEY 8(d) words it: one image per call (latency)
    fl1, _ = work_model(cfg, 1)
    dt1, step1 = rate(model, 1, 10)
    configs = [{"name": "config 3: Depth Pro [1,3,1536,1536], bf16, B = 1", "value": round(1.0 / dt1, 3), "unit": "frames/s", "ms_per_step": round(dt1 * 1e3, 3),
                "dtype": "bf16", "roofline": _dominant(model, step1, fl1, PEAK_BF16_TFLOPS), "accuracy": acc["bf16"]}]
    # tile-parallel single-image mode (second mode): the device time of one frame on `parts` GPUs, measured window
    # by window on THIS GPU (md_depth_pro_infer_windows issues the launches of each rank in turn) -- a projection, not a
    # multi-GPU measurement: the exchange (100 MB of bf16 tokens + hooks to the root, 1/parts of it per xGMI link) is an estimate
    try:
        x1 = ((torch.rand(1, 3, S, S, generator=torch.Generator().manual_seed(1234)) - mean) / std).to(tdev)
        proj = []
        for parts in (2, 4, 8):
            best = None
            for _ in range(3):
                _o, wms, tms = model.infer_windows(x1, parts, timings=True)
                if best is None or max(wms) + tms < best[0]:
                    best = (max(wms) + tms, wms, tms)
            exch_ms = 100e6 / parts / 50e9 * 1e3 + 0.03  # every peer's shard (100 MB / parts) on its own xGMI link at ~50 GB/s + group latency
            proj.append({"parts": parts, "max_window_ms": round(max(best[1]), 3), "tail_ms": round(best[2], 3), "exchange_ms_estimate": round(exch_ms, 3),
                         "ms_per_frame": round(best[0] + exch_ms, 3), "vs_one_gpu": round(dt1 * 1e3 / (best[0] + exch_ms), 3)})
        out["tile_parallel_projection"] = {
            "what": "DepthPro::infer [1,3,1536,1536] with the 37 ViT sequences split over N GPUs (md_comm_depth_pro_infer_tiles): per-window and "
                    "tail device times measured on ONE GPU, exchange estimated -- NOT a multi-GPU measurement",
            "one_gpu_ms_per_frame": round(dt1 * 1e3, 3), "projection": proj}
    except Exception as ex:  # noqa: BLE001 -- an extra must never lose the headline line
        out["tile_parallel_projection"] = {"error": f"{type(ex).__name__}: {ex}"}
    # the accurate fast mode at the headline batch, and the fp32 parity mode
    c2 = DepthProConfig()
    c2.precision, c2.max_batch = Precision.F16X2, B
    m2 = DepthPro.new(dev, c2, seed=0, init_scheme=Wt.INIT_PARITY).round_weights_to_f16()
    acc["f16x2"] = errors(m2)
    dt2, step2 = rate(m2, B, 5)
    flB, _ = work_model(cfg, B)
    roof2 = _dominant(m2, step2, flB, PEAK_BF16_TFLOPS, steps=2)
    if roof2:  # two MFMA terms per product: the executed rate is twice the algorithmic one
        roof2["executed_tflops"] = round(2 * roof2["achieved"], 2) if roof2["kernel"] != "attention" else None
    out["accurate_mode"] = {"precision": "f16x2", "value": round(B / dt2, 3), "unit": "frames/s", "ms_per_step": round(dt2 * 1e3, 3), "batch_per_gpu": B,
                            "weight_terms": m2.query("weight_terms"), "roofline": roof2,
                            "what": "activations as hi + lo IEEE-half planes (22 bits), f16 checkpoint weights exact MFMA operands: two v_mfma_f32_*_f16 per product, fp32 accumulation"}
    dt2b1, _ = rate(m2, 1, 5)  # config 3 as SURVEY 8(d) words it (B = 1) in the accurate mode
    configs[0]["accurate"] = {"precision": "f16x2", "value": round(1.0 / dt2b1, 3), "unit": "frames/s", "ms_per_step": round(dt2b1 * 1e3, 3), "accuracy": acc["f16x2"]}
    m2.destroy()
    c3 = DepthProConfig()
    c3.precision, c3.max_batch = Precision.F32, 1
    m3 = DepthPro.new(dev, c3, seed=0, init_scheme=Wt.INIT_PARITY).round_weights_to_f16()
    acc["f32"] = errors(m3)
    dt3, _ = rate(m3, 1, 3)
    out["fp32_mode_fps"] = round(1.0 / dt3, 3)
    configs[0]["fp32_mode"] = {"value": round(1.0 / dt3, 3), "unit": "frames/s", "accuracy": acc["f32"]}
    m3.destroy()
    # which number meets north_star's tolerance: the fastest mode whose depth L_inf on the accuracy frame is below 1e-3
    if ref_frame is not None:
        cands = [("bf16", None, acc["bf16"]),  # fps None: the headline `value` (filled in by the caller)
                 ("f16x2", out["accurate_mode"]["value"], acc["f16x2"]), ("f32", out["fp32_mode_fps"], acc["f32"])]
        out["_tolerance_candidates"] = [{"precision": n, "fps": f, "depth_linf": a["depth_linf"], "depth_max_rel": a["depth_max_rel"]} for (n, f, a) in cands if a]
    out["accuracy"] = ({"frame": ref_frame["what"], "vs": "fp32 CPU oracle (oracle/depth_pro_ref.py, the cpu_baseline frame)", "depth_range": ref_frame["range"],
                        "targets": {"north_star_depth_linf": 1e-3, "reference_bar_max_rel": 5e-3}, "modes": acc} if ref_frame is not None else None)
    # Depth-Anything-v3 (BASELINE configs 2 and 5): the throughput mode BASELINE names, and beside it the accurate fast mode
    # (MD_PREC_F16X2) and the fp32 parity mode, each with its depth error against ONE fp32 CPU-oracle frame per configuration
    # (the reference's statistics and bar, example/correctness.rs:1109-1111)
    for (variant, size, precs, names) in (
            ("small", 518, ("bf16",), ("config 2: Depth-Anything-v3 small, 518^2, bf16",)),
            ("metric_large", 1036, ("bf16", "fp8"), ("config 5 (bf16 yardstick): Depth-Anything-v3 large, 1036^2",
                                                    "config 5: Depth-Anything-v3 large, 1036^2, fp8 MFMA linear layers"))):
        try:
            ref = da3_reference_frame(variant, size)
        except Exception as ex:  # noqa: BLE001 -- an extra must never lose the headline line
            ref = None
            print(f"note: DA3 oracle frame failed: {type(ex).__name__}: {ex}", file=sys.stderr)

        def one(prec, steps=10):
            try:
                return measure_da3(dev, tdev, variant, size, prec, steps=steps, ref=ref)
            except Exception as ex:  # noqa: BLE001
                return {"error": f"{type(ex).__name__}: {ex}"}
        accurate = one("f16x2")
        fp32 = one("f32", steps=3)
        for prec, name in zip(precs, names):
            t = one(prec)
            # what a reader must see first (round-5 review, next #7): whether the throughput mode's depth error is inside the reference's own
            # bar (example/correctness.rs:1109-1111), and the number of the mode that is (split-half f16). fp8's error is scale-invariant
            # (profiles/r05_fp8_scale_sensitivity.txt): e4m3 operands as built cannot meet that bar; `value` is the throughput the
            # configuration names, `value_within_reference_bar` the fastest mode inside the bar.
            wb = (t.get("accuracy") or {}).get("within_reference_bar")
            ab = (accurate.get("accuracy") or {}).get("within_reference_bar")
            e = {"name": name, "within_reference_bar": wb,
                 "value_within_reference_bar": (t.get("value") if wb else (accurate.get("value") if ab else None)),
                 "mode_within_reference_bar": (prec if wb else ("f16x2" if ab else None))}
            e.update(t)
            e["accurate"] = dict(accurate, precision="f16x2")
            e["fp32_mode"] = {k: fp32.get(k) for k in ("value", "unit", "ms_per_step", "accuracy", "error") if k in fp32}
            e["reference_bar"] = DA3_REF_BAR
            e["oracle_frame"] = ({"seconds_on_host_cores": ref["seconds"], "depth_range": ref["range"]} if ref else None)
            configs.append(e)
    out["configs"] = configs
    return out


def bench_tile_parallel(args, dev, tdev, world: int, rank: int) -> int:
    """Single-image latency with the ViT stage sharded over the ranks (second mode): every step is ONE
    `DepthPro::infer([B,3,1536,1536])` (B = --batch, default 1) whose input lives on rank 0 and whose outputs arrive on rank 0;
    broadcast of the image, the ranks' windows, the token / hook exchange and the root's decoder are all inside the timed region."""
    import torch
    import torch.distributed as dist
    from burn_depth_amd import weights as Wt
    from burn_depth_amd.config import DepthProConfig, Precision
    from burn_depth_amd.depth_pro import DepthPro
    from burn_depth_amd.parallel import NativeComm
    cfg = {"full": DepthProConfig(), "small": DepthProConfig.small_test(), "tiny": DepthProConfig.tiny_test()}[args.preset]
    cfg.precision = {"bf16": Precision.BF16, "f16": Precision.F16, "f32": Precision.F32, "f16x2": Precision.F16X2}[args.precision]
    B = args.batch if args.batch and args.batch != 8 else 1
    cfg.max_batch = B
    S = cfg.img_size()
    model = DepthPro.new(dev, cfg, seed=0 if rank == 0 else 1 + rank, init_scheme=Wt.INIT_PARITY)
    ncomm = NativeComm.from_torch_distributed(dev) if world > 1 else NativeComm(dev, NativeComm.unique_id(), 1, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ncomm.broadcast_weights(model, root=0)
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t0
    model.round_weights_to_f16()
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    x = ((torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(1234)) - mean) / std).to(tdev) if rank == 0 else None
    last = {}

    def step():
        last["out"] = ncomm.infer_tiles(model, x, (B, S, S), root=0)

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        ok = bool(torch.isfinite(last["out"].depth).all().item())
        emit(({
            "metric": "frames/sec Depth Pro @1536^2 bf16" if args.preset == "full" and args.precision == "bf16" else f"frames/sec Depth Pro preset={args.preset} {args.precision}",
            "value": round(args.steps * B / elapsed, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic (seeded U[0,1) image, ImageNet-normalised; random-init weights rounded to f16)",
            "config": {"workload": f"DepthPro::infer [{B},3,{S},{S}], ONE call per step sharded over the ranks", "global_batch": B,
                       "parallelism": f"tile-parallel x{world}: the 37 B ViT sequences split over the ranks, token + hook exchange to rank 0, decoder on rank 0",
                       "comm": "native md_comm_* (ncclBroadcast of the image, grouped ncclSend / ncclRecv of tokens and hooks)",
                       "ranks_seen": ncomm.ranks_seen()},
            "finite_output": ok, "weight_broadcast_s": round(t_bcast, 4), "roofline": None, "cpu_baseline": None}))
    model.destroy()
    ncomm.destroy()
    if world > 1:
        dist.destroy_process_group()
    return 0


def bench_dry(args, world: int, rank: int) -> int:
    """The launcher / data-path contract without a GPU: gloo instead of RCCL, a CPU stand-in for the engine
    (depth = a per-image reduction, so that every gathered map can be checked against the scattered images)."""
    import torch
    import torch.distributed as dist
    from burn_depth_amd.parallel import shard_range
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    S = {"full": 96, "small": 64, "tiny": 32}[args.preset]
    B = args.batch
    g = torch.Generator().manual_seed(7)
    full = torch.rand(B * world, 3, S, S, generator=g)  # seeded, so every rank can check its shard; only rank 0's copy is sent
    x = torch.empty(B, 3, S, S)
    depth = torch.empty(B, S, S)
    gathered = [torch.empty_like(depth) for _ in range(world)] if rank == 0 else None
    chunks = list(full.split(B, 0)) if rank == 0 else None

    def step():
        if world > 1:
            dist.scatter(x, chunks, src=0)
        else:
            x.copy_(full)
        torch.sum(x, 1, out=depth)
        depth.add_(1.0)
        if world > 1:
            dist.gather(depth, gathered, dst=0)

    hb = None
    if args.native_comm:
        # `--native-comm`: the REAL double-buffer / event bookkeeping of the native path (parallel.NativePipeline) on CPU stand-ins:
        # gloo moves the bytes synchronously, every buffer access is logged against the stream clocks of a happens-before
        # checker -- a wait the GPU path would be missing shows up here as a race
        from burn_depth_amd.parallel import HappensBefore, NativePipeline
        hb = HappensBefore()
        compute = hb.stream("compute")
        nbuf = 2
        xs = [torch.empty(B, 3, S, S) for _ in range(nbuf)]
        depths = [torch.empty(B, S, S) for _ in range(nbuf)]
        gflat = [torch.empty(B * world, S, S) for _ in range(nbuf)] if rank == 0 else None

        class DryComm:
            def scatter_images(self, all_images, shard, root, stream):
                slot = [i for i, t in enumerate(xs) if t is shard][0]
                hb.access(stream, f"scatter -> xs[{slot}]", reads=("global_in",) if rank == root else (), writes=(f"xs[{slot}]",))
                if world > 1:
                    dist.scatter(shard, list(all_images.split(B, 0)) if rank == root else None, src=root)
                else:
                    shard.copy_(all_images)

            def gather_depth(self, shard, all_depth, root, stream):
                slot = [i for i, t in enumerate(depths) if t is shard][0]
                hb.access(stream, f"gather depths[{slot}]", reads=(f"depths[{slot}]",), writes=(f"gathered[{slot}]",) if rank == root else ())
                if world > 1:
                    dist.gather(shard, list(all_depth.split(B, 0)) if rank == root else None, dst=root)
                else:
                    all_depth.copy_(shard)

        def dry_infer(slot, cur):
            hb.access(cur, f"infer xs[{slot}] -> depths[{slot}]", reads=(f"xs[{slot}]",), writes=(f"depths[{slot}]",))
            torch.sum(xs[slot], 1, out=depths[slot])
            depths[slot].add_(1.0)

        pipe = NativePipeline(DryComm(), dry_infer, nbuf, rank, 0, True, True, full if rank == 0 else None, xs, depths, gflat,
                              make_stream=lambda: hb.stream("comm"), make_event=hb.event, current_stream=lambda: compute)

        def step():  # noqa: F811
            pipe.step()

    for _ in range(args.warmup):
        step()
    if hb is not None:
        pipe.drain(lambda: hb.host_join(compute))
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    b, e = shard_range(B * world, rank, world)
    native = None
    if hb is not None:
        pipe.drain(lambda: hb.host_join(compute))
        depth = depths[pipe.last_slot()]
        if rank == 0:
            gathered = [gflat[pipe.last_slot()]]
        races = list(hb.races)
        if world > 1:  # every rank's log counts
            box = [None] * world
            dist.all_gather_object(box, races)
            races = [r for rr in box for r in rr]
        native = {"pipeline": "burn_depth_amd.parallel.NativePipeline on CPU stand-ins (gloo transfers, happens-before checker)",
                  "steps_walked": pipe.k, "races": races}
    ok = torch.equal(depth, full[b:e].sum(1) + 1.0)
    if rank == 0 and (world > 1 or hb is not None):
        ok = ok and torch.equal(torch.cat(gathered, 0), full.sum(1) + 1.0)
    if native is not None:
        ok = ok and not native["races"]
    if rank == 0:
        emit(({"metric": "frames/sec dry run (CPU stand-in, gloo)", "value": round(args.steps * B * world / elapsed, 3), "unit": "frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"dry run [{B},3,{S},{S}] per rank", "batch_per_gpu": B, "global_batch": B * world,
                                     "parallelism": f"dp{world}", "scatter_inputs_from_rank0": world > 1, "gather_depth_to_rank0": world > 1},
                          "finite_output": bool(ok), "dry_run": True, "native_comm": native}))
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


def bench_da3(args, dev, tdev, world, rank) -> int:
    """frames/s of DepthAnything3::infer (metric_large, mono head) on synthetic [B,3,S,S]; same timing
    contract as the Depth Pro path (barrier + synchronize, max over ranks, weak scaling)."""
    import torch
    import torch.distributed as dist
    from burn_depth_amd import weights as Wt
    from burn_depth_amd.config import DepthAnything3Config, Precision
    from burn_depth_amd.depth_anything3 import DepthAnything3
    small = args.model == "da3_small"
    cfg = DepthAnything3Config.small() if small else DepthAnything3Config.metric_large()
    if args.image_size:
        cfg.image_size = args.image_size
    cfg.precision = {"bf16": Precision.BF16, "f16": Precision.F16, "f32": Precision.F32, "fp8": Precision.FP8, "f16x2": Precision.F16X2}[args.precision]
    cfg.max_batch = args.batch
    S, B = cfg.image_size, args.batch
    # the reference's DA3 checkpoints are f16 records (example/correctness.rs:977): measured on weights such a record can hold
    model = DepthAnything3.new(dev, cfg, seed=0, init_scheme=Wt.INIT_PARITY).round_weights_to_f16()
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
               "data": "synthetic (seeded normal images; random-init weights rounded to f16 like the reference's f16 records)",
               "config": {"workload": f"DepthAnything3::infer [{B},3,{S},{S}] per GPU, " + ("small (dual head, all outputs)" if small else "metric_large (mono head)"), "batch_per_gpu": B,
                          "global_batch": B * world, "parallelism": f"dp{world}"},
               "backbone_tflops_algorithmic": round(vit_flops / B / 1e12, 3),
               "attention_tflops": round(4.0 * B * v.num_heads * NT * NT * 64 * depth_n / (attn_ms * 1e-3) / 1e12, 1) if attn_ms else None,
               "kernels": kernels}
        emit(out)
    model.destroy()
    if world > 1:
        dist.destroy_process_group()
    return 0


def side_kernels(dev, tdev):
    """The HBM-bound stand-alone kernels at the shapes of the reference's interpolation bench (bench/interpolate.rs:32-113:
    [1,3,1536,1536] -> 768^2 / 384^2, [1,1,1536,1536] -> 1080x1920) plus the DA3 NHWC upsample (dpt.rs:611-631: the
    128-channel map at 592^2 -> 1036^2 of config 5, 296^2 -> 518^2 of config 2). HIP events on the launch stream;
    algorithmic bytes = input read once + output written once."""
    import torch
    from burn_depth_amd import ops
    res = {}

    def timed(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters

    def entry(ms, nbytes):
        return {"ms_per_step": round(ms, 4), "launches_per_step": 1, "gbs": round(nbytes / ms / 1e6, 1),
                "frac_hbm_peak": round(nbytes / ms / 1e6 / PEAK_HBM_GBS, 4)}

    for (shape, out, m) in [((1, 3, 1536, 1536), (768, 768), 0), ((1, 3, 1536, 1536), (384, 384), 0), ((1, 1, 1536, 1536), (1080, 1920), 0),
                            ((8, 3, 1536, 1536), (768, 768), 0), ((8, 1, 1536, 1536), (1080, 1920), 1)]:
        x = torch.randn(shape, device=tdev)
        y = torch.empty(shape[:2] + out, device=tdev)
        ms = timed(lambda: ops.resize_bilinear_into(dev, x, y, m))
        res[f"side:resize_bilinear {list(shape)}->{list(out)} m{m}"] = entry(ms, (x.numel() + y.numel()) * 4.0)
    for (B, H, C, OH) in [(1, 592, 128, 1036), (8, 296, 128, 518), (1, 148, 256, 296)]:
        x = torch.randn(B, H, H, C, device=tdev).to(torch.bfloat16)
        y = torch.empty(B, OH, OH, C, device=tdev, dtype=torch.bfloat16)
        ms = timed(lambda: ops.resize_nhwc_into(dev, x, y, 1))
        res[f"side:resize_nhwc bf16 [{B},{H},{H},{C}]->{OH}^2 align_corners"] = entry(ms, (x.numel() + y.numel()) * 2.0)
    return res


def accuracy_report(dev, precision):
    """Depth error of this precision mode against the fp32 CPU oracle on one seeded frame of the CI preset (ViT-L, 512^2;
    src/lib.rs:102-112): the reference's own statistics (example/correctness.rs:486-509, thresholds :887-897)."""
    import torch
    from burn_depth_amd import weights as Wt
    from burn_depth_amd.config import DepthProConfig
    from burn_depth_amd.depth_pro import DepthPro
    from oracle import depth_pro_ref as R
    cfg = DepthProConfig.small_test()
    cfg.precision = precision
    m = DepthPro.new(dev, cfg, seed=0, init_scheme=Wt.INIT_PARITY)
    W = R.weights_to_torch(Wt.generate_depth_pro_weights(cfg, 0, Wt.INIT_PARITY))
    torch.manual_seed(0)
    x = (torch.rand(1, 3, 512, 512) - torch.tensor(R.MEAN).view(1, 3, 1, 1)) / torch.tensor(R.STD).view(1, 3, 1, 1)
    d = m.infer(x.cuda()).depth.cpu()
    m.destroy()
    with torch.no_grad():
        rd = R.infer(x, W, cfg)["depth"]
    err = (d - rd).abs()
    return {"preset": "small (ViT-L, 512^2)", "vs": "fp32 CPU oracle", "depth_max_rel": float((err / rd.abs()).max()), "depth_mean_rel": float((err / rd.abs()).mean()),
            "depth_linf": float(err.max()), "depth_mean_abs": float(err.mean()), "depth_range": [float(rd.min()), float(rd.max())],
            "reference_bar": {"max_abs": 5e-3, "mean_abs": 1e-3, "max_rel": 5e-3}}


def pmc_traffic(kernel: str, B: int, args):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE are collected in separate runs of this same command; tools/pmc_traffic.py applies the
    gfx950 corrections of MI355X_MICROARCH.md and writes profiles/rNN_traffic.json). None if the
    passes were made for another batch/precision."""
    for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)
            if t.get("batch") == B and t.get("precision") == args.precision and t.get("preset") == args.preset:
                v = t["kernels"].get(kernel, {}).get("hbm_bytes_per_launch")
                if v is not None:
                    return v, f"profiles/{name}: committed rocprofv3 --pmc passes of this command (FETCH_SIZE x 2 + WRITE_SIZE, tools/pmc_traffic.py); NOT re-measured by this run"
        except (OSError, ValueError, KeyError):
            pass
    return None, None


def cpu_baseline(cfg, budget_s: float):
    """The CPU oracle (a port: the Rust reference cannot be built here) on this process's cores: ONE whole frame of
    `DepthPro::infer` at [1,3,S,S] (bench/inference.rs:21-48 times exactly that call). The frame is BASELINE config 3-(ii) --
    the seeded U[0,1) image on the seed-0 weights rounded to f16 like the reference's checkpoint records (mod.rs:206) -- so
    that the same oracle output also serves as the reference of the `accuracy` object (the arithmetic cost does not depend
    on the pixel values; config 1's zeros frame costs the same). A 2-tile ViT probe first predicts the frame time; beyond
    `budget_s` the probe-scaled estimate is reported and no reference frame is returned.
    Returns (cpu_baseline object, reference frame or None)."""
    import torch
    from burn_depth_amd import weights as Wt
    from oracle import depth_pro_ref as R
    # use this process's CPU share, not every core the host shows: the affinity mask cut by the cgroup CPU quota (a 1-GPU box of
    # this pool shows 256 CPUs under a 16-core quota; on 64 threads the oracle's GEMMs ran at 0.67 of their 16-thread rate,
    # profiles/r05_host_probe.txt -- rounds 1-4 reported "cores": 64 for what were 16 cores' worth of CPU time)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota[0] != "max":
            avail = min(avail, max(1, int(int(quota[0]) / int(quota[1]))))
    except (OSError, ValueError, IndexError):
        pass
    threads = max(1, min(avail, 64))
    torch.set_num_threads(threads)
    v = cfg.patch_vit()
    fl, _ = work_model(cfg, 1)
    frame_flops = sum(fl.values())
    nseq = 25 + 9 + 1 + 1 + (1 if cfg.fov_encoder_preset else 0)
    vit_flops_per_tile = (fl["patch_embed"] + fl["qkv_gemm"] + fl["attention"] + fl["proj_gemm"] + fl["fc1_gemm"] + fl["fc2_gemm"]) / nseq
    W = {k: R.f16_round(t) for k, t in R.weights_to_torch(Wt.generate_depth_pro_weights(cfg, 0, Wt.INIT_PARITY)).items()}
    with torch.no_grad():
        tiles = 2
        x = torch.zeros(tiles, 3, v.img_size, v.img_size)
        R.vit_forward(x[:1], W, "encoder.patch_encoder", v, v.encoder_feature_layer_ids)  # warm the thread pool / allocator
        t0 = time.perf_counter()
        R.vit_forward(x, W, "encoder.patch_encoder", v, v.encoder_feature_layer_ids)
        dt = time.perf_counter() - t0
        probe_tflops = tiles * vit_flops_per_tile / dt / 1e12
        est_frame_s = frame_flops / (probe_tflops * 1e12)
        if est_frame_s <= budget_s:
            S = cfg.img_size()
            torch.manual_seed(0)
            xin = (torch.rand(1, 3, S, S) - torch.tensor(R.MEAN).view(1, 3, 1, 1)) / torch.tensor(R.STD).view(1, 3, 1, 1)
            t0 = time.perf_counter()
            out = R.infer(xin, W, cfg)
            frame_s = time.perf_counter() - t0
            ok = tuple(out["depth"].shape) == (1, S, S) and bool(torch.isfinite(out["depth"]).all())
            what = f"seeded U[0,1) image [1,3,{S},{S}] (BASELINE config 3-(ii)), seed-0 weights rounded to f16 (an f16 checkpoint, mod.rs:206)"
            ref = {"x": xin, "depth": out["depth"], "fovx_deg": out["fovx_deg"], "what": what,
                   "range": [float(out["depth"].min()), float(out["depth"].max())]}
            return ({"value": round(1.0 / frame_s, 5), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                     "sample": f"full frame: oracle DepthPro::infer on the {what}, {frame_flops / 1e12:.2f} TFLOP in {frame_s:.1f} s",
                     "seconds_per_frame": round(frame_s, 2), "finite_output": ok, "cpu_tflops": round(frame_flops / frame_s / 1e12, 3)}, ref)
    return ({"value": round(1.0 / est_frame_s, 5), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
             "sample": f"oracle ViT-L/16 forward on {tiles} of {nseq} tiles in {dt:.2f} s, scaled by algorithmic FLOPs "
                       f"({frame_flops / 1e12:.2f} TFLOP/frame; a whole frame was predicted to take {est_frame_s:.0f} s > budget {budget_s:.0f} s)",
             "sample_seconds": round(dt, 3), "cpu_tflops": round(probe_tflops, 3)}, None)


if __name__ == "__main__":
    sys.exit(main())
