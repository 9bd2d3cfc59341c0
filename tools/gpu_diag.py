"""One-shot GPU diagnostic: runs every stand-alone operator and the end-to-end Depth Pro path
against the CPU oracle and prints an error table (never raises; meant for `gpurun` logs).

usage: python tools/gpu_diag.py [--skip-small] [--full]
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time
import traceback

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from burn_depth_amd import _lib, ops  # noqa: E402
from burn_depth_amd import weights as Wt  # noqa: E402
from burn_depth_amd.config import DepthProConfig, Precision  # noqa: E402
from burn_depth_amd.depth_pro import DepthPro, Device  # noqa: E402
from oracle import depth_pro_ref as R  # noqa: E402

RESULTS = []
_ORACLE_CACHE = {}  # (tag, ...) -> oracle result: precision sweeps of one configuration share ONE CPU oracle frame


def cfg_key(cfg):
    """Everything of a config that the fp32 oracle depends on (not the engine's precision mode or batch capacity)."""
    return tuple(sorted((k, str(v)) for k, v in vars(cfg).items() if k not in ("precision", "max_batch")))


def cached(key, fn):
    if key in _PENDING_JOBS:  # a registered full-size frame: compute it together with every other registered one, in child processes
        prefetch_processes()
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = fn()
    v = _ORACLE_CACHE[key]
    if isinstance(v, BaseException):
        raise v
    return v


# ---- oracle frames in child processes -----------------------------------------------------------------------------------
# The full-size CPU-oracle frames (7-19 TFLOP each) were 60 % of the GPU suite's wall time, one after the other on all host
# cores. `prefetch_processes` computes the frames a run will need CONCURRENTLY, one child process per frame on a share of the
# cores (tools/oracle_frames.py: no GPU, no library), when the first full-size test asks for one; the tests that need them run
# last (tests/conftest.py moves `fullsize` tests to the end). Only the CHECKER moves -- what is compared, and against what, is
# unchanged. (A first form -- a background THREAD under the earlier tests -- made the whole suite slower: two 64-thread OpenMP teams
# on the same cores, profiles/r05_pytest_gpu_thread_prefetch.log.)
_PENDING_JOBS = {}   # cache key -> list of (job string, slot) the key's value is assembled from


from oracle_frames import host_cpus  # noqa: E402  (quota-aware core count)


def host_mem_gb() -> float:
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 64.0


def register_frame(key, jobs):
    """`key`: the oracle-cache key a test will ask for; `jobs`: {slot: job string of tools/oracle_frames.py}."""
    if key not in _ORACLE_CACHE:
        _PENDING_JOBS[key] = dict(jobs)


def prefetch_processes():
    """Computes every registered frame now, in parallel child processes, and fills the oracle cache."""
    import subprocess
    import tempfile
    pending = {k: v for k, v in _PENDING_JOBS.items() if k not in _ORACLE_CACHE}
    _PENDING_JOBS.clear()
    jobs = sorted({j for v in pending.values() for j in v.values()}, key=lambda j: 0 if j.endswith(":q") else 1)  # the slowest first
    if not jobs:
        return
    cpus, mem = host_cpus(), host_mem_gb()
    width = max(1, min(len(jobs), int(mem // 24) or 1, max(1, cpus // 8)))  # a Depth Pro frame peaks near 20 GB of host memory
    threads = max(1, cpus // width)
    tmp = tempfile.mkdtemp(prefix="oracle_frames_")
    t0 = time.time()
    print(f"      [oracle frames] {len(jobs)} frames in {width} child processes x {threads} threads ({cpus} usable cores, {mem:.0f} GB available)", flush=True)
    running, results, todo = [], {}, list(jobs)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    while todo or running:
        while todo and len(running) < width:
            j = todo.pop(0)
            out = os.path.join(tmp, f"{len(results) + len(running)}.pt")
            pr = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "oracle_frames.py"), "--job", j, "--out", out, "--threads", str(threads)],
                                  env=env, cwd=ROOT)
            running.append((j, out, pr))
        time.sleep(0.5)
        for item in list(running):
            j, out, pr = item
            rc = pr.poll()
            if rc is None:
                continue
            running.remove(item)
            if rc != 0 or not os.path.exists(out):
                results[j] = RuntimeError(f"oracle frame `{j}` failed in its child process (exit code {rc})")
            else:
                results[j] = torch.load(out, weights_only=False)
                os.remove(out)
    try:
        os.rmdir(tmp)
    except OSError:
        pass
    print(f"      [oracle frames] ready after {time.time() - t0:.1f}s", flush=True)
    for key, slots in pending.items():
        parts = {slot: results[j] for slot, j in slots.items()}
        bad = [v for v in parts.values() if isinstance(v, BaseException)]
        _ORACLE_CACHE[key] = bad[0] if bad else _ASSEMBLE[key[0]](parts)


def _assemble_full(parts):
    fp = parts["fp32"]
    return dict(x=fp["x"], rgb=fp["rgb"], ref=fp["out"], refq=parts["q"]["out"] if "q" in parts else None)


_ASSEMBLE = {"full": _assemble_full, "da3": lambda parts: parts["fp32"]["out"]}


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    if a.shape != b.shape:
        return float("inf")
    if not torch.isfinite(a).all():
        return float("nan")
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def record(name, err, tol, extra=""):
    ok = err == err and err <= tol
    RESULTS.append((name, err, tol, ok, extra))
    print(f"[{'OK ' if ok else 'BAD'}] {name:58s} err={err:.3e} tol={tol:.1e} {extra}", flush=True)


def guarded(name):
    def deco(fn):
        def run(*a, **k):
            try:
                t = time.time()
                fn(*a, **k)
                torch.cuda.synchronize()
                print(f"      ({name}: {time.time() - t:.2f}s)", flush=True)
            except Exception as e:  # noqa: BLE001
                traceback.print_exc()
                RESULTS.append((name, float("nan"), 0.0, False, f"EXC {type(e).__name__}: {e}"))
                print(f"[BAD] {name}: exception {e}", flush=True)
        return run
    return deco


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def h16(x):
    return x.to(torch.float16).to(torch.float32)


# MFMA operand rounding per precision mode (Precision.BF16 = 0, F32 = 1, F16 = 3)
ROUND = {0: bf, 1: (lambda x: x), 3: h16, 4: R.f16x2_round}
PNAME = {0: "bf16", 1: "f32", 2: "fp8", 3: "f16", 4: "f16x2"}


@guarded("resize")
def check_resize(dev):
    g = torch.Generator().manual_seed(1)
    for (shape, out, m) in [((1, 1, 2, 2), (4, 4), 0), ((1, 1, 2, 2), (4, 4), 1), ((1, 1, 2, 2), (3, 1), 0), ((1, 1, 2, 2), (3, 1), 1),
                            ((2, 3, 36, 54), (96, 96), 0), ((2, 3, 96, 96), (48, 48), 0), ((1, 3, 96, 96), (24, 24), 0),
                            ((1, 1, 96, 96), (36, 54), 0), ((1, 2, 7, 5), (13, 17), 1), ((1, 2, 5, 5), (5, 5), 0),
                            # output rows wider than one 1024-column chunk of the row-per-workgroup kernel, both modes
                            ((1, 2, 40, 700), (23, 1500), 0), ((2, 1, 33, 1300), (50, 2100), 1), ((1, 1, 64, 2200), (32, 1100), 0)]:
        x = torch.rand(shape, generator=g)
        if shape == (1, 1, 2, 2):
            x = torch.tensor([1.0, 2.0, 3.0, 4.0]).reshape(shape) if out == (4, 4) else torch.tensor([4.0, 1.0, 0.0, 2.0]).reshape(shape)
        want = R.resize_bilinear(x, out, m)
        got = ops.resize_bilinear(dev, x.cuda(), out, m).cpu()
        exact = torch.equal(got, want)
        record(f"resize {shape}->{out} m{m}", rel_err(got, want), 0.0 if m == 0 else 1e-6, "bit-exact" if exact else "")


@guarded("rgb")
def check_rgb(dev):
    from burn_depth_amd.inference import rgb_to_input_tensor
    rgb = bytes(np.random.RandomState(0).randint(0, 256, size=37 * 21 * 3, dtype=np.uint8).tolist())
    want = R.rgb_to_input_tensor(rgb, 37, 21)
    got = rgb_to_input_tensor(rgb, 37, 21, dev).cpu()
    record("rgb_to_input 37x21", rel_err(got, want), 0.0, "bit-exact" if torch.equal(got, want) else "")
    kat = rgb_to_input_tensor(bytes([0, 255, 128, 255, 0, 128]), 1, 2, dev).cpu().flatten()
    exp = torch.tensor([-2.1179039, 2.2489083, 2.4285715, -2.0357141, 0.42649257, 0.42649257])
    record("rgb_to_input KAT", (kat - exp).abs().max().item(), 1e-6)


@guarded("split/merge")
def check_split_merge(dev):
    x = torch.arange(2 * 3 * 512 * 512, dtype=torch.float32).reshape(2, 3, 512, 512)
    for ov in (0.25, 0.5, 0.0):
        want, steps, stride = R.split(x, 128, ov)
        got, st = ops.split(dev, x.cuda(), 128, ov)
        record(f"split ov={ov} steps={st}", rel_err(got, want), 0.0)
    tiles = torch.rand(25 * 2, 5, 8, 8)
    record("merge 5x5 pad1", rel_err(ops.merge(dev, tiles.cuda(), 2, 1), R.merge(tiles, 2, 1)), 0.0)
    tiles = torch.rand(9 * 2, 4, 24, 24)
    record("merge 3x3 pad6", rel_err(ops.merge(dev, tiles.cuda(), 2, 6), R.merge(tiles, 2, 6)), 0.0)
    tiles = torch.rand(25, 4, 24, 24)
    record("merge 5x5 pad3", rel_err(ops.merge(dev, tiles.cuda(), 1, 3), R.merge(tiles, 1, 3)), 0.0)
    tiles = torch.rand(3, 4, 8, 8)
    record("merge 1x1", rel_err(ops.merge(dev, tiles.cuda(), 3, 2), R.merge(tiles, 3, 2)), 0.0)


@guarded("layernorm")
def check_layernorm(dev):
    g = torch.Generator().manual_seed(2)
    for rows, D in [(7, 1024), (580, 1024), (33, 256), (5, 384), (3, 64)]:
        x = torch.randn(rows, D, generator=g) * 3 + 0.5
        ga, be = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1
        want = F.layer_norm(x, (D,), ga, be, 1e-6)
        got = ops.layernorm(dev, x.cuda(), ga.cuda(), be.cuda(), 1e-6)
        record(f"layernorm {rows}x{D}", rel_err(got, want), 2e-6)
    x = torch.randn(9, 256, generator=g)
    record("layernorm non-affine", rel_err(ops.layernorm(dev, x.cuda(), None, None, 1e-5), F.layer_norm(x, (256,), None, None, 1e-5)), 2e-6)


@guarded("linear")
def check_linear(dev):
    g = torch.Generator().manual_seed(3)
    cases = [(300, 256, 128, _lib.TILE_128x128), (300, 256, 128, _lib.TILE_256x256), (577, 3072, 1024, _lib.TILE_AUTO),
             (1160, 1024, 4096, _lib.TILE_AUTO), (64, 32, 192, _lib.TILE_256x32), (1000, 32, 128, _lib.TILE_256x32),
             (129, 132, 64, _lib.TILE_128x128), (513, 260, 320, _lib.TILE_256x256), (2000, 1024, 1024, _lib.TILE_256x256),
             (2000, 1024, 1024, _lib.TILE_128x128), (300, 256, 64, _lib.TILE_256x256), (300, 256, 128, _lib.TILE_256x256), (21349, 1024, 1024, _lib.TILE_256x256),
             (700, 3072, 64, _lib.TILE_256x256), (700, 512, 128, _lib.TILE_256x256), (700, 512, 192, _lib.TILE_256x256),
             (300, 256, 128, _lib.TILE_128x64), (129, 132, 192, _lib.TILE_128x64), (1370, 384, 1536, _lib.TILE_64x64), (65, 68, 64, _lib.TILE_64x64),
             (1370, 384, 384, _lib.TILE_AUTO), (1370, 64, 576, _lib.TILE_AUTO),
             # small launches behind a long contraction: the 64 x 64 kernel splits K over two / four wave groups (gemm_kernel's KSPLIT)
             (361, 384, 3456, _lib.TILE_64x64), (300, 128, 1024, _lib.TILE_64x64), (1369, 64, 1728, _lib.TILE_64x64), (1370, 96, 768, _lib.TILE_64x64), (65, 68, 2048, _lib.TILE_AUTO), (1370, 384, 1536, _lib.TILE_AUTO)]
    for prec, pname in [(0, "bf16"), (1, "f32"), (3, "f16")]:
        rnd = ROUND[prec]
        for (M, N, K, tile) in cases:
            x = torch.randn(M, K, generator=g)
            w = torch.randn(N, K, generator=g) / math.sqrt(K)
            b = torch.randn(N, generator=g)
            x, w = rnd(x), rnd(w)
            want = F.linear(x.double(), w.double(), b.double()).float()
            got = ops.linear(dev, x.cuda(), w.cuda(), b.cuda(), 0, prec, tile)
            record(f"linear {pname} M{M} N{N} K{K} tile{tile}", rel_err(got, want), 2e-5)
        x = rnd(bf(torch.randn(200, 256, generator=g)))
        w = rnd(bf(torch.randn(512, 256, generator=g) / 16))
        b = torch.randn(512, generator=g)
        record(f"linear {pname} gelu", rel_err(ops.linear(dev, x.cuda(), w.cuda(), b.cuda(), 2, prec), F.gelu(F.linear(x, w, b))), 2e-5)
        record(f"linear {pname} relu nobias", rel_err(ops.linear(dev, x.cuda(), w.cuda(), None, 1, prec), F.relu(F.linear(x, w))), 2e-5)


def check_storage_epilogues(dev, prec=0):
    """The store epilogues the ENGINE uses (2-byte output through the staged / pixel-shuffle paths of the 256x256
    kernel), at sizes that select that kernel and leave partial tiles. Tolerance = output rounding of the storage type."""
    g = torch.Generator().manual_seed(11)
    tol = {0: 6e-3, 4: 2e-5}.get(prec, 8e-4)  # f16x2: hi + lo planes carry 22 bits: fp32 accumulation order is what is left
    bf = ROUND[prec]  # noqa: F811 -- operands representable in the mode's storage type
    pn = PNAME[prec]
    for (M, N, K, act, has_bias) in [(2000, 1024, 1024, 0, True), (513, 264, 320, 0, True), (700, 512, 128, 2, True),
                                     (1300, 256, 192, 1, False), (513, 260, 320, 0, True), (21349, 1024, 1024, 2, True)]:
        x = bf(torch.randn(M, K, generator=g))
        w = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
        b = torch.randn(N, generator=g) if has_bias else None
        want = F.linear(x.double(), w.double(), b.double() if has_bias else None)
        want = F.gelu(want) if act == 2 else (F.relu(want) if act == 1 else want)
        got = ops.linear(dev, x.cuda(), w.cuda(), b.cuda() if has_bias else None, act, prec, _lib.TILE_256x256, storage_out=True)
        record(f"linear {pn}-out M{M} N{N} K{K} act{act}", rel_err(got, want.float()), tol)
    for (B, Cin, H, W, Cout) in [(2, 128, 160, 120, 256), (1, 64, 33, 17, 32)]:
        x = bf(torch.randn(B, Cin, H, W, generator=g))
        w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
        b = torch.randn(Cout, generator=g)
        want = F.conv2d(x.double(), w.double(), b.double(), padding=1).float()
        got = ops.conv3x3(dev, x.cuda(), w.cuda(), b.cuda(), False, prec, storage_out=True)
        record(f"conv3x3 {pn}-out B{B} {Cin}->{Cout} {H}x{W}", rel_err(got, want), tol)
    for (B, Cin, H, W, Cout) in [(2, 256, 97, 88, 128), (1, 128, 150, 130, 64), (2, 256, 12, 10, 128)]:
        x = bf(torch.randn(B, Cin, H, W, generator=g))
        w = bf(torch.randn(Cin, Cout, 2, 2, generator=g) / math.sqrt(Cin))
        b = torch.randn(Cout, generator=g)
        want = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2).float()
        got = ops.deconv2x2(dev, x.cuda(), w.cuda(), b.cuda(), prec, storage_out=True)
        record(f"deconv2x2 {pn}-out B{B} {Cin}->{Cout} {H}x{W}", rel_err(got, want), tol)


def q8_mul(x, inv_scale):
    """e4m3 operand emulation with the device's arithmetic: fp32 multiply by the reciprocal scale, saturate at
    +-448 (the engine clamps before converting), torch's e4m3fn cast (bit-identical to v_cvt_pk_fp8_f32 incl.
    subnormals -- checked elementwise on the GPU)."""
    return (x.float() * inv_scale).clamp(-448, 448).to(torch.float8_e4m3fn).float()


def q8_rows(w):
    """per-output-row weight quantisation as pack_fp8_rows_kernel does it: scale = amax * (1/448), inv = 1/scale (fp32)"""
    amax = w.abs().amax(1, keepdim=True).float()
    sc = torch.where(amax > 0, amax * torch.tensor(1.0 / 448.0, dtype=torch.float32), torch.ones_like(amax))
    return q8_mul(w, 1.0 / sc), sc


FP8_ACT_SCALE = float(np.float32(8.0) / np.float32(448.0))  # static scale of LayerNorm / attention outputs


def check_linear_fp8(dev):
    """MD_PREC_FP8 GEMM: e4m3 operands (activations on the static scale 8/448, weights scaled per output row),
    fp32 accumulation. Compared with the same quantisation done in torch: only accumulation order differs."""
    g = torch.Generator().manual_seed(5)
    xs = torch.tensor(FP8_ACT_SCALE, dtype=torch.float32)
    for (M, N, K, tile, act) in [(300, 256, 128, _lib.TILE_128x128, 0), (300, 256, 256, _lib.TILE_256x256, 0), (1370, 3072, 1024, _lib.TILE_AUTO, 0),
                                 (2740, 4096, 1024, _lib.TILE_256x256, 2), (2740, 1024, 4096, _lib.TILE_256x256, 0), (513, 260, 384, _lib.TILE_AUTO, 0),
                                 (300, 256, 256, _lib.TILE_128x64, 0), (1370, 384, 1536, _lib.TILE_64x64, 2), (5477, 1024, 1024, _lib.TILE_AUTO, 0)]:
        x = torch.randn(M, K, generator=g) * 1.5
        w = torch.randn(N, K, generator=g) / math.sqrt(K)
        b = torch.randn(N, generator=g)
        xq = q8_mul(x, 1.0 / xs)
        wq, wsc = q8_rows(w)
        want = (xq.double() @ wq.double().t()).float() * (xs * wsc.t()) + b
        if act == 2:
            want = F.gelu(want)
        got = ops.linear(dev, x.cuda(), w.cuda(), b.cuda(), act, 2, tile)
        record(f"linear fp8 M{M} N{N} K{K} tile{tile} act{act}", rel_err(got, want), 2e-4 if act == 2 else 5e-5)
        full = F.linear(x, w, b)
        if act == 2:
            full = F.gelu(full)
        record(f"linear fp8 M{M} N{N} K{K} quantisation error vs fp32 (informative)", rel_err(got, full), 8e-2)


def attn_ref(qkv, heads, quant):
    """fp64 attention with the engine's operand roundings: q is rounded AFTER the softmax scale is folded in
    (oracle round_q_prescaled), k and v as stored, P before P.V."""
    T, N, _ = qkv.shape
    q, k, v = qkv.reshape(T, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    q, k, v = R.round_q_prescaled(q, quant).double(), quant(k).double(), quant(v).double()
    s = (q @ k.transpose(-2, -1)) * 0.125
    pu = torch.exp(s - s.amax(-1, keepdim=True))
    o = (quant(pu.float()).double() @ v) / pu.sum(-1, keepdim=True)
    return o.transpose(1, 2).reshape(T, N, heads * 64).float()


def mean_rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().mean() / (b.abs().mean() + 1e-12)).item()


@guarded("attention")
def check_attention(dev):
    """Fused attention (bf16 / f16 operands, P rounded to the operand type before P.V, output stored in the operand type)
    against an fp64 reference that rounds P the same way. The largest error the output rounding alone can cause is one
    ulp of the largest output (2^-8 bf16, 2^-11 f16); the MEAN error is the sensitive check (a 1 % kernel error shows as
    1e-2 there)."""
    g = torch.Generator().manual_seed(4)
    for prec, rnd, tol_max, tol_mean in [(0, bf, 8e-3, 2.5e-3), (3, h16, 1e-3, 4e-4)]:
        pn = PNAME[prec]
        # longer sequences: exact multiples of the tile, a one-key last tile, 64 tiles
        for (T, N, heads) in [(2, 65, 4), (3, 577, 2), (1, 577, 16), (2, 64, 1), (1, 130, 3), (1, 1370, 2), (2, 1024, 2), (1, 1025, 1),
                              (1, 2000, 3), (1, 4095, 1), (1, 4096, 1)]:
            qkv = torch.randn(T, N, 3 * heads * 64, generator=g)
            qkv[..., :heads * 64] *= 2.0
            want = attn_ref(qkv, heads, rnd)
            got = ops.attention(dev, qkv.cuda(), heads, prec)
            record(f"attention {pn} T{T} N{N} h{heads} max", rel_err(got, want), tol_max)
            # the mean error creeps up with the key count (2.43e-3 at 577 keys, 2.52e-3 at 4096 in bf16, with or without the key split)
            record(f"attention {pn} T{T} N{N} h{heads} mean", mean_rel(got, want), tol_mean * (1.06 if N > 2048 else 1.0))
        # The data-dependent branches (cdna guide rule 26): the bf16 kernel's fast body (no maximum) must hand over to the
        # running-maximum body (a) at tile 0 when a row maximum is outside +-64 log2 units, (b) at a later tile when a
        # row sum reaches 2^100, and the running-maximum body must rescale more than once (c). Logits in log2 units =
        # q.k / 8 * 1.4427.
        def spike(label, edit, rows, N=577):
            qkv = torch.randn(1, N, 3 * 64, generator=g)
            edit(qkv)
            want = attn_ref(qkv, 1, rnd)
            got = ops.attention(dev, qkv.cuda(), 1, prec)
            record(f"attention {pn} {label} finite", 0.0 if bool(torch.isfinite(got).all()) else 1.0, 0.0)
            record(f"attention {pn} {label} max", rel_err(got, want), tol_max)
            record(f"attention {pn} {label} rows {rows}", rel_err(got[0, rows], want[0, rows]), tol_max)

        def late_keys(qkv):  # query 3 meets a key worth ~69 log2 units at token 500 and ~104 at token 570 (tile 8): case (b), (c)
            qkv[0, 500, 64:128] = qkv[0, 3, :64] * 6.0
            qkv[0, 570, 64:128] = qkv[0, 3, :64] * 9.0
            qkv[0, 40, :64] *= 8.0

        def first_tile(qkv):  # query 5 against key 7 in tile 0: ~81 log2 units: case (a); then a larger one late: case (c)
            qkv[0, 7, 64:128] = qkv[0, 5, :64] * 7.0
            qkv[0, 400, 64:128] = qkv[0, 5, :64] * 12.0

        def all_low(qkv):  # every score of query 9 is very negative (its keys are anti-aligned): row maximum ~ -73 << -64
            qkv[0, :, 64:128] = -qkv[0, 9:10, :64] * 0.9 + 0.05 * qkv[0, :, 64:128]
            qkv[0, 9, :64] *= 7.0

        spike("late keys", late_keys, [3, 40, 100])
        spike("first tile", first_tile, [5, 6, 300])
        spike("all low", all_low, [9, 10, 576])

        # the same hand-overs on a 1370-key sequence (22 tiles): a range failure deep inside the walk, a row sum overflowing near
        # its end, and the all-low row
        def first_tile_group1(qkv):
            qkv[0, 390, 64:128] = qkv[0, 5, :64] * 7.0
            qkv[0, 1200, 64:128] = qkv[0, 5, :64] * 12.0

        def late_keys_group3(qkv):
            qkv[0, 1200, 64:128] = qkv[0, 3, :64] * 6.0
            qkv[0, 1330, 64:128] = qkv[0, 3, :64] * 9.0
            qkv[0, 40, :64] *= 8.0

        spike("1370 keys, out-of-range key at 390", first_tile_group1, [5, 6, 300, 1369], N=1370)
        spike("1370 keys, late keys", late_keys_group3, [3, 40, 1000], N=1370)
        spike("1370 keys, all low", all_low, [9, 10, 1369], N=1370)
    for (T, N, heads) in [(2, 65, 4), (3, 577, 2), (1, 130, 3), (1, 1370, 2)]:
        qkv = torch.randn(T, N, 3 * heads * 64, generator=g)
        qkv[..., :heads * 64] *= 2.0
        record(f"attention f32  T{T} N{N} h{heads}", rel_err(ops.attention(dev, qkv.cuda(), heads, 1), attn_ref(qkv, heads, R.identity)), 2e-5)


@guarded("conv")
def check_convs(dev):
    g = torch.Generator().manual_seed(5)
    for prec, pname, tol in [(0, "bf16", 2e-5), (1, "f32", 2e-5), (3, "f16", 2e-5)]:
        rnd = ROUND[prec]
        for (B, Cin, H, W, Cout) in [(1, 64, 16, 16, 64), (2, 128, 24, 20, 256), (1, 256, 48, 48, 256), (1, 64, 33, 17, 32)]:
            x = torch.randn(B, Cin, H, W, generator=g)
            w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
            b = torch.randn(Cout, generator=g)
            x, w = rnd(x), rnd(w)
            want = F.conv2d(x.double(), w.double(), b.double(), padding=1).float()
            got = ops.conv3x3(dev, x.cuda(), w.cuda(), b.cuda(), False, prec)
            record(f"conv3x3 {pname} B{B} {Cin}->{Cout} {H}x{W}", rel_err(got, want), tol)
        x = bf(torch.randn(1, 64, 12, 12, generator=g))
        w = bf(torch.randn(64, 64, 3, 3, generator=g) / 24)
        record(f"conv3x3 {pname} pre-relu nobias", rel_err(ops.conv3x3(dev, x.cuda(), w.cuda(), None, True, prec), F.conv2d(F.relu(x), w, None, padding=1)), tol)
        # the last shape is large enough for the 256x256 kernel (its one-division pixel-shuffle epilogue, partial last tile)
        for (B, Cin, H, W, Cout) in [(1, 64, 8, 8, 64), (2, 256, 12, 10, 128), (1, 1024, 24, 24, 256), (2, 256, 97, 88, 128)]:
            x = torch.randn(B, Cin, H, W, generator=g)
            w = torch.randn(Cin, Cout, 2, 2, generator=g) / math.sqrt(Cin)
            b = torch.randn(Cout, generator=g)
            x, w = rnd(x), rnd(w)
            want = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2).float()
            got = ops.deconv2x2(dev, x.cuda(), w.cuda(), b.cuda(), prec)
            record(f"deconv2x2 {pname} B{B} {Cin}->{Cout} {H}x{W}", rel_err(got, want), tol)
    for (B, Cin, H, W, Cout, k, s, p, relu) in [(2, 64, 16, 16, 32, 3, 2, 1, True), (1, 32, 8, 8, 16, 3, 2, 1, True), (2, 8, 6, 6, 1, 6, 1, 0, False),
                                                  (1, 256, 48, 48, 128, 3, 2, 1, True)]:
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(k * k * Cin)
        b = torch.randn(Cout, generator=g)
        want = F.conv2d(x, w, b, stride=s, padding=p)
        if relu:
            want = F.relu(want)
        record(f"conv_direct {Cin}->{Cout} k{k}s{s}p{p}", rel_err(ops.conv2d_direct(dev, x.cuda(), w.cuda(), b.cuda(), s, p, relu), want), 2e-5)


@guarded("split ops")
def check_split_ops(dev):
    """MD_PREC_F16X2 operators: fp32 activations carried as hi + lo half planes (22 bits), weights as exact halves (two MFMA
    terms) or as hi + lo (three). Against fp64 on the UNROUNDED inputs: what is left is the 2^-22 operand error and fp32
    accumulation."""
    g = torch.Generator().manual_seed(21)
    P = 4
    for wexact in (True, False):
        wl = "w16" if wexact else "w32"
        for (M, N, K, tile) in [(300, 256, 128, _lib.TILE_128x128), (300, 256, 128, _lib.TILE_256x256), (577, 3072, 1024, _lib.TILE_AUTO), (300, 256, 128, _lib.TILE_128x64), (300, 132, 192, _lib.TILE_64x64),
                                (1160, 1024, 4096, _lib.TILE_AUTO), (64, 32, 192, _lib.TILE_256x32), (129, 132, 64, _lib.TILE_128x128),
                                (513, 260, 320, _lib.TILE_256x256), (2000, 1024, 1024, _lib.TILE_256x256), (700, 512, 192, _lib.TILE_256x256)]:
            x = torch.randn(M, K, generator=g) * 1.7
            w = torch.randn(N, K, generator=g) / math.sqrt(K)
            w = h16(w) if wexact else w
            b = torch.randn(N, generator=g)
            want = F.linear(x.double(), w.double(), b.double()).float()
            record(f"linear f16x2 {wl} M{M} N{N} K{K} tile{tile}", rel_err(ops.linear(dev, x.cuda(), w.cuda(), b.cuda(), 0, P, tile), want), 3e-6)
        for (M, N, K, act) in [(2000, 1024, 1024, 0), (513, 264, 320, 0), (700, 512, 128, 2), (1300, 256, 192, 1), (513, 260, 320, 0)]:
            x = torch.randn(M, K, generator=g)
            w = torch.randn(N, K, generator=g) / math.sqrt(K)
            w = h16(w) if wexact else w
            b = torch.randn(N, generator=g)
            want = F.linear(x.double(), w.double(), b.double())
            want = F.gelu(want) if act == 2 else (F.relu(want) if act == 1 else want)
            got = ops.linear(dev, x.cuda(), w.cuda(), b.cuda(), act, P, _lib.TILE_256x256, storage_out=True)
            record(f"linear f16x2-out {wl} M{M} N{N} K{K} act{act}", rel_err(got, want.float()), 3e-6)
        for (B, Cin, H, W, Cout) in [(1, 64, 16, 16, 64), (2, 128, 24, 20, 256), (1, 256, 48, 48, 256), (1, 64, 33, 17, 32), (2, 128, 160, 120, 256)]:
            x = torch.randn(B, Cin, H, W, generator=g)
            w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
            w = h16(w) if wexact else w
            b = torch.randn(Cout, generator=g)
            want = F.conv2d(x.double(), w.double(), b.double(), padding=1).float()
            record(f"conv3x3 f16x2 {wl} B{B} {Cin}->{Cout} {H}x{W}", rel_err(ops.conv3x3(dev, x.cuda(), w.cuda(), b.cuda(), False, P), want), 3e-6)
            record(f"conv3x3 f16x2-out {wl} B{B} {Cin}->{Cout} {H}x{W}", rel_err(ops.conv3x3(dev, x.cuda(), w.cuda(), b.cuda(), False, P, storage_out=True), want), 3e-6)
        for (B, Cin, H, W, Cout) in [(1, 64, 8, 8, 64), (2, 256, 12, 10, 128), (1, 1024, 24, 24, 256), (2, 256, 97, 88, 128)]:
            x = torch.randn(B, Cin, H, W, generator=g)
            w = torch.randn(Cin, Cout, 2, 2, generator=g) / math.sqrt(Cin)
            w = h16(w) if wexact else w
            b = torch.randn(Cout, generator=g)
            want = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2).float()
            record(f"deconv2x2 f16x2 {wl} B{B} {Cin}->{Cout} {H}x{W}", rel_err(ops.deconv2x2(dev, x.cuda(), w.cuda(), b.cuda(), P), want), 3e-6)
            record(f"deconv2x2 f16x2-out {wl} B{B} {Cin}->{Cout} {H}x{W}", rel_err(ops.deconv2x2(dev, x.cuda(), w.cuda(), b.cuda(), P, storage_out=True), want), 3e-6)
    # small values: the lo plane is a SUBNORMAL half below |x| = 2^-3 (kept by the conversion and the MFMA: tools/probes)
    x = torch.randn(300, 256, generator=g) * 1e-3
    w = h16(torch.randn(128, 256, generator=g))
    record("linear f16x2 small activations (1e-3)", rel_err(ops.linear(dev, x.cuda(), w.cuda(), None, 0, P), F.linear(x.double(), w.double()).float()), 3e-4)
    for (T, N, heads) in [(2, 65, 4), (3, 577, 2), (1, 577, 16), (2, 64, 1), (1, 130, 3), (1, 1370, 2), (1, 1025, 1), (1, 2000, 3)]:
        qkv = torch.randn(T, N, 3 * heads * 64, generator=g)
        qkv[..., :heads * 64] *= 2.0
        want = attn_ref(qkv, heads, R.identity)
        got = ops.attention(dev, qkv.cuda(), heads, P)
        record(f"attention f16x2 T{T} N{N} h{heads} max", rel_err(got, want), 2e-5)
        record(f"attention f16x2 T{T} N{N} h{heads} mean", mean_rel(got, want), 3e-6)
    # the fall-back to the running-maximum body (see check_attention)
    qkv = torch.randn(1, 577, 3 * 64, generator=g)
    qkv[0, 500, 64:128] = qkv[0, 3, :64] * 6.0
    qkv[0, 570, 64:128] = qkv[0, 3, :64] * 9.0
    qkv[0, 40, :64] *= 8.0
    record("attention f16x2 late keys max", rel_err(ops.attention(dev, qkv.cuda(), 1, P), attn_ref(qkv, 1, R.identity)), 5e-4)
    # the same on a 1370-key sequence
    qkv = torch.randn(1, 1370, 3 * 64, generator=g)
    qkv[0, 1200, 64:128] = qkv[0, 3, :64] * 6.0
    qkv[0, 1330, 64:128] = qkv[0, 3, :64] * 9.0
    qkv[0, 40, :64] *= 8.0
    record("attention f16x2 1370 keys, late keys max", rel_err(ops.attention(dev, qkv.cuda(), 1, P), attn_ref(qkv, 1, R.identity)), 5e-4)


# depth tolerances (max-rel, focal rel, mean-rel) and tap tolerance per precision mode, against the fp32 oracle.
# The reference's own bar for a backend is max-abs 5e-3 / mean-abs 1e-3 / max-rel 5e-3 (example/correctness.rs:887-897).
E2E_TOL = {0: ((8e-2, 5e-3, 8e-3), 3e-2), 1: ((1e-3, 1e-3, 1e-4), 2e-4), 3: ((1.2e-2, 1e-3, 1.2e-3), 4e-3), 4: ((3e-4, 1e-4, 2e-5), 1e-4)}
FOV_TOL = {0: (0.05, 2e-3), 1: (1e-3, 2e-5), 3: (8e-3, 3e-4), 4: (1e-3, 2e-5)}
# Full-size frames have 2.4 M pixels: the MAXIMUM relative error of a rounding-noise process over that many samples is an
# extreme-value statistic (bf16: 4.9e-2 and 1.6e-1 measured for two equally accurate kernels, mean-rel 3.0e-3 both), so the
# reduced-precision modes are held to the 99.9th percentile and the mean, with a loose sanity bound on the maximum.
# (max-rel sanity bound, p99.9 rel, mean-rel)
# MD_PREC_F16X2 is held to the reference's own bar on the MAXIMUM (max-rel 5e-3, example/correctness.rs:887-897) and, below,
# to BASELINE's L_inf < 1e-3
FULL_TOL = {0: (0.3, 5e-2, 8e-3), 1: (1e-3, 1e-3, 1e-4), 3: (6e-2, 6e-3, 1.2e-3), 4: (5e-3, 3e-4, 5e-5)}
FULL_LINF = {1: 1e-3, 4: 1e-3}  # north_star: depth L_inf < 1e-3 against the reference CPU path
# What catches a LOCALISED defect (a wrong halo row of one 3x3 tile, a mis-merged tile border) at full size, where the depth
# maximum is an extreme-value statistic: every debug tap of a reduced-precision mode against the fp32 mode's tap of the same
# frame (the fp32 mode itself is held to the oracle at 1e-3 / L_inf 1e-3 above) -- rms-rel = rms(diff) / rms(tap) and
# max/peak = max|diff| / max|tap|. Bounds = 2 x (rms-rel) and 2.5 x (max/peak) what profiles/r03_stage_errors_*.txt measured
# (bf16: encoder 4.5-5.4e-3 / 5.8e-3, decoder 6.6-8.3e-3 / 7.9e-3, head 8.2-8.6e-3 / 9.8e-3, canonical 3.1e-3 / 6.1e-3; f16 an
# eighth of that; f16x2 1.1-2.7e-6 / 4.0e-6): rounding noise is spread evenly, a defect of O(1) relative size in one row of a
# 768-row map alone is max/peak ~ 1 and rms-rel ~ 3.6e-2.
FULL_TAPS = ([f"encoder_feature_{i}" for i in range(5)] + [f"decoder_fusion_{i}" for i in (4, 3, 2, 1, 0)] +
             ["head_conv0", "head_deconv", "canonical_inverse_depth"])
#           (encoder rms, decoder rms, head rms, canonical rms, max/peak, canonical max/peak)
FULL_TAP_TOL = {0: (1.1e-2, 1.7e-2, 1.8e-2, 6.5e-3, 2.5e-2, 1.6e-2), 3: (1.4e-3, 2.1e-3, 2.2e-3, 8e-4, 3.1e-3, 2e-3), 4: (6e-6, 6e-6, 6e-6, 3e-6, 1.2e-5, 6e-6)}


def tap_stats(got: np.ndarray, ref: np.ndarray):
    """(rms-rel, max / peak) of a tap against its reference; the reductions run on the GPU in fp64 over 64 M-element chunks (the
    largest taps hold 300 M values: fp64 copies on the host would cost gigabytes and most of a minute)."""
    g, r = torch.from_numpy(got).flatten(), torch.from_numpy(ref).flatten()
    se = sr = 0.0
    dmax = rmax = 0.0
    for i in range(0, g.numel(), 1 << 26):
        a, b = g[i:i + (1 << 26)].cuda().double(), r[i:i + (1 << 26)].cuda().double()
        d = (a - b).abs()
        se += float((d * d).sum())
        sr += float((b * b).sum())
        dmax = max(dmax, float(d.max()))
        rmax = max(rmax, float(b.abs().max()))
    return (se / max(sr, 1e-300)) ** 0.5, dmax / (rmax + 1e-30)


def pctl(t: torch.Tensor, q: float) -> float:
    flat = t.flatten()
    k = max(1, min(flat.numel(), int(round(q * flat.numel()))))
    return float(flat.kthvalue(k).values)


def run_e2e(dev, cfg, label, B, hw, precision, taps=True, scheme=Wt.INIT_PARITY, tols=None, emulated=True, timing=True, f16_weights=False, ln_fold=None):
    cfg.precision = precision
    cfg.max_batch = max(B, 1)
    t0 = time.time()
    model = DepthPro.new(dev, cfg, seed=0, init_scheme=scheme)
    if ln_fold is not None:  # md_model_set_option("ln_fold"): 0 off, 1 automatic, 2 on whenever the model can
        model.set_option("ln_fold", ln_fold)
        record(f"{label} ln_fold_active", float(model.query("ln_fold_active")), 1.0 if ln_fold == 2 else 0.0)
    print(f"      model created in {time.time() - t0:.1f}s  workspace={model.query('workspace_bytes') / 1e9:.2f} GB weights={model.query('weight_bytes') / 1e9:.2f} GB", flush=True)
    W = R.weights_to_torch(Wt.generate_depth_pro_weights(cfg, 0, scheme))
    if f16_weights:  # an f16 checkpoint of the seeded weights (mod.rs:206) on both sides
        model.round_weights_to_f16()
        W = {k: R.f16_round(v) for k, v in W.items()}
        scheme = (scheme, "f16")
    if precision == Precision.F16X2:
        record(f"{label} weight_terms", float(model.query("weight_terms")), 2.0 if f16_weights else 3.0, "2 = f16-exact weights, 3 = hi + lo weights")
    # spot-check that the C++ generator produced the same weights
    for n in ("encoder.patch_encoder.blocks.0.attn.qkv.weight", "head.conv_out.weight", "fov.encoder_proj.bias"):
        if n in W:
            got = model.get_tensor(n, W[n].numel())
            record(f"{label} seeded weight {n.split('.')[-3]}.{n.split('.')[-1]}", float(np.abs(got - W[n].numpy().reshape(-1)).max()), 0.0)
    torch.manual_seed(0)
    H, Wd = hw
    img = torch.rand(B, 3, H, Wd)
    x = (img - torch.tensor(R.MEAN).view(1, 3, 1, 1)) / torch.tensor(R.STD).view(1, 3, 1, 1)
    if taps:
        model.enable_taps(True)
    t0 = time.time()
    out = model.infer(x.cuda())
    torch.cuda.synchronize()
    print(f"      first infer {time.time() - t0:.2f}s", flush=True)
    q = {Precision.BF16: R.bf16_round, Precision.F16: R.f16_round, Precision.F16X2: R.f16x2_round}.get(precision, R.identity)
    t0 = time.time()
    def oracle_fp32():
        with torch.no_grad():
            return R.infer(x, W, cfg, q=R.identity, debug=True)
    ref = cached(("depth_pro", cfg_key(cfg), B, tuple(hw), scheme), oracle_fp32)  # same seeds -> same x, W for every precision
    print(f"      oracle fp32 {time.time() - t0:.1f}s", flush=True)
    tol, ttol = E2E_TOL[precision]
    tol = tols or tol
    ftol = FOV_TOL[precision]
    d, rd = out.depth.cpu(), ref["depth"]
    err = (d - rd).abs()
    relmax = (err / rd.abs()).max().item()
    relmean = (err / rd.abs()).mean().item()
    record(f"{label} depth max-rel vs fp32 oracle", relmax, tol[0], f"mean-rel={relmean:.2e} L_inf={err.max().item():.2e} mean-abs={err.mean().item():.2e} depth in [{rd.min():.3f},{rd.max():.3f}]")
    record(f"{label} depth mean-rel vs fp32 oracle", relmean, tol[2])
    record(f"{label} fovx_deg abs", (out.fovx_deg.cpu() - ref['fovx_deg']).abs().max().item(), ftol[0], f"fov={ref['fovx_deg'].tolist()}")
    record(f"{label} focallength rel", rel_err(out.focallength_px, ref["focallength_px"]), tol[1])
    record(f"{label} fovy_rad abs", (out.fovy_rad.cpu() - ref['fovy_rad']).abs().max().item(), ftol[1])
    if precision != Precision.F32 and emulated:
        with torch.no_grad():
            refq = R.infer(x, W, cfg, q=q, debug=True)
        rq = refq["depth"]
        record(f"{label} depth max-rel vs operand-rounding oracle", ((d - rq).abs() / rq.abs()).max().item(), tol[0],
               f"mean-rel={((d - rq).abs() / rq.abs()).mean().item():.2e}")
    else:
        refq = ref
    if taps:
        dbg = refq["debug"]
        names = {f"encoder_feature_{i}": dbg["encoder"]["features"][i] for i in range(5)}
        names.update({f"decoder_fusion_{i}": dbg["fusions"][i] for i in range(5)})
        names.update(decoder_lowres_feature=dbg["decoder_lowres"], decoder_feature=dbg["decoder_features"],
                     head_conv0=dbg["head"]["conv0"], head_deconv=dbg["head"]["deconv"], canonical_inverse_depth=dbg["canonical"])
        for n, t in names.items():
            try:
                got = torch.from_numpy(model.read_tap(n))
                record(f"{label} tap {n}", rel_err(got, t), ttol, f"shape={tuple(got.shape)}")
            except Exception as e:  # noqa: BLE001
                record(f"{label} tap {n}", float("nan"), ttol, f"EXC {e}")
    if timing:
        model.enable_taps(False)
        model.enable_timing(True)
        model.infer(x.cuda())
        tm = model.read_timing()
        tot = sum(v[0] for v in tm.values())
        print(f"      kernel time {tot:.2f} ms/batch: " + ", ".join(f"{k}={v[0]:.2f}ms/{v[1]}" for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0])[:12]), flush=True)
        model.enable_timing(False)
    model.destroy()
    return out, ref


def full_size_key(frame, f16_weights, scheme, want_q):
    return ("full", frame, bool(f16_weights), scheme, bool(want_q))


def full_size_reference(frame="seeded", f16_weights=False, scheme=Wt.INIT_PARITY, want_q=False):
    """The CPU-oracle side of `run_full_size`, in this process (a registered frame comes from a child process instead,
    `prefetch_processes`): the input frame, the fp32 oracle's result and (seeded frame, `want_q`) the result of the oracle that
    rounds every MFMA operand to bf16 where the engine does."""
    import oracle_frames
    parts = {"fp32": oracle_frames.full_size_frame(frame, f16_weights, scheme, "fp32")}
    if want_q:
        parts["q"] = oracle_frames.full_size_frame(frame, f16_weights, scheme, "q")
    return _assemble_full(parts)


def run_full_size(dev, precisions=(Precision.F32, Precision.F16, Precision.BF16), f16_weights=False, frame="seeded", scheme=Wt.INIT_PARITY, emulated=True):
    """The default DepthProConfig at full size, every precision mode in `precisions` against ONE fp32 CPU-oracle frame (the
    oracle costs ~19 TFLOP: about a minute on the GPU box's host cores). `frame`:
      "seeded"   BASELINE config 3-(ii): torch.manual_seed(0) U[0,1) image [1,3,1536,1536], normalised;
      "zeros"    BASELINE config 1: zeros [1,3,1536,1536] (example/inference.rs plumbing; with scheme = INIT_REFERENCE the
                 `DepthPro::new` initialisation of bench/inference.rs:25-27);
      "test_jpg" BASELINE config 3-(iii): the reference's assets/image/test.jpg (540 x 360, decoded pixels committed as
                 tests/golden/test_jpg_rgb.npy) through `infer_from_rgb` (src/inference.rs:128-137) -- both resizes of
                 DepthPro::infer (mod.rs:317-354) run.
    f16_weights: the weights rounded to f16 on both sides, as the reference's checkpoint records hold them (mod.rs:206)."""
    want_q = frame == "seeded" and Precision.BF16 in precisions and emulated
    fr = cached(full_size_key(frame, f16_weights, scheme, want_q), lambda: full_size_reference(frame, f16_weights, scheme, want_q))
    x, rgb, ref, refq = fr["x"], fr["rgb"], fr["ref"], fr["refq"]
    rd = ref["depth"]
    tag = ("" if frame == "seeded" else f"/{frame}") + ("/f16w" if f16_weights else "") + ("/refinit" if scheme == Wt.INIT_REFERENCE else "")
    want_taps = frame == "seeded" and Precision.F32 in precisions and any(int(p) in FULL_TAP_TOL for p in precisions)
    ref_taps = None
    for precision in sorted(precisions, key=lambda p: 0 if p == Precision.F32 else 1):  # the fp32 mode first: its taps are the others' reference
        c = DepthProConfig()
        c.precision = precision
        c.max_batch = 1
        model = DepthPro.new(dev, c, seed=0, init_scheme=scheme)
        if f16_weights:
            model.round_weights_to_f16()
        if want_taps:
            model.enable_taps(True)
        out = model.infer_from_rgb(rgb.tobytes(), rgb.shape[1], rgb.shape[0]) if rgb is not None else model.infer(x.cuda())
        torch.cuda.synchronize()
        if want_taps:
            taps = {n: model.read_tap(n) for n in FULL_TAPS}
            if precision == Precision.F32:
                ref_taps = taps
            elif ref_taps is not None and int(precision) in FULL_TAP_TOL:
                tt = FULL_TAP_TOL[int(precision)]
                for n in FULL_TAPS:
                    rms, mp = tap_stats(taps[n], ref_taps[n])
                    rt = tt[3] if n.startswith("canonical") else (tt[0] if n.startswith("encoder") else (tt[1] if n.startswith("decoder") else tt[2]))
                    record(f"full/{PNAME[int(precision)]} tap {n} rms-rel vs fp32 mode", rms, rt)
                    record(f"full/{PNAME[int(precision)]} tap {n} max/peak vs fp32 mode", mp, tt[5] if n.startswith("canonical") else tt[4])
            del taps
        tol, _ = E2E_TOL[precision]
        ftol = FOV_TOL[precision]
        d = out.depth.cpu()
        err = (d - rd).abs()
        rel = err / rd.abs()
        label = f"full/{PNAME[int(precision)]}" + tag
        ft = FULL_TOL[int(precision)]
        record(f"{label} output shape and finiteness", 0.0 if (tuple(d.shape) == tuple(rd.shape) and bool(torch.isfinite(d).all())) else 1.0, 0.0, f"shape={tuple(d.shape)}")
        record(f"{label} depth max-rel vs fp32 oracle", rel.max().item(), ft[0],
               f"p99.9-rel={pctl(rel, 0.999):.2e} mean-rel={rel.mean().item():.2e} L_inf={err.max().item():.2e} mean-abs={err.mean().item():.2e} depth in [{rd.min():.3g},{rd.max():.3g}]")
        if int(precision) in FULL_LINF and float(rd.max()) < 100.0:  # an absolute bound needs depths of ordinary size
            record(f"{label} depth L_inf vs fp32 oracle", err.max().item(), FULL_LINF[int(precision)])
        record(f"{label} depth p99.9 rel vs fp32 oracle", pctl(rel, 0.999), ft[1])
        record(f"{label} depth mean-rel vs fp32 oracle", rel.mean().item(), ft[2])
        if precision == Precision.BF16 and refq is not None:
            relq = (d - refq).abs() / refq.abs()
            record(f"{label} depth max-rel vs bf16-operand-rounding oracle", relq.max().item(), ft[0], f"p99.9-rel={pctl(relq, 0.999):.2e} mean-rel={relq.mean().item():.2e}")
            record(f"{label} depth p99.9 rel vs bf16-operand-rounding oracle", pctl(relq, 0.999), ft[1])
            record(f"{label} depth mean-rel vs bf16-operand-rounding oracle", relq.mean().item(), ft[2])
        if rgb is None:  # infer_from_rgb returns DepthPrediction {depth, focallength_px, fovy_rad} (src/inference.rs:10-14)
            record(f"{label} fovx_deg abs", (out.fovx_deg.cpu() - ref['fovx_deg']).abs().max().item(), ftol[0], f"fov={ref['fovx_deg'].tolist()}")
        else:
            record(f"{label} fovy_rad abs", (out.fovy_rad.cpu() - ref['fovy_rad']).abs().max().item(), ftol[1])
        # f_px = 0.5 W / tan(0.5 fovx): its relative error is fovx's relative error, so the bound cannot be tighter than the fov bound
        # over the fov itself (config 1 -- zeros, reference initialisation -- has fovx = 0.057 deg: 3.9e-4 deg of bf16 error, inside the
        # 5e-2 deg fov bound, is 6.9e-3 of the focal length)
        fov_ref = float(ref["fovx_deg"].abs().min())
        record(f"{label} focallength rel", rel_err(out.focallength_px, ref["focallength_px"]), max(tol[1], ftol[0] / max(fov_ref, 1e-6) if rgb is None else tol[1]))
        model.destroy()


def run_shard_batch(dev, B=8):
    """BASELINE config 4's per-GPU shard: B = 8 images at 1536^2 in one infer; images 0 and B-1 must be bit-equal to the
    same images run alone (B is a pure batch dimension: encoder.rs:216-225,249-255)."""
    cfg = DepthProConfig()
    cfg.max_batch = B
    model = DepthPro.new(dev, cfg, seed=0, init_scheme=Wt.INIT_PARITY)
    S = cfg.img_size()
    g = torch.Generator().manual_seed(11)
    x = ((torch.rand(B, 3, S, S, generator=g) - torch.tensor(R.MEAN).view(1, 3, 1, 1)) / torch.tensor(R.STD).view(1, 3, 1, 1)).cuda()
    ob = model.infer(x)
    torch.cuda.synchronize()
    record(f"shard B{B} finite positive depth", 0.0 if bool((torch.isfinite(ob.depth) & (ob.depth > 0)).all()) else 1.0, 0.0)
    for i in (0, B - 1):
        o1 = model.infer(x[i:i + 1].contiguous())
        same = torch.equal(o1.depth[0], ob.depth[i]) and torch.equal(o1.fovx_deg[0], ob.fovx_deg[i]) and torch.equal(o1.focallength_px[0], ob.focallength_px[i])
        record(f"shard B{B} image {i} bit-equal to its B=1 run", 0.0 if same else float((o1.depth[0] - ob.depth[i]).abs().max()), 0.0)
    record(f"shard B{B} images differ from each other", 0.0 if not torch.equal(ob.depth[0], ob.depth[B - 1]) else 1.0, 0.0)
    model.destroy()


# the reference's own Depth-Anything-v3 acceptance bar (example/correctness.rs:1109-1111): max-abs, mean-abs, max-rel of the depth
DA3_REF_BAR = (5e-3, 1e-3, 1e-2)


def run_da3(dev, cfg, label, B, precision, scheme=Wt.INIT_PARITY, taps=False, f16_weights=False, x=None):
    """`f16_weights`: the seeded weights rounded to f16 on both sides, as the reference's DA3 records hold them
    (NamedMpkFileRecorder<HalfPrecisionSettings>, example/correctness.rs:977); `x`: an input other than the seeded normal one."""
    from burn_depth_amd.depth_anything3 import DepthAnything3
    from oracle import da3_ref as D3
    cfg.precision = precision
    cfg.max_batch = B
    t0 = time.time()
    model = DepthAnything3.new(dev, cfg, seed=0, init_scheme=scheme)
    print(f"      da3 model created in {time.time() - t0:.1f}s  workspace={model.query('workspace_bytes') / 1e9:.2f} GB", flush=True)
    W = R.weights_to_torch(Wt.generate_da3_weights(cfg, 0, scheme))
    if f16_weights:
        model.round_weights_to_f16()
        W = {k: R.f16_round(v) for k, v in W.items()}
        scheme = (scheme, "f16")
    if precision == Precision.F16X2:
        record(f"{label} weight_terms", float(model.query("weight_terms")), 2.0 if f16_weights else 3.0, "2 = f16-exact weights, 3 = hi + lo weights")
    hp = "head_dual" if cfg.dual_head else "head_mono"
    for n in ("backbone.pretrained.blocks.0.attn.qkv.weight", f"{hp}.resize_layers.0.conv_t.weight", f"{hp}.scratch.output_conv2.conv2.weight"):
        got = model.get_tensor(n, W[n].numel())
        record(f"{label} seeded weight {n.split('.')[-3]}.{n.split('.')[-1]}", float(np.abs(got - W[n].numpy().reshape(-1)).max()), 0.0)
    torch.manual_seed(1)
    S = cfg.image_size
    x_key = "seeded" if x is None else ("given", tuple(x.shape), float(x.double().sum()))
    if x is None:
        x = torch.randn(B, 3, S, cfg.image_width or S)
    if taps:
        model.enable_taps(True)
    out = model.infer(x.cuda())
    torch.cuda.synchronize()
    t0 = time.time()
    def oracle_fp32():
        with torch.no_grad():
            return D3.infer(x, W, cfg, debug=taps)
    ref = cached(("da3", cfg_key(cfg), B, scheme, bool(taps), x_key), oracle_fp32)  # same seeds -> same x, W for every precision
    print(f"      da3 oracle fp32 {time.time() - t0:.1f}s", flush=True)
    if taps:  # DepthTrace (depth_anything3/mod.rs:241-246) and the head's stages against the oracle's intermediates
        dbg = ref["debug"]
        names = {f"backbone_tokens_{i}": dbg["hooks"][i] for i in range(4)}
        names.update({f"stage_{i}": dbg["stage_feats"][i] for i in range(4)})
        names.update({f"layer{i + 1}_rn": dbg["rn"][i] for i in range(4)})
        names["head_input"] = dbg["fused"]
        if cfg.dual_head:
            names.update(aux_neck=dbg["aux_neck"], aux_head_input=dbg["aux_head_input"])
        ttol = {Precision.F32: 2e-4, Precision.F16X2: 2e-4, Precision.F16: 4e-3}.get(precision, 3e-2)
        for n, t in names.items():
            try:
                got = torch.from_numpy(model.read_tap(n))
                record(f"{label} tap {n}", rel_err(got.reshape(t.shape), t), ttol, f"shape={tuple(got.shape)}")
            except Exception as e:  # noqa: BLE001
                record(f"{label} tap {n}", float("nan"), ttol, f"EXC {e}")
        model.enable_taps(False)
    d, rd = out.depth.cpu(), ref["depth"]
    rel = (d - rd).abs() / rd.abs()
    if precision == Precision.FP8:
        # e4m3 operands in the four ViT linear layers: compared with the oracle running the SAME quantisation
        # (bf16 operand rounding elsewhere), and reported against the fp32 oracle
        record(f"{label} depth mean-rel vs fp32 oracle (quantisation error)", rel.mean().item(), 6e-2, f"max-rel={rel.max().item():.2e}")
        # the second CPU frame (the oracle with the engine's quantisation) only below 3000 tokens: at 1036^2 it costs 80 s of the
        # GPU suite for bounds no tighter than the fp32 comparison above; the 518^2 tests carry it
        if (cfg.image_size // 14) * ((getattr(cfg, "image_width", 0) or cfg.image_size) // 14) < 3000:
            with torch.no_grad():
                refq = D3.infer(x, W, cfg, q=R.bf16_round, fp8=True)
            rq = refq["depth"]
            relq = (d - rq).abs() / rq.abs()
            record(f"{label} depth max-rel vs fp8-emulating oracle", relq.max().item(), 1.5e-1, f"mean-rel={relq.mean().item():.2e}")
            record(f"{label} depth mean-rel vs fp8-emulating oracle", relq.mean().item(), 1.5e-2)
            record(f"{label} oracle: fp8 emulation vs fp32 mean-rel (informative)", ((rq - rd).abs() / rd.abs()).mean().item(), 6e-2)
        model.enable_timing(True)
        model.infer(x.cuda())
        tm = model.read_timing()
        tot = sum(v[0] for v in tm.values())
        print(f"      da3 kernel time {tot:.2f} ms/batch: " + ", ".join(f"{k}={v[0]:.2f}ms/{v[1]}" for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0])[:10]), flush=True)
        model.destroy()
        return
    tol = {Precision.BF16: (8e-2, 1e-2), Precision.F16: (1.2e-2, 1.5e-3)}.get(precision, (1e-3, 1e-4))
    record(f"{label} depth max-rel vs fp32 oracle", rel.max().item(), tol[0], f"mean-rel={rel.mean().item():.2e} L_inf={(d - rd).abs().max().item():.2e} depth in [{rd.min():.3f},{rd.max():.3f}]")
    record(f"{label} depth mean-rel vs fp32 oracle", rel.mean().item(), tol[1])
    if precision in (Precision.F32, Precision.F16X2):  # the accurate modes are held to the reference's own DA3 bar as well
        err = (d - rd).abs()
        record(f"{label} depth max-abs (reference bar, correctness.rs:1109)", err.max().item(), DA3_REF_BAR[0])
        record(f"{label} depth mean-abs (reference bar, correctness.rs:1110)", err.mean().item(), DA3_REF_BAR[1])
        record(f"{label} depth max-rel (reference bar, correctness.rs:1111)", rel.max().item(), DA3_REF_BAR[2])
    if cfg.dual_head:  # every other field of DepthAnything3Inference (mod.rs:231-239)
        bf = precision == Precision.BF16
        k = {Precision.BF16: 1.0, Precision.F16: 0.15}.get(precision, 0.0)  # f16: 3 more mantissa bits than bf16; f32 / f16x2: the tight bounds
        t_rel, t_abs, t_pose = (8e-2 * k or 1e-3), (8e-2 * k or 1e-3), (3e-2 * k or 2e-4)
        for name, rt, at in (("depth_confidence", t_rel, 0.0), ("aux_confidence", t_rel, 0.0),
                             ("aux", 0.0, t_abs), ("pose_encoding", 0.0, t_pose),
                             ("extrinsics", 0.0, t_pose)):
            g, w = getattr(out, name).cpu(), ref[name]
            assert g.shape == w.shape, (name, g.shape, w.shape)
            if rt:
                record(f"{label} {name} max-rel", ((g - w).abs() / w.abs()).max().item(), rt, f"range [{w.min():.3f},{w.max():.3f}]")
            else:
                record(f"{label} {name} max-abs", (g - w).abs().max().item(), at, f"range [{w.min():.3f},{w.max():.3f}]")
        g, w = out.intrinsics.cpu(), ref["intrinsics"]
        fin = torch.isfinite(w)
        record(f"{label} intrinsics rel (finite entries)", rel_err(g[fin], w[fin]), t_pose, f"fx={w[0, 0, 0, 0]:.2f} fy={w[0, 0, 1, 1]:.2f}")
        record(f"{label} intrinsics non-finite pattern", float((torch.isfinite(g) != fin).sum().item()), 0.0)
    model.enable_timing(True)
    model.infer(x.cuda())
    tm = model.read_timing()
    tot = sum(v[0] for v in tm.values())
    print(f"      da3 kernel time {tot:.2f} ms/batch: " + ", ".join(f"{k}={v[0]:.2f}ms/{v[1]}" for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0])[:12]), flush=True)
    model.enable_timing(False)
    model.destroy()


def prefetch_da3(cfg, B, scheme=Wt.INIT_PARITY):
    """Registers the fp32 oracle frame `run_da3(dev, cfg, .., B, ..)` (seeded input, no taps, fp32-valued weights) will ask for."""
    register_frame(("da3", cfg_key(cfg), B, scheme, False, "seeded"), {"fp32": f"da3:{cfg.variant}:{cfg.image_size}:{B}:{int(scheme)}"})


def prefetch_full_size(frame="seeded", f16_weights=False, scheme=Wt.INIT_PARITY, want_q=False):
    w = "f16w" if f16_weights else "f32w"
    jobs = {"fp32": f"full:{frame}:{w}:{int(scheme)}:fp32"}
    if want_q:
        jobs["q"] = f"full:{frame}:{w}:{int(scheme)}:q"
    register_frame(full_size_key(frame, f16_weights, scheme, want_q), jobs)


def camera_inputs(B, V, H, W, seed=3):
    """Seeded world-to-camera extrinsics [B, V, 3, 4] whose rotations reach all four branches of the reference's matrix_to_quaternion
    (camera.rs:418-514: trace > 0, and near-pi turns about x, y, z) and intrinsics [B, V, 3, 3] with fx, fy either side of W/2, H/2."""
    import math
    g = np.random.default_rng(seed)

    def rot(axis, ang):
        a = np.asarray(axis, float)
        a /= np.linalg.norm(a)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        return np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * K @ K
    axes = [(1, 2, 3), (1, .1, .1), (.1, 1, .1), (.1, .1, 1)]
    E, I = np.zeros((B, V, 3, 4), np.float32), np.zeros((B, V, 3, 3), np.float32)
    for b in range(B):
        for v in range(V):
            i = (b * V + v) % 4
            ax = np.asarray(axes[i]) + 0.05 * g.standard_normal(3)
            E[b, v, :, :3] = rot(ax, 0.3 + 0.4 * g.random() if i == 0 else 2.9 + 0.2 * g.random())
            E[b, v, :, 3] = g.uniform(-0.5, 0.5, 3)
            fx, fy = W * (0.35 + 0.5 * g.random()), H * (0.35 + 0.5 * g.random())
            I[b, v] = [[fx, 0, W / 2], [0, fy, H / 2], [0, 0, 1]]
    return torch.from_numpy(E), torch.from_numpy(I)


def run_da3_with_camera(dev, cfg, label, B, V, precision, scheme=Wt.INIT_PARITY, host_inputs=False):
    """`DepthAnything3::infer_with_camera` (mod.rs:301-309): the camera encoder's token against the oracle's, then every output of
    the conditioned inference; the conditioning must move the depth away from the plain `infer` result."""
    from burn_depth_amd.depth_anything3 import DepthAnything3
    from oracle import da3_ref as D3
    cfg.precision = precision
    cfg.max_batch = B
    model = DepthAnything3.new(dev, cfg, seed=0, init_scheme=scheme)
    W = R.weights_to_torch(Wt.generate_da3_weights(cfg, 0, scheme))
    torch.manual_seed(1)
    S, Sw = cfg.image_size, cfg.image_width or cfg.image_size
    x = torch.randn(B, 3, S, Sw)
    E, I = camera_inputs(B, V, S, Sw)
    with torch.no_grad():
        ref = D3.infer(x, W, cfg, debug=True, extrinsics=E, intrinsics=I)
        plain = cached(("da3", cfg_key(cfg), B, scheme, False, "seeded"), lambda: D3.infer(x, W, cfg))
    model.enable_taps(True)
    xin = x if host_inputs else x.cuda()
    out = model.infer_with_camera(xin, E, I)
    torch.cuda.synchronize()
    if cfg.camera_encoder:
        tok = torch.from_numpy(model.read_tap("camera_token"))
        record(f"{label} camera_token vs oracle (rel)", rel_err(tok.reshape(ref["debug"]["camera_token"].shape), ref["debug"]["camera_token"]), 2e-5,
               f"|token|max={ref['debug']['camera_token'].abs().max():.3f}")
    model.enable_taps(False)
    d, rd = out.depth.cpu(), ref["depth"]
    rel = (d - rd).abs() / rd.abs()
    tol = {Precision.BF16: (8e-2, 1e-2), Precision.F16: (1.2e-2, 1.5e-3)}.get(precision, (1e-3, 1e-4))
    record(f"{label} depth max-rel vs fp32 oracle", rel.max().item(), tol[0], f"mean-rel={rel.mean().item():.2e}")
    record(f"{label} depth mean-rel vs fp32 oracle", rel.mean().item(), tol[1])
    moved = ((rd - plain["depth"]).abs() / plain["depth"].abs()).mean().item()
    if cfg.camera_encoder:
        got_moved = ((d - plain["depth"]).abs() / plain["depth"].abs()).mean().item()
        record(f"{label} conditioning moves the depth (mean-rel vs plain infer, oracle {moved:.2e})", -got_moved, -0.25 * moved)
        k = {Precision.BF16: 1.0, Precision.F16: 0.15}.get(precision, 0.0)
        for name, at in (("pose_encoding", 3e-2 * k or 2e-4), ("extrinsics", 3e-2 * k or 2e-4), ("aux", 8e-2 * k or 1e-3)):
            g, w = getattr(out, name).cpu(), ref[name]
            record(f"{label} {name} max-abs", (g - w).abs().max().item(), at)
    else:  # no camera encoder in this variant: the camera inputs are ignored (mod.rs:522-527)
        record(f"{label} camera inputs ignored without an encoder (oracle)", moved, 0.0)
        record(f"{label} camera inputs ignored without an encoder (engine)", (d - model.infer(x.cuda()).depth.cpu()).abs().max().item(), 0.0)
    model.enable_timing(True)
    model.infer_with_camera(x.cuda(), E.cuda(), I.cuda())
    tm = model.read_timing()
    if "camera_encoder" in tm:
        print(f"      camera_encoder kernel: {tm['camera_encoder'][0] * 1e3:.0f} us of {sum(v[0] for v in tm.values()):.2f} ms (B={B}, views={V})", flush=True)
    model.enable_timing(False)
    model.destroy()


def run_da3_from_tokens(dev, cfg, label, B, precision, lead_row=False, host_inputs=False, scheme=Wt.INIT_PARITY):
    """`DepthAnything3::infer_from_tokens` (mod.rs:389-469; the head-only comparison of example/da3_small_correctness.rs:278-322): the four
    hooks' patch tokens of the ORACLE's backbone go into the engine's head; every head output against the oracle's head on the same
    tokens. `lead_row`: tokens carry one leading row (patch_token_start = 1) that must be skipped."""
    from burn_depth_amd.depth_anything3 import DepthAnything3
    from oracle import da3_ref as D3
    cfg.precision = precision
    cfg.max_batch = B
    model = DepthAnything3.new(dev, cfg, seed=0, init_scheme=scheme)
    W = R.weights_to_torch(Wt.generate_da3_weights(cfg, 0, scheme))
    torch.manual_seed(1)
    S, Sw = cfg.image_size, cfg.image_width or cfg.image_size
    x = torch.randn(B, 3, S, Sw)
    with torch.no_grad():
        full = cached(("da3", cfg_key(cfg), B, scheme, True, "seeded"), lambda: D3.infer(x, W, cfg, debug=True))
        toks = [t.clone() for t in full["debug"]["hooks"]]
        if lead_row:  # a row the head must never read: make it poisonous
            toks = [torch.cat([torch.full((B, 1, t.shape[2]), 1e30), t], 1) for t in toks]
        ref = D3.infer_from_tokens(toks, W, cfg, S, Sw)
    # the oracle's own two paths agree (the head sees the same tokens either way)
    record(f"{label} oracle: from-tokens depth == infer depth", (ref["depth"] - full["depth"]).abs().max().item(), 0.0)
    out = model.infer_from_tokens([t if host_inputs else t.cuda() for t in toks], S, Sw)
    torch.cuda.synchronize()
    d, rd = out.depth.cpu(), ref["depth"]
    rel = (d - rd).abs() / rd.abs()
    tol = {Precision.BF16: (8e-2, 1e-2), Precision.F16: (1.2e-2, 1.5e-3)}.get(precision, (1e-3, 1e-4))
    record(f"{label} depth max-rel vs the oracle's head", rel.max().item(), tol[0], f"mean-rel={rel.mean().item():.2e}")
    record(f"{label} depth mean-rel vs the oracle's head", rel.mean().item(), tol[1])
    if cfg.dual_head:
        k = {Precision.BF16: 1.0, Precision.F16: 0.15}.get(precision, 0.0)
        for name, rt, at in (("depth_confidence", 8e-2 * k or 1e-3, 0.0), ("aux_confidence", 8e-2 * k or 1e-3, 0.0), ("aux", 0.0, 8e-2 * k or 1e-3)):
            g, w = getattr(out, name).cpu(), ref[name]
            if rt:
                record(f"{label} {name} max-rel", ((g - w).abs() / w.abs()).max().item(), rt)
            else:
                record(f"{label} {name} max-abs", (g - w).abs().max().item(), at)
        record(f"{label} no camera prediction (mod.rs:468)", float(out.pose_encoding is not None or out.extrinsics is not None), 0.0)
    # the full path on the engine's own backbone stays what it was (the token staging does not leak into it)
    e2e = model.infer(x.cuda()).depth.cpu()
    relf = (e2e - full["depth"]).abs() / full["depth"].abs()
    record(f"{label} infer() after infer_from_tokens max-rel", relf.max().item(), tol[0])
    model.destroy()


# tolerance of a replayed stage against the fp32 oracle by precision mode: max|diff| / max|ref| per tensor
REPLAY_TOL = {Precision.F32: 1e-4, Precision.F16X2: 1e-4, Precision.F16: 4e-3, Precision.BF16: 3e-2}


def run_decoder_head_replay(dev, cfg, label, B, precision, host_inputs=False, scheme=Wt.INIT_PARITY, f16_weights=False):
    """`DepthPro::decoder_from_features` and `DepthPro::head_debug` (depth_pro/mod.rs:262-307) on caller tensors against the
    oracle's `decoder_forward_debug` / `head_debug` -- the decoder replay of example/correctness.rs:538-560 (PyTorch's encoder
    features into the decoder) and the head replay of :382-390, with seeded features standing in for PyTorch's."""
    cfg.precision = precision
    cfg.max_batch = max(B, 1)
    model = DepthPro.new(dev, cfg, seed=0, init_scheme=scheme)
    W = R.weights_to_torch(Wt.generate_depth_pro_weights(cfg, 0, scheme))
    if f16_weights:
        model.round_weights_to_f16()
        W = {k: R.f16_round(v) for k, v in W.items()}
    tol = REPLAY_TOL[precision]
    shapes = model.decoder_level_shapes()
    record(f"{label} decoder levels", float(len(shapes)), 5.0, f"shapes={shapes}")
    def oracle_side():
        g = torch.Generator().manual_seed(11)
        feats_ = [torch.randn(B, c, s, s, generator=g) * 0.5 for c, s in shapes]
        with torch.no_grad():
            dec = R.decoder_forward_debug(feats_, W)
            head = R.head_debug(dec[0], W)
        return feats_, dec, head
    # one oracle run serves every precision of a configuration (the default configuration's decoder + head are 4.9 TFLOP on the host)
    feats, (rf, rl, rfus), rh = cached(("replay", cfg_key(cfg), B, scheme, bool(f16_weights)), oracle_side)
    del W
    src = feats if host_inputs else [f.cuda() for f in feats]
    of, ol, ofus = model.decoder_from_features(src)
    record(f"{label} decoder_from_features features", rel_err(of, rf), tol, f"shape={tuple(of.shape)}")
    record(f"{label} decoder_from_features lowres", rel_err(ol, rl), tol, f"shape={tuple(ol.shape)}")
    for i in range(5):
        record(f"{label} decoder_from_features fusion_{i}", rel_err(ofus[i], rfus[i]), tol, f"shape={tuple(ofus[i].shape)}")
    record(f"{label} fusion_0 is the feature map", float((ofus[0] != of).sum().item()), 0.0)
    # the head on the ORACLE's decoder feature (what the reference's harness feeds: the decoder output, correctness.rs:382-390)
    hd = model.head_debug(rf if host_inputs else rf.cuda())
    for n in ("conv0", "deconv", "conv1", "relu", "pre_out", "canonical"):
        got = getattr(hd, n)
        record(f"{label} head_debug {n}", rel_err(got, rh[n]), tol, f"shape={tuple(got.shape)}")
    record(f"{label} head_debug relu == max(conv1, 0)", float((hd.relu != hd.conv1.clamp_min(0)).sum().item()), 0.0)
    record(f"{label} head_debug canonical == max(pre_out, 0)", float((hd.canonical != hd.pre_out.clamp_min(0)).sum().item()), 0.0)
    # the un-fused head against the product path's fused head on the engine's own decoder feature
    torch.manual_seed(0)
    S = model.img_size()
    x = (torch.rand(B, 3, S, S) - 0.45) / 0.225
    model.enable_taps(True)
    model.infer(x.cuda())
    dfeat = torch.from_numpy(model.read_tap("decoder_feature"))
    canon = torch.from_numpy(model.read_tap("canonical_inverse_depth"))
    model.enable_taps(False)
    hd2 = model.head_debug(dfeat.cuda())
    record(f"{label} head_debug canonical vs the fused head of infer", rel_err(hd2.canonical, canon), tol, f"canonical max={canon.max().item():.3f}")
    out_again = model.infer(x.cuda())
    record(f"{label} infer still finite after the replays", float((~torch.isfinite(out_again.depth)).sum().item()), 0.0)
    # a fork (own workspace, the root's weights) gives the same bits
    fk = model.fork()
    ff, fl_, ffus = fk.decoder_from_features(src)
    record(f"{label} fork: decoder_from_features bit-equal to the root", float((ff != of).sum().item() + (fl_ != ol).sum().item() + sum((a != b).sum().item() for a, b in zip(ffus, ofus))), 0.0)
    record(f"{label} fork: head_debug bit-equal to the root", float((fk.head_debug(dfeat.cuda()).canonical != hd2.canonical).sum().item()), 0.0)
    fk.destroy()
    model.destroy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-small", action="store_true")
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = Device.default()
    print("device:", torch.cuda.get_device_name(0), "| lib:", _lib.load().md_version().decode(), flush=True)
    only = set(args.only.split(",")) if args.only else None

    def want(k):
        return only is None or k in only
    if want("ops"):
        check_rgb(dev)
        check_resize(dev)
        check_split_merge(dev)
        check_layernorm(dev)
        check_linear(dev)
        check_linear_fp8(dev)
        check_attention(dev)
        check_convs(dev)
        check_storage_epilogues(dev)
    if want("attention") and only is not None:
        check_attention(dev)
    if want("linear") and only is not None:
        check_linear(dev)
        check_linear_fp8(dev)
    if want("tiny"):
        guarded("tiny f32")(run_e2e)(dev, DepthProConfig.tiny_test(), "tiny/f32", 1, (512, 512), Precision.F32)
        guarded("tiny bf16")(run_e2e)(dev, DepthProConfig.tiny_test(), "tiny/bf16", 1, (512, 512), Precision.BF16)
        guarded("tiny bf16 B2 resize")(run_e2e)(dev, DepthProConfig.tiny_test(), "tiny/bf16/B2/360x540", 2, (360, 540), Precision.BF16, taps=False)
        guarded("tiny f32 B2 resize")(run_e2e)(dev, DepthProConfig.tiny_test(), "tiny/f32/B2/360x540", 2, (360, 540), Precision.F32, taps=False)
    if want("small") and not args.skip_small:
        guarded("small bf16")(run_e2e)(dev, DepthProConfig.small_test(), "small/bf16", 1, (512, 512), Precision.BF16)
        guarded("small f32")(run_e2e)(dev, DepthProConfig.small_test(), "small/f32", 1, (512, 512), Precision.F32)
    if want("da3"):
        from burn_depth_amd.config import DepthAnything3Config
        guarded("da3 tiny f32")(run_da3)(dev, DepthAnything3Config.tiny_test(), "da3-tiny/f32", 2, Precision.F32)
        guarded("da3 tiny bf16")(run_da3)(dev, DepthAnything3Config.tiny_test(), "da3-tiny/bf16", 2, Precision.BF16)
        c98 = DepthAnything3Config.tiny_test()
        c98.image_size = 98  # 7x7 grid from a 5x5 pos_embed: exercises the bicubic pos-embed interpolation
        guarded("da3 tiny98 f32")(run_da3)(dev, c98, "da3-tiny98/f32", 1, Precision.F32)
        guarded("da3 tiny-dual f32")(run_da3)(dev, DepthAnything3Config.tiny_dual_test(), "da3-tinydual/f32", 2, Precision.F32)
        # non-square inputs (any multiples of 14, mod.rs:509-520): 70 x 98 and 112 x 84 on the 5 x 5 position table
        cns = DepthAnything3Config.tiny_test()
        cns.image_size, cns.image_width = 70, 98
        guarded("da3 tiny 70x98 f32")(run_da3)(dev, cns, "da3-tiny70x98/f32", 2, Precision.F32)
        cnd = DepthAnything3Config.tiny_dual_test()
        cnd.image_size, cnd.image_width = 112, 84
        guarded("da3 tiny-dual 112x84 f32")(run_da3)(dev, cnd, "da3-tinydual112x84/f32", 1, Precision.F32)
        guarded("da3 tiny-dual 112x84 bf16")(run_da3)(dev, cnd, "da3-tinydual112x84/bf16", 1, Precision.BF16)
        guarded("da3 tiny-dual bf16")(run_da3)(dev, DepthAnything3Config.tiny_dual_test(), "da3-tinydual/bf16", 2, Precision.BF16)
        guarded("da3 tiny fp8")(run_da3)(dev, DepthAnything3Config.tiny_test(), "da3-tiny/fp8", 2, Precision.FP8)
        guarded("da3 tiny-dual fp8")(run_da3)(dev, DepthAnything3Config.tiny_dual_test(), "da3-tinydual/fp8", 2, Precision.FP8)
        if not args.skip_small:
            guarded("da3 large f32")(run_da3)(dev, DepthAnything3Config.metric_large(), "da3-large/f32", 1, Precision.F32)
            guarded("da3 large bf16")(run_da3)(dev, DepthAnything3Config.metric_large(), "da3-large/bf16", 1, Precision.BF16)
            guarded("da3 small f32")(run_da3)(dev, DepthAnything3Config.small(), "da3-small/f32", 1, Precision.F32)
            guarded("da3 small bf16")(run_da3)(dev, DepthAnything3Config.small(), "da3-small/bf16", 2, Precision.BF16)
            guarded("da3 large fp8")(run_da3)(dev, DepthAnything3Config.metric_large(), "da3-large/fp8", 1, Precision.FP8)
            guarded("da3 small fp8")(run_da3)(dev, DepthAnything3Config.small(), "da3-small/fp8", 1, Precision.FP8)
    if want("replay"):
        for pr, nm in ((Precision.F32, "f32"), (Precision.F16X2, "f16x2"), (Precision.BF16, "bf16")):
            guarded(f"replay tiny {nm}")(run_decoder_head_replay)(dev, DepthProConfig.tiny_test(), f"replay-tiny/{nm}", 1, pr, f16_weights=pr == Precision.F16X2)
    if want("f16"):
        guarded("f16 storage")(check_storage_epilogues)(dev, 3)
        guarded("tiny f16")(run_e2e)(dev, DepthProConfig.tiny_test(), "tiny/f16", 1, (512, 512), Precision.F16)
        if not args.skip_small:
            guarded("small f16")(run_e2e)(dev, DepthProConfig.small_test(), "small/f16", 1, (512, 512), Precision.F16)
            from burn_depth_amd.config import DepthAnything3Config
            guarded("da3 small f16")(run_da3)(dev, DepthAnything3Config.small(), "da3-small/f16", 1, Precision.F16)
            guarded("da3 large f16")(run_da3)(dev, DepthAnything3Config.metric_large(), "da3-large/f16", 1, Precision.F16)
    if want("f16x2"):
        check_split_ops(dev)
        guarded("tiny f16x2")(run_e2e)(dev, DepthProConfig.tiny_test(), "tiny/f16x2", 1, (512, 512), Precision.F16X2)
        guarded("tiny f16x2 f16w")(run_e2e)(dev, DepthProConfig.tiny_test(), "tiny/f16x2/f16w", 1, (512, 512), Precision.F16X2, f16_weights=True)
        guarded("tiny f16x2 B2 resize")(run_e2e)(dev, DepthProConfig.tiny_test(), "tiny/f16x2/B2/360x540", 2, (360, 540), Precision.F16X2, taps=False, f16_weights=True)
        from burn_depth_amd.config import DepthAnything3Config
        guarded("da3 tiny f16x2")(run_da3)(dev, DepthAnything3Config.tiny_test(), "da3-tiny/f16x2", 2, Precision.F16X2, taps=True)
        guarded("da3 tiny-dual f16x2 f16w")(run_da3)(dev, DepthAnything3Config.tiny_dual_test(), "da3-tinydual/f16x2/f16w", 2, Precision.F16X2, taps=True, f16_weights=True)
        if not args.skip_small:
            guarded("small f16x2 f16w")(run_e2e)(dev, DepthProConfig.small_test(), "small/f16x2/f16w", 1, (512, 512), Precision.F16X2, f16_weights=True)
            guarded("da3 small f16x2 f16w")(run_da3)(dev, DepthAnything3Config.small(), "da3-small/f16x2/f16w", 1, Precision.F16X2, f16_weights=True)
            guarded("da3 large f16x2 f16w")(run_da3)(dev, DepthAnything3Config.metric_large(), "da3-large/f16x2/f16w", 1, Precision.F16X2, f16_weights=True)
    if args.full or (want("full") and only is not None):  # every mode on an f16 checkpoint of the seeded weights, one oracle frame
        guarded("full size")(run_full_size)(dev, (Precision.F32, Precision.F16X2, Precision.F16, Precision.BF16), f16_weights=True)
    if want("testjpg") and only is not None:
        guarded("test.jpg")(run_full_size)(dev, (Precision.F32, Precision.F16X2), f16_weights=True, frame="test_jpg")
    if want("config1") and only is not None:
        guarded("config 1")(run_full_size)(dev, (Precision.F32, Precision.F16X2, Precision.BF16), frame="zeros", scheme=Wt.INIT_REFERENCE)
    if want("shard") and only is not None:
        guarded("shard")(run_shard_batch)(dev)
    if want("config5") and only is not None:
        from burn_depth_amd.config import DepthAnything3Config
        for prec in (Precision.FP8, Precision.BF16, Precision.F16, Precision.F16X2, Precision.F32):
            c5 = DepthAnything3Config.metric_large()
            c5.image_size = 1036
            guarded("config5")(run_da3)(dev, c5, f"da3-large-1036/{PNAME[int(prec)]}", 1, prec)
    bad = [r for r in RESULTS if not r[3]]
    print(f"\n==== {len(RESULTS) - len(bad)}/{len(RESULTS)} checks within tolerance ====")
    for r in bad:
        print("BAD:", r[0], f"err={r[1]:.3e} tol={r[2]:.1e}", r[4])
    return 0


if __name__ == "__main__":
    sys.exit(main())
