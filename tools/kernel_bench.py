"""Micro-benchmark of the MFMA kernels on the Depth Pro shapes: interleaved rounds in ONE process
(HIP events on the launch stream inside md_bench_gemm / md_bench_attention), random operands."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from burn_depth_amd import _lib  # noqa: E402
from burn_depth_amd.depth_pro import Device  # noqa: E402


def main():
    dev = Device(0)
    lib = _lib.load()
    rows = 37 * 580
    gemms = [("qkv", 0, rows, 3072, 1024, 0, 0), ("proj", 0, rows, 1024, 1024, 0, 0), ("fc1", 0, rows, 4096, 1024, 0, 0),
             ("fc2", 0, rows, 1024, 4096, 0, 0), ("sq8k", 0, 8192, 8192, 8192, 0, 0),
             ("conv768_256", 1, 768 * 768, 256, 256, 768, 768), ("conv384_256", 1, 384 * 384, 256, 256, 384, 384)]
    big = 37 * 580 * 4
    gemms += [("qkvB4", 0, big, 3072, 1024, 0, 0), ("fc1B4", 0, big, 4096, 1024, 0, 0), ("projB4", 0, big, 1024, 1024, 0, 0),
              ("fc2B4", 0, big, 1024, 4096, 0, 0)]
    tiles = [(0, "256pp16"), (0 + 1024, "256pp16_gelu"), (0 + 8 * 256, "pixshuf"), (0 + (8 + 32) * 256, "pixshuf_stamps"), (0 + 16 * 256, "256pp16_rmw"), (0 + (16 + 32) * 256, "rmw_stamps"), (0 + 32 * 256, "256pp16_stamps"), (0 + (32 + 64) * 256, "stamps_nostore"), (0 + (32 + 128) * 256, "stamps_nostage"), (0 + (32 + 64 + 128) * 256, "stamps_nostore_nostage"), (0 + (32 + 4) * 256, "256pp16_gelu_stamps"), (1, "128v1"), (0 + 256, "256pp16_noloads"), (0 + 512, "256pp_freezek"), (0 + 512 + 256, "256pp16_freezek_noloads")]
    gemms += [("hd_k128", 0, 4 * 768 * 768, 512, 128, 0, 0), ("hd_k256", 0, 4 * 768 * 768, 512, 256, 0, 0), ("hd_k512", 0, 4 * 768 * 768, 512, 512, 0, 0)]
    gemms += [("dense768_2304", 0, 768 * 768, 256, 2304, 0, 0)]
    gemms += [("headdeconv", 0, 4 * 768 * 768, 256, 128, 4 * 768, 768), ("decdeconv", 0, 4 * 384 * 384, 1024, 256, 4 * 384, 384)]
    n3 = 74 * 74 + 1  # Depth-Anything-v3 at 1036^2 (BASELINE config 5): one sequence of 5477 tokens
    gemms += [("da3_qkv", 0, n3, 3072, 1024, 0, 0), ("da3_proj", 0, n3, 1024, 1024, 0, 0), ("da3_fc1", 0, n3, 4096, 1024, 0, 0),
              ("da3_fc2", 0, n3, 1024, 4096, 0, 0)]
    ns = 37 * 37 + 1  # Depth-Anything-v3 at 518^2 (BASELINE config 2: small, D = 384): one sequence of 1370 tokens
    gemms += [("s_qkv", 0, ns, 1152, 384, 0, 0), ("s_proj", 0, ns, 384, 384, 0, 0), ("s_fc1", 0, ns, 1536, 384, 0, 0),
              ("s_fc2", 0, ns, 384, 1536, 0, 0), ("s_conv148", 1, 148 * 148, 64, 64, 148, 148), ("s_conv74", 1, 74 * 74, 64, 64, 74, 74)]
    # wave-quantisation study (Depth Pro at B = 1: proj / fc2 have 336 tiles of 256^2 = 1.31 rounds): one full round of 256^2
    # tiles against the remaining rows on 128^2 tiles
    gemms += [("full_proj", 0, 16384, 1024, 1024, 0, 0), ("rem_proj", 0, 5076, 1024, 1024, 0, 0), ("full_fc2", 0, 16384, 1024, 4096, 0, 0),
              ("rem_fc2", 0, 5076, 1024, 4096, 0, 0)]
    tiles += [(1 + 16 * 256, "128v1_rmw"), (1 + 1024, "128v1_gelu"), (3, "128x64"), (4, "64x64")]
    gemms += [("s_conv296", 1, 296 * 296, 64, 64, 296, 296), ("s_conv37", 1, 37 * 37, 64, 64, 37, 37), ("s_conv296_32", 1, 296 * 296, 32, 64, 296, 296),
              ("l_conv148", 1, 148 * 148, 256, 256, 148, 148), ("l_conv74", 1, 74 * 74, 256, 256, 74, 74), ("l_conv37", 1, 37 * 37, 256, 256, 37, 37),
              ("b_qkv", 0, ns, 2304, 768, 0, 0), ("b_proj", 0, ns, 768, 768, 0, 0), ("b_fc2", 0, ns, 768, 3072, 0, 0),
              ("L_qkv", 0, ns, 3072, 1024, 0, 0), ("L_proj", 0, ns, 1024, 1024, 0, 0), ("L_fc1", 0, ns, 4096, 1024, 0, 0), ("L_fc2", 0, ns, 1024, 4096, 0, 0)]
    prec = int(os.environ.get("PREC", "0"))
    only = os.environ.get("ONLY")
    if only:
        gemms = [g for g in gemms if g[0] in only.split(",")]
    if os.environ.get("TILES"):
        tiles = [t for t in tiles if t[1] in os.environ["TILES"].split(",")]
    for name, mode, M, N, K, a0, a1 in gemms:
        flops = 2.0 * M * N * K * (9 if mode == 1 else 1)
        best = {}
        for rnd in range(3):
            for tile, tn in tiles:
                ms = C.c_float()
                _lib.check(lib.md_bench_gemm(dev.handle, mode, M, N, K, a0, a1, prec, tile, 10, C.byref(ms)))
                best[tn] = min(best.get(tn, 1e9), ms.value)
        print(f"{name:12s} M={M} N={N} K={K}: " + "  ".join(f"{k} {v:.4f} ms {flops / v / 1e9:.0f} TF" for k, v in best.items()), flush=True)
    for (T, N, h) in ([] if only and 'attn' not in only else [(37, 577, 16), (296, 577, 16), (1, 1370, 16), (1, 5477, 16)]):
        ms = C.c_float()
        best = 1e9
        for rnd in range(3):
            _lib.check(lib.md_bench_attention(dev.handle, T, N, h, 10, C.byref(ms)))
            best = min(best, ms.value)
        fl = 4.0 * T * h * N * N * 64
        print(f"attention T={T} N={N} h={h}: {best:.4f} ms {fl / best / 1e9:.0f} TF", flush=True)


if __name__ == "__main__":
    main()
