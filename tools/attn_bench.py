"""Attention kernel micro-benchmark (md_bench_attention_ex): useful TFLOP/s = 4 * T * heads * N^2 * 64 / time.
usage: python tools/attn_bench.py                       Depth Pro shapes at B = 1 / 8, DA3 518^2 / 1036^2; bf16 fast body,
                                                         bf16 running-maximum body, f16
       python tools/attn_bench.py T N heads prec scale [hip|asm]  one configuration (for rocprofv3 --pmc passes, tools/pmc_collect.sh);
                                                         `hip` keeps 577-token bf16 launches off the assembly kernel (md_debug_attention_asm)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from burn_depth_amd import _lib  # noqa: E402
from burn_depth_amd.depth_pro import Device  # noqa: E402


def main():
    dev = Device(0)
    lib = _lib.load()
    ms = C.c_float()
    if len(sys.argv) >= 6:
        T, N, heads, prec = (int(a) for a in sys.argv[1:5])
        if len(sys.argv) >= 7:
            lib.md_debug_attention_asm(0 if sys.argv[6] == "hip" else 1)
        _lib.check(lib.md_bench_attention_ex(dev.handle, T, N, heads, prec, C.c_float(float(sys.argv[5])), 10, C.byref(ms)))
        fl = 4.0 * T * heads * N * N * 64
        print(f"T={T} N={N} heads={heads} prec={prec}: {ms.value:.4f} ms = {fl / ms.value / 1e9:.0f} TFLOP/s", flush=True)
        return
    for (T, N, heads) in [(37, 577, 16), (296, 577, 16), (1, 1370, 6), (1, 1370, 16), (8, 1370, 16), (1, 2738, 16), (1, 5477, 16)]:
        fl = 4.0 * T * heads * N * N * 64
        row = []
        for name, prec, scale in [("bf16 fast", 0, 0.7), ("bf16 safe", 0, 4.0), ("f16", 3, 0.7)]:
            best = 1e9
            for _ in range(3):
                _lib.check(lib.md_bench_attention_ex(dev.handle, T, N, heads, prec, C.c_float(scale), 20, C.byref(ms)))
                best = min(best, ms.value)
            row.append(f"{name}: {best:.4f} ms = {fl / best / 1e9:.0f} TFLOP/s ({fl / best / 1e9 / 2500:.3f})")
        print(f"T={T} N={N} heads={heads}: " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
