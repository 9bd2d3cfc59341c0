"""A/B of the two bf16 attention forms at Depth Pro's shapes (md_bench_attention_ex with md_debug_attention_asm 0 / 1):
the HIP kernel (4 waves x 32 queries, four workgroups per CU) against the assembly-owned kernel (attn577_gfx950.s)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from burn_depth_amd import _lib  # noqa: E402
from burn_depth_amd.depth_pro import Device  # noqa: E402


def main():
    dev = Device(0)
    lib = _lib.load()
    ms = C.c_float()
    for (T, N, heads) in [(37, 577, 16), (296, 577, 16)]:
        fl = 4.0 * T * heads * N * N * 64
        for scale in (0.7, 4.0):
            row = []
            for form in (0, 1, 0, 1):
                lib.md_debug_attention_asm(form)
                best = 1e9
                for _ in range(3):
                    _lib.check(lib.md_bench_attention_ex(dev.handle, T, N, heads, 0, C.c_float(scale), 20, C.byref(ms)))
                    best = min(best, ms.value)
                row.append(f"{'asm' if form else 'hip'} {best * 1e3:.1f} us {fl / best / 1e9:.0f} TF ({fl / best / 1e9 / 2500:.3f})")
            print(f"T={T} N={N} heads={heads} qk_scale={scale}: " + " | ".join(row), flush=True)
    lib.md_debug_attention_asm(1)


if __name__ == "__main__":
    main()
