"""Soak of the assembly attention kernel at the headline's launch size (296 sequences x 16 heads x 577 tokens): ITER launches on fresh
random operands, every output element against the HIP kernel's on the same operands (the two differ by the rounding of the row sums:
at most one bf16 ulp of the row's largest output) and the first launch against the fp64 reference. A wait-state or counted-wait mistake
shows as wrong values on SOME waves of SOME launches (cdna_hip_programming.md section 5.7): this looks at all of them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from burn_depth_amd import _lib, ops  # noqa: E402
from burn_depth_amd.depth_pro import Device  # noqa: E402


def main(iters=30, T=296, heads=16):
    dev = Device(0)
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(7)
    worst = 0.0
    for it in range(iters):
        qkv = torch.randn(T, 577, 3 * heads * 64, generator=g, device="cuda")
        qkv[..., :heads * 64] *= 1.0 + (it % 4)        # logits of a few units up to ~20
        lib.md_debug_attention_asm(1)
        a = ops.attention(dev, qkv, heads, 0)
        lib.md_debug_attention_asm(0)
        h = ops.attention(dev, qkv, heads, 0)
        lib.md_debug_attention_asm(1)
        assert bool(torch.isfinite(a).all()), f"launch {it}: non-finite output"
        peak = h.abs().amax(dim=-1, keepdim=True).clamp_min(1e-6)
        err = ((a - h).abs() / peak).max().item()
        worst = max(worst, err)
        assert err <= 1.6e-2, f"launch {it}: assembly and HIP kernel differ by {err:.3e} of a row's peak"
        if it == 0:
            import gpu_diag as D
            want = D.attn_ref(qkv[:4].cpu(), heads, D.bf)
            print(f"launch 0 against the fp64 reference (4 sequences): max {D.rel_err(a[:4], want):.3e}, mean {D.mean_rel(a[:4], want):.3e}", flush=True)
        if it % 10 == 9:
            print(f"{it + 1} launches, worst assembly - HIP difference {worst:.3e} of a row's peak", flush=True)
    print(f"soak ok: {iters} launches x {T * heads} units, worst difference {worst:.3e}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
