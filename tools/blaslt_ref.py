"""Library yardstick for the GEMM shapes of the step: torch.matmul (hipBLASLt / rocBLAS) in bf16 on the same shapes as
tools/kernel_bench.py. Not part of the product path; a number to hold the hand-written tiles against."""
import sys
import time

import torch


def bench(m, n, k, dtype=torch.bfloat16, iters=20):
    a = torch.randn(m, k, device="cuda", dtype=dtype)
    w = torch.randn(n, k, device="cuda", dtype=dtype)
    bias = torch.randn(n, device="cuda", dtype=dtype)
    for _ in range(3):
        torch.nn.functional.linear(a, w, bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.nn.functional.linear(a, w, bias)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * m * n * k / ms / 1e9


def main():
    shapes = [("qkv", 170792, 3072, 1024), ("proj", 170792, 1024, 1024), ("fc1", 170792, 4096, 1024), ("fc2", 170792, 1024, 4096),
              ("sq8k", 8192, 8192, 8192), ("dense768_2304", 589824, 256, 2304)]
    for name, m, n, k in shapes:
        for dt in (torch.bfloat16, torch.float16):
            ms, tf = bench(m, n, k, dt)
            print(f"{name:14s} M={m} N={n} K={k} {str(dt)[6:]:9s} {ms:8.4f} ms {tf:7.1f} TF", flush=True)


if __name__ == "__main__":
    main()
