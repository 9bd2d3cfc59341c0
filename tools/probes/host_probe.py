"""What the GPU box's host gives a process: cores (affinity, cgroup quota), memory, and how torch's CPU matmul / conv scale with
the thread count (sizes of the oracle's hot loops). Used to size tools/gpu_diag.py's child-process oracle frames."""
import os
import time

import torch

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try:
        print(p, open(p).read().strip())
    except OSError as e:
        print(p, "-", e.__class__.__name__)
for line in open("/proc/meminfo"):
    if line.startswith(("MemTotal", "MemAvailable")):
        print(line.strip())
print("torch threads", torch.get_num_threads())
a = torch.randn(8 * 577, 1024)
w = torch.randn(4096, 1024)
x = torch.randn(1, 256, 768, 768)
k = torch.randn(256, 256, 3, 3)
for n in (64, 32, 16, 8):
    torch.set_num_threads(n)
    torch.nn.functional.linear(a, w)
    t = time.time()
    for _ in range(5):
        torch.nn.functional.linear(a, w)
    dt = (time.time() - t) / 5
    torch.nn.functional.conv2d(x, k, padding=1)
    t = time.time()
    torch.nn.functional.conv2d(x, k, padding=1)
    dc = time.time() - t
    print(f"threads {n:3d}: linear {2 * a.shape[0] * 1024 * 4096 / dt / 1e12:.2f} TFLOP/s, conv3x3 {2 * 768 * 768 * 256 * 256 * 9 / dc / 1e12:.2f} TFLOP/s", flush=True)
