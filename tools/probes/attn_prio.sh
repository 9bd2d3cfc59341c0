#!/bin/bash
# NOTE: the MD_ATTN_VARIANT code these builds select lives at commit 6e0680d (removed afterwards: every variant measured null or slower).
# Wave-priority placements in the fused attention kernel (MD_ATTN_VARIANT, kernels/attention.hip): builds each variant on the GPU box
# and times the Depth Pro launch (T.N = 296 x 577, 16 heads, bf16 fast body) and DA3's N = 5477. Run from the repo root:
#   bash tools/probes/attn_prio.sh > gpurun_out/attn_prio.txt
for v in 0 1 2 3; do
  touch burn_depth_amd/csrc/kernels/attention.hip
  make -C burn_depth_amd/csrc EXTRA=-DMD_ATTN_VARIANT=$v -j16 > /dev/null 2>&1 || { echo "variant $v: build failed"; continue; }
  echo "== MD_ATTN_VARIANT=$v"
  for i in 1 2 3; do timeout -k 10 120 python3 tools/attn_bench.py 296 577 16 0 0.7; done
  timeout -k 10 120 python3 tools/attn_bench.py 1 5477 16 0 0.7
  timeout -k 10 120 python3 tools/attn_bench.py 296 577 16 4 0.7 2>/dev/null || true
done
touch burn_depth_amd/csrc/kernels/attention.hip
make -C burn_depth_amd/csrc -j16 > /dev/null 2>&1
