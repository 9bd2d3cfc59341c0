#!/bin/bash
# NOTE: the MD_ATTN_VARIANT=4 code lives at commit 6e0680d (removed afterwards: correct, 0.84x).
# The loader-wave form of the bf16 attention kernel (MD_ATTN_VARIANT=4: a fifth wave per workgroup issues every LDS-DMA piece) against the
# shipped kernel (0): operator checks, stand-alone rate, in-model step. Run from the repo root on the GPU box:
#   bash tools/probes/attn_loader.sh > gpurun_out/attn_loader.txt
for v in 0 4; do
  touch burn_depth_amd/csrc/kernels/attention.hip
  make -C burn_depth_amd/csrc EXTRA=-DMD_ATTN_VARIANT=$v -j16 > /dev/null 2>&1 || { echo "variant $v: build failed"; continue; }
  echo "== MD_ATTN_VARIANT=$v"
  timeout -k 10 300 python3 tools/gpu_diag.py --only attention 2>&1 | grep -E "^\[(OK |BAD)\]" | awk '{c[$1]++} END {for (k in c) print "   operator checks", k, c[k]}'
  for i in 1 2 3; do timeout -k 10 120 python3 tools/attn_bench.py 296 577 16 0 0.7 2>/dev/null; done
  timeout -k 10 120 python3 tools/attn_bench.py 1 5477 16 0 0.7 2>/dev/null
  timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 6 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('   in-model:', d['value'], 'frames/s, attention', k['attention'], 'step', d['ms_per_step'])"
done
touch burn_depth_amd/csrc/kernels/attention.hip
make -C burn_depth_amd/csrc -j16 > /dev/null 2>&1
