#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for cfg in "0 4" "32 4"; do
  set -- $cfg
  echo "== MD_ATTN_PIN=$1 (32 = batched asm LDS reads) MD_ATTN_NW=$2"
  for i in 1 2; do MD_ATTN_PIN=$1 MD_ATTN_NW=$2 timeout -k 10 100 python tools/attn_bench.py 296 577 16 0 0.7 2>&1 | grep -v amdgpu.ids; done
  MD_ATTN_PIN=$1 MD_ATTN_NW=$2 timeout -k 10 100 python tools/attn_bench.py 1 5477 16 0 0.7 2>&1 | grep -v amdgpu.ids
  MD_ATTN_PIN=$1 MD_ATTN_NW=$2 timeout -k 10 100 python tools/attn_bench.py 8 1370 16 0 0.7 2>&1 | grep -v amdgpu.ids
  MD_ATTN_PIN=$1 MD_ATTN_NW=$2 timeout -k 10 300 python - > gpurun_out/r2_attn_check_$1_$2.log 2>&1 <<'PY'
import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import gpu_diag as d
from burn_depth_amd.depth_pro import Device
dev=Device(0)
d.check_attention(dev)
bad=[r for r in d.RESULTS if not r[3]]
print(len(d.RESULTS)-len(bad),"/",len(d.RESULTS))
PY
  grep -E "BAD|/ " gpurun_out/r2_attn_check_$1_$2.log | head -4
done
