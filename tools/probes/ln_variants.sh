#!/bin/bash
# LayerNorm kernel forms (kernels/ops.hip), in-model: bench.py --no-extras, `layernorm` family (49 launches per step of 8 frames) and frames/s.
#   default build = the shipped form; EXTRA=-DMD_LN_NO_PREFETCH = the plain row loop
# Run from the repo root on the GPU box: bash tools/probes/ln_variants.sh > gpurun_out/ln_variants.txt
for v in "" "-DMD_LN_NO_PREFETCH"; do
  touch burn_depth_amd/csrc/kernels/ops.hip
  make -C burn_depth_amd/csrc EXTRA="$v" -j16 > /dev/null 2>&1 || { echo "variant '$v': build failed"; continue; }
  echo "== EXTRA='$v'"
  timeout -k 10 300 python3 tools/gpu_diag.py --only ops 2>&1 | grep -E "layernorm" | head -8
  for p in bf16 f16x2; do
    timeout -k 10 300 python3 bench.py --precision $p --no-extras --no-cpu-baseline --steps 6 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('   $p:', d['value'], 'frames/s, layernorm', k['layernorm'], 'step', d['ms_per_step'])"
  done
done
touch burn_depth_amd/csrc/kernels/ops.hip
make -C burn_depth_amd/csrc -j16 > /dev/null 2>&1
