#!/bin/bash
# Where the fc1 tile loop's time outside the main loop goes (gemm256p_kernel, bf16, folded): timing-only builds, EXTRA=-DMD_PABL=bits
# (1: no fold / GELU arithmetic in the epilogue, 2: no stores), one box. Results are WRONG in those builds; only fc1's time is read. Run from the
# repo root on a GPU box: bash tools/probes/fc1_loop_ablation.sh
for v in "-DMD_PABL=1" "-DMD_PABL=2" "-DMD_PABL=3" ""; do
  touch burn_depth_amd/csrc/kernels/gemm_impl.h
  make -C burn_depth_amd/csrc EXTRA="$v" -j16 > /dev/null 2>&1 || { echo "variant '$v': build failed"; continue; }
  echo "== EXTRA='$v'"
  timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 6 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('   bf16:', d['value'], 'frames/s, fc1', k['fc1_gemm']['ms_per_step'], 'ms, qkv', k['qkv_gemm']['ms_per_step'], 'step', d['ms_per_step'])"
done
