#!/bin/bash
# Where the epilogue time of the read-modify-write tile loop (gemm256r_kernel: proj, fc2; bf16, with the LayerNorm fold's producer part) goes:
# timing-only builds, EXTRA=-DMD_RABL=bits (1: no x loads, 2: no x stores, 4: no stores of the next GEMM's operand, 8: no row statistics; 0: only
# the loop-top wait becomes vmcnt(0)), one box. Results are WRONG in those builds; only the proj / fc2 times are read. Run from the repo root on a
# GPU box: bash tools/probes/rmw_loop_ablation.sh
for v in "-DMD_RABL=0" "-DMD_RABL=1" "-DMD_RABL=2" "-DMD_RABL=4" "-DMD_RABL=8" "-DMD_RABL=6" "-DMD_RABL=15" ""; do
  touch burn_depth_amd/csrc/kernels/gemm_impl.h
  make -C burn_depth_amd/csrc EXTRA="$v" -j16 > /dev/null 2>&1 || { echo "variant '$v': build failed"; continue; }
  echo "== EXTRA='$v'"
  timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 6 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('   bf16:', d['value'], 'frames/s, proj', k['proj_gemm']['ms_per_step'], 'fc2', k['fc2_gemm']['ms_per_step'], 'fc1', k['fc1_gemm']['ms_per_step'], 'qkv', k['qkv_gemm']['ms_per_step'], 'step', d['ms_per_step'])"
done
