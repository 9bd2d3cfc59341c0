#!/bin/bash
# The f16 / f16x2 GELU epilogue (kernels/gemm_impl.h): the packed degree-16 polynomial in the shifted variable (EXTRA=-DMD_GELU_POLY16) against the
# shipped Abramowitz & Stegun 7.1.26 form, on one box. Run from the repo root: bash tools/probes/gelu_ab.sh
for v in "-DMD_GELU_POLY16" ""; do
  touch burn_depth_amd/csrc/kernels/gemm_impl.h
  make -C burn_depth_amd/csrc EXTRA="$v" -j16 > /dev/null 2>&1 || { echo "variant '$v': build failed"; continue; }
  echo "== EXTRA='$v'"
  for p in f16 f16x2; do
    timeout -k 10 300 python3 bench.py --precision $p --no-extras --no-cpu-baseline --steps 6 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('   $p:', d['value'], 'frames/s, fc1', k['fc1_gemm']['ms_per_step'], 'ms, step', d['ms_per_step'])"
  done
done
touch burn_depth_amd/csrc/kernels/gemm_impl.h
make -C burn_depth_amd/csrc -j16 > /dev/null 2>&1
