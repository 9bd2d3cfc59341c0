#!/bin/bash
# Per-kernel register / scratch / occupancy table of one HIP translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/kernel_resources.sh burn_depth_amd/csrc/kernels/gemm_bf16.hip
set -e
src="$1"
dir="$(cd "$(dirname "$0")/../burn_depth_amd/csrc" && pwd)"
extra=""
case "$src" in *attention.hip) extra="-fno-slp-vectorize";; esac  # as in the Makefile
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -I"$dir" -I"$dir/kernels" -c "$src" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|  VGPRs:|AGPRs:|ScratchSize|Occupancy \[" | sed 's/.*remark: *//; s/ *\[-Rpass.*//' |
  awk '/Function Name/ {if (n) print line; n=1; sub(/Function Name: /,""); line=$0; next} {line=line " | " $0} END {print line}'
