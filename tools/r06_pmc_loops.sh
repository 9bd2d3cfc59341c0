#!/bin/bash
# SQ counters of the tile-loop kernels (gemm256p_kernel: fc1 form; gemm256r_kernel: proj form) on the shapes of tools/r06_pmc_fc1.sh (B = 4), the same
# three rocprofv3 --pmc passes: the counters beside the kernels that ship at the end of round 6.
set -o pipefail
export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
export ONLY=fc1B4 TILES=256pp16_gelu
bash tools/pmc_collect.sh gemm256p_kernel gpurun_out/r06_pmc_fc1_loop.json -- python3 $GRAFT_REPO_ROOT/tools/kernel_bench.py > gpurun_out/r06_pmc_fc1_loop.log 2>&1 || { tail -5 gpurun_out/r06_pmc_fc1_loop.log; exit 1; }
cd $GRAFT_REPO_ROOT
export ONLY=projB4 TILES=256pp16_rmw
bash tools/pmc_collect.sh gemm256r_kernel gpurun_out/r06_pmc_proj_loop.json -- python3 $GRAFT_REPO_ROOT/tools/kernel_bench.py > gpurun_out/r06_pmc_proj_loop.log 2>&1 || { tail -5 gpurun_out/r06_pmc_proj_loop.log; exit 2; }
cat gpurun_out/r06_pmc_fc1_loop.json; echo; cat gpurun_out/r06_pmc_proj_loop.json
