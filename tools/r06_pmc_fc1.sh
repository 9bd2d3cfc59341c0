#!/bin/bash
# SQ counters of the shipped fc1 kernel form (256 x 256 tile, bias + GELU store; md_bench_gemm on the fc1 shape at B = 4) and of the
# read-modify-write form (proj shape): where a K = 1024 tile's cycles go (round-5 review, item 1: the counters beside the kernel that ships).
set -o pipefail
export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
export ONLY=fc1B4 TILES=256pp16_gelu
bash tools/pmc_collect.sh gemm256_kernel gpurun_out/r06_pmc_fc1_gelu.json -- python3 $GRAFT_REPO_ROOT/tools/kernel_bench.py > gpurun_out/r06_pmc_fc1.log 2>&1 || { tail -5 gpurun_out/r06_pmc_fc1.log; exit 1; }
cd $GRAFT_REPO_ROOT
export ONLY=projB4 TILES=256pp16_rmw
bash tools/pmc_collect.sh gemm256_kernel gpurun_out/r06_pmc_proj_rmw.json -- python3 $GRAFT_REPO_ROOT/tools/kernel_bench.py > gpurun_out/r06_pmc_proj.log 2>&1 || { tail -5 gpurun_out/r06_pmc_proj.log; exit 2; }
cat gpurun_out/r06_pmc_fc1_gelu.json; echo; cat gpurun_out/r06_pmc_proj_rmw.json
