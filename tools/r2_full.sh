#!/bin/bash
# full GPU test suite, then the default bench line (round-2 evidence run)
set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_pytest_gpu.log | tail -15 &&
timeout -k 10 400 python bench.py --side-kernels 2>gpurun_out/r02_bench.err | tee gpurun_out/r02_bench.json
