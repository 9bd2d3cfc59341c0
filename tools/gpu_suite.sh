#!/bin/bash
# full `-m gpu` suite + smoke() on a GPU box (run through gpurun from the repo root); logs under gpurun_out/
set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_pytest_gpu.log | tail -6 &&
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_smoke.log
