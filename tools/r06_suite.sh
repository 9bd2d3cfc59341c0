#!/bin/bash
# full `-m gpu` suite + smoke() on a GPU box (run through gpurun from the repo root); logs under gpurun_out/.
set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -v > gpurun_out/r06_pytest_gpu.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/r06_pytest_gpu.log | grep -E "FAILED|Error|passed|failed|BAD" | tail -12
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/r06_smoke.log | tail -5
exit $rc
