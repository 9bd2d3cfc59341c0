#!/bin/bash
# full `-m gpu` suite + smoke() on a GPU box (run through gpurun from the repo root); logs under gpurun_out/.
# pytest writes straight into its log file (no pipe in between: a pipe holds the progress dots back and the box's watchdog takes
# 7 silent minutes for a hang); -v prints one line per test as it finishes.
set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -v > gpurun_out/r02_pytest_gpu.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/r02_pytest_gpu.log | tail -8
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/r02_smoke.log
exit $rc
