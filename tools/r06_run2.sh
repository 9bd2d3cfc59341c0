set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -v -k "layernorm_fold" > gpurun_out/r06_fold_tests.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/r06_fold_tests.log | grep -E "PASS|FAIL|Error|error|assert|BAD" | tail -30
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r06_bench_fold_on.json 2> gpurun_out/r06_bench_fold_on.err
rc=$?
tail -c 600 gpurun_out/r06_bench_fold_on.json
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --ln-fold off > gpurun_out/r06_bench_fold_off.json 2> gpurun_out/r06_bench_fold_off.err
rc=$?
tail -c 300 gpurun_out/r06_bench_fold_off.json
exit $rc
