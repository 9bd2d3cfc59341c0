set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -v -k "f16x2 or storage or decoder_from or split_half or config1 or small_preset or tile_parallel" > gpurun_out/r06_f16x2_tests.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/r06_f16x2_tests.log | grep -E "FAILED|Error|passed|failed|BAD" | tail -12
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline --no-extras --precision f16x2 --steps 5 --warmup 2 > gpurun_out/r06_bench_f16x2.json 2> gpurun_out/r06_bench_f16x2.err
rc=$?
tail -c 300 gpurun_out/r06_bench_f16x2.json
exit $rc
