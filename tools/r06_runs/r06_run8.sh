set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "layernorm_fold or direct_store" > gpurun_out/r06_raw_tests.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/r06_raw_tests.log | grep -E "FAILED|Error|passed|failed|BAD|assert" | tail -6
[ $rc -eq 0 ] || exit $rc
for b in 8 1; do for m in finish-launch auto finish-launch auto; do
  timeout -k 10 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-extras --ln-fold $m > gpurun_out/r06_raw_${b}_$m.json 2> gpurun_out/r06_raw_$m.err || exit 1
  python - $b $m <<'PY'
import json,sys
b,m=sys.argv[1],sys.argv[2]
d=json.loads(open(f"gpurun_out/r06_raw_{b}_{m}.json").read().strip().splitlines()[-1]); k=d["kernels"]
print('B',b,m, d["value"], d["ms_per_step"], {n: k[n]["ms_per_step"] for n in ("qkv_gemm","proj_gemm","fc1_gemm","fc2_gemm","ln_finish") if n in k}, flush=True)
PY
done; done
