set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gelu or linear or persistent or layernorm_fold or direct_store or config1 or smoke" > gpurun_out/r06_gelu_tests.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/r06_gelu_tests.log | tail -4
[ $rc -eq 0 ] || exit $rc
for m in on on; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --persistent-fc1 $m > gpurun_out/r06_g_$m.json 2> gpurun_out/r06_g_$m.err || exit 1
  python - $m <<'PY'
import json,sys
m=sys.argv[1]
d=json.loads(open(f"gpurun_out/r06_g_{m}.json").read().strip().splitlines()[-1]); k=d["kernels"]
print(m, d["value"], d["ms_per_step"], {n: k[n]["ms_per_step"] for n in ("qkv_gemm","proj_gemm","fc1_gemm","fc2_gemm","dec_conv3x3") if n in k}, d["box"]["d2d_copy_tbs"], flush=True)
PY
done
