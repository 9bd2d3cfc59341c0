set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gemm_family or storage or layernorm_fold or small_preset or tiny or linear or operator or conv" > gpurun_out/r06_ds_tests.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/r06_ds_tests.log | grep -E "FAILED|Error|passed|failed|BAD" | tail -8
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/gpu_diag.py --only linear > gpurun_out/r06_ds_diag.log 2>&1 || { tail -5 gpurun_out/r06_ds_diag.log; exit 2; }
tail -2 gpurun_out/r06_ds_diag.log
for m in off on off on; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --direct-store $m > gpurun_out/r06_ds_$m.json 2> gpurun_out/r06_ds_$m.err || exit 1
  python - $m <<'PY'
import json,sys
m=sys.argv[1]
d=json.loads(open(f"gpurun_out/r06_ds_{m}.json").read().strip().splitlines()[-1]); k=d["kernels"]
print(m, d["value"], d["ms_per_step"], {n: k[n]["ms_per_step"] for n in ("qkv_gemm","proj_gemm","fc1_gemm","fc2_gemm","dec_conv3x3","enc_proj","patch_embed") if n in k}, flush=True)
PY
done
