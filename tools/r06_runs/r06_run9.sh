set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 600 python tools/r06_persist_check.py > gpurun_out/r06_persist_check.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/r06_persist_check.log | tail -8
[ $rc -eq 0 ] || exit $rc
for m in fc1qkv on fc1qkv on; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --persistent-fc1 $m > gpurun_out/r06_pf_$m.json 2> gpurun_out/r06_pf_$m.err || exit 1
  python - $m <<'PY'
import json,sys
m=sys.argv[1]
d=json.loads(open(f"gpurun_out/r06_pf_{m}.json").read().strip().splitlines()[-1]); k=d["kernels"]
print(m, d["value"], d["ms_per_step"], {n: k[n]["ms_per_step"] for n in ("qkv_gemm","proj_gemm","fc1_gemm","fc2_gemm","dec_conv3x3") if n in k}, flush=True)
PY
done
