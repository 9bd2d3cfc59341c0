set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
for m in off neutral auto off neutral auto; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --ln-fold $m > gpurun_out/r06_ab_$m.json 2> gpurun_out/r06_ab_$m.err || exit 1
  python - $m <<'PY'
import json,sys
m=sys.argv[1]
d=json.loads(open(f"gpurun_out/r06_ab_{m}.json").read().strip().splitlines()[-1]); k=d["kernels"]
print(m, d["value"], d["ms_per_step"], {n: k[n]["ms_per_step"] for n in ("layernorm","qkv_gemm","proj_gemm","fc1_gemm","fc2_gemm","attention") if n in k}, flush=True)
PY
done
