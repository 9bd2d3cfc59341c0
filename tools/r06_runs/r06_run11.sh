set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
for st in "0,0" "2000,0" "1000,0" "3000,0" "2000,1500" "2000,3000" "0,0"; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --stagger $st > gpurun_out/r06_st.json 2> gpurun_out/r06_st.err || exit 1
  python - $st <<'PY'
import json,sys
m=sys.argv[1]
d=json.loads(open("gpurun_out/r06_st.json").read().strip().splitlines()[-1]); k=d["kernels"]
print(m, d["value"], d["ms_per_step"], {n: k[n]["ms_per_step"] for n in ("qkv_gemm","proj_gemm","fc1_gemm","fc2_gemm") if n in k}, flush=True)
PY
done
