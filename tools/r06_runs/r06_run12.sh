set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
for st in "-1,-1,0,0" "-1,-1,1700,0" "-1,-1,800,0" "-1,-1,0,1600" "-1,-1,0,800" "-1,-1,1700,1600" "0,0,0,0" "-1,-1,0,0"; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --stagger=$st > gpurun_out/r06_st.json 2> gpurun_out/r06_st.err || exit 1
  python - $st <<'PY'
import json,sys
m=sys.argv[1]
d=json.loads(open("gpurun_out/r06_st.json").read().strip().splitlines()[-1]); k=d["kernels"]
print(m, d["value"], d["ms_per_step"], {n: k[n]["ms_per_step"] for n in ("qkv_gemm","attention","proj_gemm","fc1_gemm","fc2_gemm") if n in k}, flush=True)
PY
done
