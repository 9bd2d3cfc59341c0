# split-half mode: start offset in the proj loop (32 k-tiles per tile there: `which` = 1)
set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
for st in "-1,0,-1,-1" "-1,2000,-1,-1" "-1,3500,-1,-1" "-1,0,-1,-1" "-1,2000,-1,-1"; do
  timeout -k 10 300 python bench.py --precision f16x2 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --stagger=$st > gpurun_out/r06_st.json 2> gpurun_out/r06_st.err || exit 1
  python - $st <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r06_st.json").read().strip().splitlines()[-1]); k=d["kernels"]
print(sys.argv[1], d["value"], d["ms_per_step"], {n: k[n]["ms_per_step"] for n in ("qkv_gemm","attention","proj_gemm","fc1_gemm","fc2_gemm") if n in k}, flush=True)
PY
done
