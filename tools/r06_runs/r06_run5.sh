set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
cp burn_depth_amd/libmi_depth.so /tmp/good.so
for v in good nostats nocopy good nostats nocopy; do
  if [ $v = good ]; then cp /tmp/good.so burn_depth_amd/libmi_depth.so; else cp burn_depth_amd/libmi_$v.so burn_depth_amd/libmi_depth.so; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r06_v_$v.json 2> gpurun_out/r06_v_$v.err || exit 1
  python - $v <<'PY'
import json,sys
m=sys.argv[1]
d=json.loads(open(f"gpurun_out/r06_v_{m}.json").read().strip().splitlines()[-1]); k=d["kernels"]
print(m, d["value"], d["ms_per_step"], {n: k[n]["ms_per_step"] for n in ("qkv_gemm","proj_gemm","fc1_gemm","fc2_gemm") if n in k}, flush=True)
PY
done
cp /tmp/good.so burn_depth_amd/libmi_depth.so
