# the decoder's lean 3 x 3 convolutions on the tile loop (persist bit 8): bit identity, then on | noconv pairs
set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 600 python tools/r06_persist_check.py > gpurun_out/r06_persist_check.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/r06_persist_check.log | tail -10
[ $rc -eq 0 ] || exit $rc
for m in on noconv on noconv; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --persistent-fc1 $m > gpurun_out/r06_st.json 2> gpurun_out/r06_st.err || exit 1
  python - $m <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r06_st.json").read().strip().splitlines()[-1]); k=d["kernels"]
print(sys.argv[1], d["value"], d["ms_per_step"], {n: k[n]["ms_per_step"] for n in ("qkv_gemm","attention","proj_gemm","fc1_gemm","fc2_gemm","dec_conv3x3") if n in k}, d["box"]["d2d_copy_tbs"], flush=True)
PY
done
