#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_attn_bench_final.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r2_pytest.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench5.json 2> gpurun_out/r2_bench5.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2_bench5.json").read().strip().splitlines()[-1])
print(d["value"], "fps", d["ms_per_step"], "ms/step", "mfma_frac", d["frame_mfma_frac"], {k:(v["ms_per_step"], v.get("tflops"), v.get("frac_mfma_peak"), v.get("gbs")) for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:8]})
PY
