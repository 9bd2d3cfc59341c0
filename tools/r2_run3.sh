#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for pin in 0 1; do
  echo "== MD_ATTN_PIN=$pin"
  MD_ATTN_PIN=$pin timeout -k 10 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_attn_bench_pin$pin.log
done
timeout -k 10 300 python - > gpurun_out/r2_attn_check.log 2>&1 <<'PY'
import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import gpu_diag as d
from burn_depth_amd.depth_pro import Device
dev=Device(0)
d.check_attention(dev)
bad=[r for r in d.RESULTS if not r[3]]
print(len(d.RESULTS)-len(bad),"/",len(d.RESULTS))
PY
echo "attn check rc=$?"; grep -E "BAD|/ " gpurun_out/r2_attn_check.log | head -40
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r2_pytest.log
for pin in 0 1; do
MD_ATTN_PIN=$pin timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench3_pin$pin.json 2> gpurun_out/r2_bench3.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r2_bench3_pin$pin.json").read().strip().splitlines()[-1])
print("pin$pin", d["value"], "fps", d["ms_per_step"], "ms/step", "mfma_frac", d["frame_mfma_frac"], {k:(v["ms_per_step"], v.get("tflops"), v.get("frac_mfma_peak"), v.get("gbs")) for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:8]}, d["kernels"].get("pyramid_patchify"))
PY
done
