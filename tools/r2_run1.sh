#!/bin/bash
# round-2 GPU session 1: new precision mode + cleaned GEMM kernel parity, full-size accuracy, throughput of the three modes
set -o pipefail
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 500 python tools/gpu_diag.py --only ops,tiny,f16 > gpurun_out/r2_diag1.log 2>&1; echo "diag1 rc=$?"
tail -25 gpurun_out/r2_diag1.log
timeout -k 10 400 python tools/gpu_diag.py --only full > gpurun_out/r2_full.log 2>&1; echo "full rc=$?"
grep -E "full|BAD|====" gpurun_out/r2_full.log | tail -20
for p in bf16 f16 f32; do
  st=5; [ $p = f32 ] && st=2
  timeout -k 10 300 python bench.py --precision $p --steps $st --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench_$p.json 2> gpurun_out/r2_bench_$p.err; echo "bench $p rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r2_bench_$p.json").read().strip().splitlines()[-1])
    print("$p", d["value"], "fps", d["ms_per_step"], "ms/step", "mfma_frac", d["frame_mfma_frac"], {k:(v["ms_per_step"], v.get("tflops")) for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:8]})
except Exception as e:
    print("$p parse failed", e); print(open("gpurun_out/r2_bench_$p.err").read()[-1500:])
PY
done
