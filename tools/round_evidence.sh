#!/bin/bash
# One gpurun call that re-creates the round's measured artifacts under gpurun_out/: tools/refresh_profiles.sh (default bench
# line with side kernels, rocprofv3 kernel stats, PMC traffic passes) plus the other configurations quoted in DESIGN.md
# (B = 1 eager / graph, f16 with the accuracy report, f32, config 5 in bf16 / fp8, config 2). Copy what is judged into profiles/.
set -o pipefail
export PYTHONUNBUFFERED=1
R=${ROUND:-r05}
export ROUND=$R
bash tools/refresh_profiles.sh || exit 1
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --batch 1 --steps 20 2>gpurun_out/b1.err > gpurun_out/${R}_b1.json
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --batch 1 --steps 20 --graph 2>>gpurun_out/b1.err > gpurun_out/${R}_b1_graph.json
for p in bf16 fp8; do timeout -k 10 200 python bench.py --model da3_large --image-size 1036 --precision $p --graph --no-cpu-baseline --no-extras --steps 20 2>gpurun_out/cfg5.err > gpurun_out/${R}_cfg5_$p.json; done
timeout -k 10 200 python bench.py --model da3_large --image-size 1036 --precision fp8 --no-cpu-baseline --no-extras --steps 20 2>>gpurun_out/cfg5.err > gpurun_out/${R}_cfg5_fp8_eager.json
timeout -k 10 200 python bench.py --model da3_small --graph --no-cpu-baseline --no-extras --steps 50 2>>gpurun_out/cfg5.err > gpurun_out/${R}_cfg2_small_graph.json
timeout -k 10 300 python bench.py --no-cpu-baseline --precision f16 --accuracy 2>gpurun_out/f16.err > gpurun_out/${R}_bench_f16.json
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --precision f32 --steps 3 --warmup 1 2>gpurun_out/f32.err > gpurun_out/${R}_bench_f32_full.json
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --precision f16x2 --steps 5 --warmup 2 2>gpurun_out/f16x2.err > gpurun_out/${R}_bench_f16x2.json
timeout -k 10 200 python bench.py --model da3_small --no-cpu-baseline --no-extras --steps 50 2>>gpurun_out/cfg5.err > gpurun_out/${R}_cfg2_small_eager.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r0[0-9]_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
