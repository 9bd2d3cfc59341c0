#!/bin/bash
# Round-6 measured artifacts in one gpurun call (run from the repo root on a GPU box): the round's evidence set of tools/round_evidence.sh
# (default bench line with side kernels, rocprofv3 kernel stats, PMC traffic passes, B = 1 eager / graph, f16 / f32 / f16x2, configs 2 and 5),
# the rocprofv3 kernel stats of configs 2 / 5, the LayerNorm-fold A/B, and host_io with two host threads under either attention form.
set -o pipefail
export PYTHONUNBUFFERED=1
export ROUND=r06
cd $GRAFT_REPO_ROOT
bash tools/round_evidence.sh > gpurun_out/r06_evidence.log 2>&1 || { tail -5 gpurun_out/r06_evidence.log; exit 1; }
tail -25 gpurun_out/r06_evidence.log
cd $GRAFT_REPO_ROOT
bash tools/da3_profiles.sh > gpurun_out/r06_da3_profiles.log 2>&1 || { tail -5 gpurun_out/r06_da3_profiles.log; exit 2; }
cd $GRAFT_REPO_ROOT
for f in asm hip; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --host-io --attention-form $f --steps 5 --warmup 2 > gpurun_out/r06_host_io_$f.json 2> gpurun_out/r06_host_io_$f.err || exit 3
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --ln-fold off --steps 10 --warmup 3 > gpurun_out/r06_bench_ln_unfolded.json 2> gpurun_out/r06_unfolded.err || exit 4
python - <<'PY'
import json
for f in ("asm", "hip"):
    d = json.loads(open(f"gpurun_out/r06_host_io_{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], json.dumps(d.get("host_io"))[:600])
d = json.loads(open("gpurun_out/r06_bench_ln_unfolded.json").read().strip().splitlines()[-1])
print("ln unfolded", d["value"], d["ms_per_step"])
PY
