#!/bin/bash
# Round-4 closing evidence (after the DA3 launch work): full GPU suite, the default bench line, per-configuration lines, config 2's
# timeline under rocprofv3. Outputs under gpurun_out/ev3/ (copy what is judged into profiles/).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ev3
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=15 > "$OUT/r04_pytest_gpu.log" 2>&1; rc=$?
tail -3 "$OUT/r04_pytest_gpu.log"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > "$OUT/r04_bench.json" 2> "$OUT/bench.err" || exit 3
python - "$OUT/r04_bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench", d["value"], d["roofline"]["frac"], [(c.get("name", "")[:30], c.get("value")) for c in d.get("configs", [])])
PY
line() { timeout -k 10 300 python bench.py "$@" 2>/dev/null; }
line --model da3_small --precision bf16 --graph --steps 200 --warmup 20 > "$OUT/r04_cfg2_small_bf16.json"
line --model da3_small --precision f16x2 --graph --steps 200 --warmup 20 > "$OUT/r04_cfg2_small_f16x2.json"
line --model da3_small --precision f32 --graph --steps 100 --warmup 10 > "$OUT/r04_cfg2_small_f32.json"
line --model da3_small --precision bf16 --batch 8 --graph --steps 50 --warmup 5 > "$OUT/r04_cfg2_small_bf16_b8.json"
line --model da3_large --image-size 1036 --precision bf16 --graph --steps 50 --warmup 5 > "$OUT/r04_cfg5_bf16.json"
line --model da3_large --image-size 1036 --precision fp8 --graph --steps 50 --warmup 5 > "$OUT/r04_cfg5_fp8.json"
line --model da3_large --image-size 1036 --precision f16x2 --graph --steps 30 --warmup 5 > "$OUT/r04_cfg5_f16x2.json"
line --model da3_large --image-size 518 --precision bf16 --graph --steps 100 --warmup 10 > "$OUT/r04_da3_large_518_bf16.json"
for f in "$OUT"/r04_cfg*.json "$OUT"/r04_da3*.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$(basename $f)', d['value'], d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_cfg2" -- python3 "$ROOT/bench.py" --model da3_small --graph --steps 30 --warmup 5 --no-cpu-baseline --no-extras > "$OUT/cfg2_prof.json" 2> "$OUT/cfg2_prof.err" || exit 4
cd "$ROOT"
f=$(find "$OUT/prof_cfg2" -name "*kernel_trace.csv" | head -1)
python tools/probes/cfg2_timeline.py "$f" > "$OUT/r04_cfg2_timeline_after.txt"
head -3 "$OUT/r04_cfg2_timeline_after.txt"
rm -rf "$OUT/prof_cfg2"
echo done
