#!/bin/bash
# Round-4 closing evidence after the DA3 head regrouping: full GPU suite, the default bench line, config 2's timeline under rocprofv3.
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
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_cfg2" -- python3 "$ROOT/bench.py" --model da3_small --graph --steps 30 --warmup 5 --no-cpu-baseline --no-extras > "$OUT/cfg2_prof.json" 2> "$OUT/cfg2_prof.err" || exit 4
cd "$ROOT"
f=$(find "$OUT/prof_cfg2" -name "*kernel_trace.csv" | head -1)
python tools/probes/cfg2_timeline.py "$f" > "$OUT/r04_cfg2_timeline_after.txt"
head -3 "$OUT/r04_cfg2_timeline_after.txt"
rm -rf "$OUT/prof_cfg2"
echo done
