set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_cfg5" -- python3 "$ROOT/bench.py" --model da3_large --image-size 1036 --precision fp8 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/cfg5_prof.json" 2> "$OUT/cfg5_prof.err" || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_cfg2" -- python3 "$ROOT/bench.py" --model da3_small --steps 50 --warmup 5 --no-cpu-baseline --no-extras > "$OUT/cfg2_prof.json" 2> "$OUT/cfg2_prof.err" || exit 3
find "$OUT/prof_cfg5" "$OUT/prof_cfg2" -name '*kernel_trace.csv' -delete
echo done
