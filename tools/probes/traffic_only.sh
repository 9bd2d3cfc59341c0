set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_fetch $OUT/prof_write
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_fetch" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras --dump-launch-order "$OUT/launch_order.json" > "$OUT/bench_fetch.json" 2> "$OUT/prof_fetch.err" || exit 3
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_write" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras --dump-launch-order "$OUT/launch_order.json" > "$OUT/bench_write.json" 2> "$OUT/prof_write.err" || exit 4
python3 "$ROOT/tools/pmc_traffic.py" "$OUT/prof_fetch" "$OUT/prof_write" "$OUT/launch_order.json" "$OUT/traffic.json" 8 bf16 full > "$OUT/traffic.txt" || exit 5
find "$OUT/prof_fetch" "$OUT/prof_write" -name '*counter_collection.csv' -delete
grep -E "attention|qkv|fc1" $OUT/traffic.txt
