#!/bin/bash
# Re-creates the judged measurement artifacts on a GPU box (run from the repo root through gpurun):
#   gpurun_out/prof_fetch|prof_write  separate --pmc passes (FETCH_SIZE / WRITE_SIZE) of the bench command
#   gpurun_out/traffic.json           per-launch HBM bytes per kernel family (tools/pmc_traffic.py); also copied to
#                                     profiles/${ROUND}_traffic.json (ROUND defaults to r05) of the box's snapshot so the bench line below carries it
#   gpurun_out/bench.json             default bench line (roofline.traffic + cpu_baseline included)
#   gpurun_out/prof_stats/            rocprofv3 --kernel-trace --stats of the same command (+ its own bench line)
# Copy the summaries into profiles/ afterwards.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
BATCH=${BATCH:-8}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_fetch" -- \
  python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras --dump-launch-order "$OUT/launch_order.json" \
  > "$OUT/bench_fetch.json" 2> "$OUT/prof_fetch.err" || exit 3
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_write" -- \
  python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras --dump-launch-order "$OUT/launch_order.json" \
  > "$OUT/bench_write.json" 2> "$OUT/prof_write.err" || exit 4
python3 "$ROOT/tools/pmc_traffic.py" "$OUT/prof_fetch" "$OUT/prof_write" "$OUT/launch_order.json" "$OUT/traffic.json" \
  "$BATCH" bf16 full > "$OUT/traffic.txt" || exit 5
cp "$OUT/traffic.json" "$ROOT/profiles/${ROUND:-r05}_traffic.json"
timeout -k 10 500 python3 "$ROOT/bench.py" --steps 10 --warmup 3 --side-kernels > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -- \
  python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/bench_prof.json" 2> "$OUT/prof_stats.err" || exit 2
# keep the merge-back small: drop the raw per-dispatch counter CSVs, keep stats + summaries
find "$OUT/prof_fetch" "$OUT/prof_write" -name '*counter_collection.csv' -delete
find "$OUT/prof_stats" -name '*kernel_trace.csv' -delete
echo refreshed
