set -o pipefail
mkdir -p gpurun_out/ks2
timeout -k 10 400 python -m pytest tests -x -q -m gpu -k "check_attention or split_half_operators or depth_anything3_small_end_to_end or config2" > gpurun_out/ks2/tests.log 2>&1; rc=$?; tail -4 gpurun_out/ks2/tests.log
[ $rc -ne 0 ] && { grep -n "BAD" gpurun_out/ks2/tests.log | head -20; exit $rc; }
echo "== attn bench, small-launch form on"; timeout -k 10 200 python tools/attn_bench.py 2>&1 | tee gpurun_out/ks2/attn_on.txt | cut -c1-150
echo "== off"; MD_ATTN_KEYSPLIT=0 timeout -k 10 200 python tools/attn_bench.py 2>&1 | tee gpurun_out/ks2/attn_off.txt | cut -c1-150
for p in bf16 f16x2; do
for k in 1 0; do echo "== config 2 $p keysplit=$k"; MD_ATTN_KEYSPLIT=$k timeout -k 10 200 python bench.py --model da3_small --precision $p --graph --steps 300 --warmup 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done; done
for k in 1 0; do echo "== da3_large 518 keysplit=$k"; MD_ATTN_KEYSPLIT=$k timeout -k 10 200 python bench.py --model da3_large --image-size 518 --precision bf16 --graph --steps 100 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
