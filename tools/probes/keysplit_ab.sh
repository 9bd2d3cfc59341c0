set -o pipefail
mkdir -p gpurun_out/ks
timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "check_attention or split_half_operators" > gpurun_out/ks/attn_check.log 2>&1; rc=$?; tail -5 gpurun_out/ks/attn_check.log
[ $rc -ne 0 ] && { grep -n "BAD" gpurun_out/ks/attn_check.log | head -20; exit $rc; }
echo "== attn bench, key split on"; timeout -k 10 200 python tools/attn_bench.py 2>&1 | tee gpurun_out/ks/attn_on.txt | grep "N=1370\|N=2738\|N=5477"
echo "== attn bench, key split off"; MD_ATTN_KEYSPLIT=0 timeout -k 10 200 python tools/attn_bench.py 2>&1 | tee gpurun_out/ks/attn_off.txt | grep "N=1370\|N=2738\|N=5477"
for p in bf16 f16x2; do
echo "== config 2 $p on";  timeout -k 10 200 python bench.py --model da3_small --precision $p --graph --steps 200 --warmup 20 2>/dev/null | tee gpurun_out/ks/cfg2_${p}_on.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "== config 2 $p off"; MD_ATTN_KEYSPLIT=0 timeout -k 10 200 python bench.py --model da3_small --precision $p --graph --steps 200 --warmup 20 2>/dev/null | tee gpurun_out/ks/cfg2_${p}_off.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
echo "== da3_large 518 bf16 on"; timeout -k 10 200 python bench.py --model da3_large --image-size 518 --precision bf16 --graph --steps 100 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "== da3_large 518 bf16 off"; MD_ATTN_KEYSPLIT=0 timeout -k 10 200 python bench.py --model da3_large --image-size 518 --precision bf16 --graph --steps 100 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
