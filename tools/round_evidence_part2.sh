set -o pipefail
export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ev2
O=gpurun_out/ev2
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --batch 1 --steps 20 2>$O/b1.err > $O/r04_b1.json
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --batch 1 --steps 20 --graph 2>>$O/b1.err > $O/r04_b1_graph.json
for p in bf16 fp8 f16x2; do timeout -k 10 200 python bench.py --model da3_large --image-size 1036 --precision $p --graph --no-cpu-baseline --no-extras --steps 20 2>>$O/cfg5.err > $O/r04_cfg5_$p.json; done
for p in bf16 f16x2; do timeout -k 10 200 python bench.py --model da3_small --precision $p --graph --no-cpu-baseline --no-extras --steps 50 2>>$O/cfg5.err > $O/r04_cfg2_small_$p.json; done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --precision f16x2 --steps 5 --warmup 2 2>$O/f16x2.err > $O/r04_bench_f16x2.json
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --precision f16 --steps 5 --warmup 2 2>$O/f16.err > $O/r04_bench_f16.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ev2/r04_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
