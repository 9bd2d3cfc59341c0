set -o pipefail
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])"; }
run dp_b1 --batch 1
run dp_b1_graph --batch 1 --graph
run dp_b4 --batch 4
run dp_b16 --batch 16 --steps 5
run da3L --model da3_large
run da3L_graph --model da3_large --graph
run da3L_b8 --model da3_large --batch 8
run da3L_1036 --model da3_large --image-size 1036
run da3L_1036_fp8 --model da3_large --image-size 1036 --precision fp8
run da3S --model da3_small
run da3S_graph --model da3_small --graph
run da3S_b8 --model da3_small --batch 8
run da3L_graph_fp8 --model da3_large --graph --precision fp8
run da3L_b8_fp8 --model da3_large --batch 8 --precision fp8
run da3S_b8_fp8 --model da3_small --batch 8 --precision fp8
