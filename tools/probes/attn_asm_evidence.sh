#!/bin/bash
# Evidence for the assembly-owned Depth Pro attention kernel (kernels/attn577_gfx950.s) on a GPU box, from the repo root:
#   gpurun_out/attn_asm/ab.txt         stand-alone A/B against the HIP kernel (md_bench_attention_ex, T = 37 and 296)
#   gpurun_out/attn_asm/stamps.txt     s_memtime stamps of workgroup 0 / wave 0 over its first units + the timing ablations
#   gpurun_out/attn_asm/pmc_asm.json   SQ counters per dispatch of md_attn577_bf16 (tools/pmc_collect.sh: separate --pmc passes)
#   gpurun_out/attn_asm/pmc_hip.json   the same for the HIP kernel at the same shape
#   gpurun_out/attn_asm/inmodel.txt    bench.py --attention-form asm | hip | asm: the `attention` family inside the Depth Pro step
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/attn_asm
mkdir -p "$O"
cd "$ROOT"
timeout -k 10 200 python3 tools/attn_asm/ab_bench.py > "$O/ab.txt" 2>&1 || exit 1
timeout -k 10 400 python3 tools/attn_asm/run_co.py 296 -- "" stamps stamps+stamps2 loop2 noexp nodma nolds nostore nopv+nos+noexp > "$O/stamps.txt" 2>&1 || exit 2
bash tools/pmc_collect.sh md_attn577 "$O/pmc_asm.json" -- python3 "$ROOT/tools/attn_bench.py" 296 577 16 0 0.7 asm > "$O/pmc_asm.log" 2>&1 || exit 3
bash tools/pmc_collect.sh attention_kernel "$O/pmc_hip.json" -- python3 "$ROOT/tools/attn_bench.py" 296 577 16 0 0.7 hip > "$O/pmc_hip.log" 2>&1 || exit 4
cd "$ROOT"
for form in asm hip asm; do
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --attention-form $form > "$O/bench_$form.json" 2> "$O/bench_$form.err" || exit 5
  python3 - "$O/bench_$form.json" "$form" <<'PY' >> "$O/inmodel.txt"
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
a = d["kernels"]["attention"]
print(f"{sys.argv[2]}: {d['value']} frames/s, {d['ms_per_step']} ms/step, attention {a['ms_per_step']} ms = {a['tflops']} TFLOP/s = {a['frac_mfma_peak']} of the peak, box d2d {d['box']['d2d_copy_tbs']} TB/s")
PY
done
rm -rf "$ROOT/gpurun_out/pmc_tmp"
echo done
