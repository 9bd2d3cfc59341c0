#!/bin/bash
# SQ / GRBM counters of one kernel, in separate rocprofv3 --pmc passes of the same command (8 SQ slots per pass), averaged
# per dispatch of the kernels whose name contains $1 and written as JSON.
# usage: tools/pmc_collect.sh <kernel-substring> <out.json> -- <program> <args...>     (run on the GPU box, from the repo root)
# <program> must be the GPU program ITSELF (python3 tools/attn_bench.py ..., ./bench ...): with --pmc the profiler initialises the GPU
# before the program starts, so an env / bash -c / launcher hop in front of it is an exec from a GPU-initialised process, which
# the GPU boxes refuse. Exits non-zero when any pass fails.
set -o pipefail
KERN="$1"; OUTJ="$2"; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
case "$OUTJ" in /*) ;; *) OUTJ="$ROOT/$OUTJ";; esac
W=$ROOT/gpurun_out/pmc_tmp
rm -rf "$W"; mkdir -p "$W"
cd /tmp && export TMPDIR=/tmp
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
        "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
        "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES")
i=0
FAILED=0
for P in "${PASSES[@]}"; do
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d "$W/p$i" -- "$@" > "$W/p$i.out" 2> "$W/p$i.err" || { echo "pass $i failed"; tail -5 "$W/p$i.err"; FAILED=1; }
  tail -2 "$W/p$i.out"; ls "$W/p$i" 2>/dev/null | head -3
  i=$((i+1))
done
python3 - "$KERN" "$OUTJ" "$W" <<'PY'
import csv, glob, json, sys, collections
kern, outj, w = sys.argv[1:4]
acc = collections.defaultdict(list)
for f in glob.glob(w + "/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for d in per.values():
        for k, v in d.items():
            acc[k].append(v)
out = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
out["_dispatches"] = {k: len(v) for k, v in sorted(acc.items())}  # per counter: every counter is collected in exactly one pass
json.dump(out, open(outj, "w"), indent=1)
print(json.dumps(out))
PY
rm -rf "$W"
exit $FAILED
