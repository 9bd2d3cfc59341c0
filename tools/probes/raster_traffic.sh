#!/bin/bash
# L2-side (fabric) read bytes and time of the qkv / fc1 GEMMs as a function of the raster group width (n-tiles walked together inside an
# XCD's tile range): the round-4 review read qkv's 3.4x algorithmic traffic as lost time. Needs a DIAG build (make DIAG=1: the
# MD_GEMM_RASTER_GN switch does not exist in the shipped library). Run on the GPU box from the repo root:
#   make -C burn_depth_amd/csrc DIAG=1 -B -j16 && bash tools/probes/raster_traffic.sh > gpurun_out/raster_traffic.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for gn in 0 2 3 4 6; do
  export MD_GEMM_RASTER_GN=$gn ONLY=qkvB4,fc1B4 TILES=256pp16
  echo "== raster_gn=$gn (0 = plain n-fastest)"
  timeout -k 10 200 python3 $ROOT/tools/kernel_bench.py 2>&1 | grep -E "^(qkvB4|fc1B4)"
  rm -rf /tmp/rt_$gn
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/rt_$gn -- python3 $ROOT/tools/kernel_bench.py > /dev/null 2>&1
  python3 - /tmp/rt_$gn <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "gemm256_kernel" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r["Grid_Size"])].append(float(r["Counter_Value"]))
for (k, g), v in sorted(acc.items()):
    # FETCH_SIZE: KB, and half of the bytes of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section): x 1024 x 2
    print(f"   grid {g:>9s}: fabric reads {sum(v) / len(v) * 2048 / 1e6:9.1f} MB per launch over {len(v)} launches")
PY
done
