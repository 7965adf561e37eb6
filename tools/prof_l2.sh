#!/bin/bash
# L2 request / hit / miss / fabric counters (separate rocprofv3 --pmc passes) of the kernels matching $1 while running "$2..."
#   tools/prof_l2.sh w4_packed_gemm_kernel env SWEEP_M=8192 SWEEP_LAYERS=4 python tools/gemm_sweep.py
set -u
PAT=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_l2_$PAT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $set | tr ' ' '-')
  ( cd $ROOT && timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- "$@" > $OUT/$tag.log 2>&1 < /dev/null )
done
python3 - <<PY
import csv, glob, collections
csv.field_size_limit(1 << 30)
for f in sorted(glob.glob("$OUT/*/pmc_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "$PAT" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k, len(v), round(sum(v) / len(v)))
PY
