#!/bin/bash
# L2 side of BASELINE config 3's GEMM: request / hit / miss counters of w8a8_tiled_kernel (separate --pmc passes).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_w8a8_l2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z_0-9a-z]*\|TCP_TCC_[A-Z_0-9a-z]*" | sort -u > $OUT/counters.txt
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $set | tr ' ' '-')
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- python $ROOT/tools/w8a8_config3.py --gemm-only > $OUT/$tag.log 2>&1 < /dev/null
done
python3 - <<PY
import csv, glob, collections
csv.field_size_limit(1 << 30)
for f in sorted(glob.glob("$OUT/*/pmc_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "w8a8_tiled_kernel" in r["Kernel_Name"]:
            acc[(r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f.split("/")[-2], k, len(v), sum(v) / len(v))
PY
