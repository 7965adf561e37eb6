#!/bin/bash
# Arbitrary PMC counters (one rocprofv3 --pmc pass per ';'-separated set, kernel trace only) of the kernels matching PAT:
#   tools/prof_pmc.sh PAT "GRBM_GUI_ACTIVE;SQ_INSTS_VALU SQ_INSTS_MFMA" cmd...
# Prints the mean per dispatch of every counter and the mean dispatch duration (-> clock = GRBM_GUI_ACTIVE / duration).
set -u
PAT=$1; SETS=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_pmc_$PAT; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra ARR <<< "$SETS"
i=0
for set in "${ARR[@]}"; do
  i=$((i+1))
  ( cd $ROOT && timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/set$i -o pmc -- "$@" > $OUT/set$i.log 2>&1 < /dev/null )
done
python3 - <<PY
import csv, glob, collections
csv.field_size_limit(1 << 30)
for f in sorted(glob.glob("$OUT/*/pmc_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "$PAT" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:48], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k, len(v), round(sum(v) / len(v)))
for f in sorted(glob.glob("$OUT/*/pmc_kernel_trace.csv"))[:1]:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "$PAT" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:48], r.get("Grid_Size_X", ""))].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, v in sorted(acc.items()):
        print("duration_ns (under pmc)", k, len(v), round(sum(v) / len(v)))
PY
