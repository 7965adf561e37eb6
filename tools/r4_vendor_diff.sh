#!/bin/bash
# round 4: vendor i8 / f16 GEMM vs this library's 256-tile kernels at 8192 x 4096 x 4096 - kernel trace (names, LDS, registers, grid)
# and the same PMC sets for both (the "timeline diff" of VERDICT r3 item 1b).  Output: gpurun_out/r4_vendor/
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4_vendor; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $ROOT && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python tools/vendor_probe.py all > $OUT/trace.log 2>&1 < /dev/null )
python3 - <<PY > $OUT/trace_summary.txt
import csv, glob, collections
csv.field_size_limit(1 << 30)
for f in glob.glob("$OUT/trace/**/t_kernel_trace.csv", recursive=True):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        if d < 50000: continue
        k = (r["Kernel_Name"][:200], r.get("LDS_Block_Size",""), r.get("Scratch_Size",""), r.get("VGPR_Count",""), r.get("Accum_VGPR_Count",""), r.get("SGPR_Count",""),
             r.get("Workgroup_Size_X", r.get("Workgroup_Size","")), r.get("Grid_Size_X", r.get("Grid_Size","")))
        acc.setdefault(k, []).append(d)
    for k, v in acc.items():
        print(len(v), "x", round(sum(v)/len(v)/1e3, 1), "us min", round(min(v)/1e3,1), "| lds", k[1], "scratch", k[2], "vgpr", k[3], "agpr", k[4], "sgpr", k[5], "wg", k[6], "grid", k[7], "|", k[0])
PY
cat $OUT/trace_summary.txt
cd $ROOT
SETS="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS;SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT;FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum"
for leg in i8 ours_i8 f16 ours_w4; do
  case $leg in i8) PAT=Cijk;; f16) PAT=Cijk;; ours_i8) PAT=w8a8_gemm256;; ours_w4) PAT=w4_gemm256;; esac
  bash tools/prof_pmc.sh $PAT "$SETS" python tools/vendor_probe.py $leg > $OUT/pmc_$leg.txt 2>&1
  rm -rf gpurun_out/prof_pmc_$PAT
done
tail -n 40 $OUT/pmc_*.txt
