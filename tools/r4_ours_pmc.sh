set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4_vendor; mkdir -p $OUT
python tools/gemm_yardstick.py > $OUT/yardstick.txt 2>&1; cat $OUT/yardstick.txt
SETS="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS;SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT;FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum"
for leg in ours_i8 ours_w4; do
  case $leg in ours_i8) PAT=w8a8_gemm256;; ours_w4) PAT=w4_gemm256;; esac
  bash tools/prof_pmc.sh $PAT "$SETS" python tools/vendor_probe.py $leg > $OUT/pmc_$leg.txt 2>&1
  rm -rf gpurun_out/prof_pmc_$PAT
done
tail -n 25 $OUT/pmc_ours*.txt | cut -c1-200
