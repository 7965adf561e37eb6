# round 5: PMC passes of the int4g32 256-tile GEMM at 8192 x 4096 x 4096: the product's 16x16x32 body (w4_gemm256x16_kernel) and round 4's 32x32x16
# body beside it (developer library, QLINEAR_G256_MI16=0) - SQ_WAIT_ANY share, LDS / VALU / MFMA instruction counts, fabric traffic
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5_pmc; mkdir -p $OUT
SETS="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS;SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT;FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum"
bash tools/prof_pmc.sh w4_gemm256 "$SETS" python tools/vendor_probe.py ours_w4 > $OUT/pmc_16x16.txt 2>&1
rm -rf gpurun_out/prof_pmc_w4_gemm256
QLINEAR_LIB_PATH=chatglm_q_amd/csrc/libqlinear_hip_dev.so QLINEAR_G256_MI16=0 bash tools/prof_pmc.sh w4_gemm256 "$SETS" python tools/vendor_probe.py ours_w4 > $OUT/pmc_32x32.txt 2>&1
rm -rf gpurun_out/prof_pmc_w4_gemm256
tail -n 30 $OUT/pmc_*.txt | cut -c1-200
