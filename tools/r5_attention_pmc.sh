#!/bin/bash
# round 5: PMC passes of the decode attention launch (capacity 256, the group kernel of round 5 and round 2's beside it): instruction counts per
# kind, wave cycles / waits, bytes from the fabric - the counters behind profiles/r05_attention_timeline.txt
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5_att_pmc; mkdir -p $OUT
SETS="GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS;SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT;FETCH_SIZE;TCC_HIT_sum TCC_MISS_sum;TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
export QLINEAR_LIB_PATH=chatglm_q_amd/csrc/libqlinear_hip_dev.so
CAPS=256 bash tools/prof_pmc.sh decode_attention_group "$SETS" python tools/attention_sweep.py > $OUT/pmc_round5.txt 2>&1
rm -rf gpurun_out/prof_pmc_decode_attention_group
QLINEAR_ATTENTION_R2=1 CAPS=256 bash tools/prof_pmc.sh decode_attention_mfma "$SETS" python tools/attention_sweep.py > $OUT/pmc_round2.txt 2>&1
rm -rf gpurun_out/prof_pmc_decode_attention_mfma
tail -n 40 $OUT/pmc_*.txt | cut -c1-200
