#!/bin/bash
# A/B of builds of the DEVELOPER library on the W4A8 GEMM (M = 8192; a recorded experiment since round 4, chatglm_q_amd/dev/experiments.py):
# tools/ab/libqlinear_hip_<name>.so are developer builds (-DQL_DEV_TUNING ...), selected through QLINEAR_DEV_LIB_PATH; "cur" = chatglm_q_amd/csrc/libqlinear_hip_dev.so
out=gpurun_out/w4a8_ab.txt; : > $out
for colg in ${COLGS:-0 1}; do
  for lib in ${LIBS:-cur pkd1 pkd2}; do
    if [ $lib = cur ]; then unset QLINEAR_DEV_LIB_PATH; else export QLINEAR_DEV_LIB_PATH=$PWD/tools/ab/libqlinear_hip_$lib.so; fi
    echo "== COLG=$colg lib=$lib" >> $out
    QLINEAR_W4A8_COLG=$colg timeout 200 python tools/w4a8_sweep.py ${ROWS:-8192} 2>/dev/null < /dev/null | grep -E '"(qkv|o_proj|w_in|w_out)|w4a8_gemm_TOPs' | paste - - >> $out
    if [ -n "$PARITY" ]; then QLINEAR_W4A8_COLG=$colg timeout 300 python -m pytest tests/test_dev_experiments_gpu.py -q -x -k w4a8 2>&1 | tail -1 >> $out; fi
  done
done
cat $out
