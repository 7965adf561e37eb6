#!/bin/bash
# int4 f16-MFMA GEMM (M = 8192, the four layer shapes) with different builds of the library (tools/ab/libqlinear_hip_<name>.so)
out=gpurun_out/gemm_libs.txt; : > $out
for lib in "$@"; do
  if [ $lib = cur ]; then unset QLINEAR_LIB_PATH; else export QLINEAR_LIB_PATH=$PWD/tools/ab/libqlinear_hip_$lib.so; fi
  echo "== $lib" >> $out
  timeout 200 python tools/w4a8_sweep.py ${ROWS:-8192} 2>/dev/null < /dev/null | grep -E '"(qkv|o_proj|w_in|w_out)|w4a16_TFLOPs' | paste - - >> $out
  if [ -n "$PARITY" ]; then timeout 300 python -m pytest tests/test_parity_gpu.py -q -k "int4_vs_oracle or config5" 2>&1 | tail -1 >> $out; fi
done
cat $out
