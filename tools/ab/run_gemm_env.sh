#!/bin/bash
# int4 f16-MFMA GEMM (M = 8192, the four layer shapes) under different environment settings: one line per setting
out=gpurun_out/gemm_env.txt; : > $out
for setting in "$@"; do
  echo "== $setting" >> $out
  env $setting timeout 200 python tools/w4a8_sweep.py ${ROWS:-8192} 2>/dev/null < /dev/null | grep -E '"(qkv|o_proj|w_in|w_out)|w4a16_TFLOPs' | paste - - >> $out
done
cat $out
