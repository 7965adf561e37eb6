#!/bin/bash
# W8A8 tile-major GEMM alone (512 and 8192 rows x 4096 x 4096) under different environment settings; parity when PARITY=1
out=gpurun_out/w8a8_env.txt; : > $out
for setting in "$@"; do
  echo "== $setting" >> $out
  env $setting timeout 200 python tools/w8a8_config3.py --gemm-only 2>/dev/null < /dev/null >> $out
  if [ -n "$PARITY" ]; then env $setting timeout 300 python -m pytest tests/test_parity_gpu.py -q -x -k "w8a8" 2>&1 | tail -1 >> $out; fi
done
cat $out
