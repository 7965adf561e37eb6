for v in 0 1 8 9 16 32 41 57 6 63; do
  if [ $v = 0 ]; then L=""; else L="QLINEAR_LIB_PATH=$PWD/tools/microbench/libql_abl$v.so"; fi
  c=$(env $L python tools/w8a8_config3.py --gemm-only | head -1)
  h=$(env $L W8A8_NSETS=1 python tools/w8a8_config3.py --gemm-only | head -1)
  echo "variant $v | cold $c | hot $h"
done
