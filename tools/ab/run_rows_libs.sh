#!/bin/bash
# small-row int4 forward with different builds of the library (tools/ab/libqlinear_hip_<name>.so; "cur" = product)
out=gpurun_out/rows_libs.txt; : > $out
for lib in "$@"; do
  if [ $lib = cur ]; then unset QLINEAR_LIB_PATH; else export QLINEAR_LIB_PATH=$PWD/tools/ab/libqlinear_hip_$lib.so; fi
  echo "== $lib" >> $out
  timeout 300 python tools/gemv_rows.py ${ROWS:-2 4} 2>/dev/null < /dev/null >> $out
done
cat $out
