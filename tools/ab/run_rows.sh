#!/bin/bash
# small-row int4 forward under different environment settings: tools/gemv_rows.py ROWS per setting
out=gpurun_out/rows_env.txt; : > $out
for setting in "$@"; do
  echo "== $setting" >> $out
  env $setting timeout 300 python tools/gemv_rows.py ${ROWS:-1 2 3 4} 2>/dev/null < /dev/null >> $out
done
cat $out
