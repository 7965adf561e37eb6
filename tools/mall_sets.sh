#!/bin/bash
# Headline GEMV per-launch time against the number of weight sets in the rotation: 2 sets sit in the L2s, 8-24 sets (75-225 MB)
# in the Infinity Cache, 72 sets (680 MB) come from HBM.  Output: gpurun_out/mall_sets.txt
out=gpurun_out/mall_sets.txt; : > $out
for s in ${@:-2 8 16 24 48 72}; do
  timeout 150 python bench.py --no-cpu-baseline --sets $s < /dev/null 2>/dev/null | tail -1 > /tmp/line.json
  python - $s >> $out <<'PY'
import json, sys
d = json.load(open("/tmp/line.json"))
r = d["roofline"]
print("sets", sys.argv[1], "ms_per_step", d["ms_per_step"], "us_per_launch", r["us_per_launch"], "span", r.get("kernel_span"))
PY
done
cat $out
