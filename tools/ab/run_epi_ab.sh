# A/B of two builds of the weight-only 256-tile kernel, interleaved runs (run-to-run spread is 3 - 5 %)
for i in 1 2 3; do
  echo "== new"; python tools/gemm_yardstick.py w4 2>&1 | grep -v amdgpu.ids
  echo "== old epilogue"; QLINEAR_LIB_PATH=tools/microbench/libql_g256_oldepi.so python tools/gemm_yardstick.py w4 2>&1 | grep -v amdgpu.ids
done
