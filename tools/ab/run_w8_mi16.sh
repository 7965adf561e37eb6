#!/bin/bash
# round 5: int8 weight-only 256-tile GEMM on the 16x16x32 body (product) beside round 4's 32x32x16 body (QLINEAR_G256_MI16=0, developer library),
# interleaved; first the int8 256-tile parity tests on the product library
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests -m gpu -x -q -k "int8 or w8" 2>&1 | tail -4
export QLINEAR_LIB_PATH=chatglm_q_amd/csrc/libqlinear_hip_dev.so
for i in 1 2 3; do
  echo "== 16x16x32 body, half-tile last round (product)"; timeout 600 python tools/w8_256_sweep.py 2>&1 | grep -v amdgpu.ids
  echo "== 32x32x16 body (round 4)"; QLINEAR_G256_MI16=0 timeout 600 python tools/w8_256_sweep.py 2>&1 | grep -v amdgpu.ids
done
} 2>&1 | tee gpurun_out/w8_mi16_ab.txt
