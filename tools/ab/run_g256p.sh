#!/bin/bash
# round 5: the int4g32 256-tile launch: half-tile last round bit-equal to whole tiles (small grids per round through QLINEAR_G256_PGRID), then an
# interleaved A/B at 8192 rows: 16x16x32 body with / without the half-tile last round, and the 32x32x16 body (QLINEAR_G256_MI16=0)
export QLINEAR_LIB_PATH=chatglm_q_amd/csrc/libqlinear_hip_dev.so
mkdir -p gpurun_out
{
QLINEAR_GEMM_256_MIN_BLOCKS=1 QLINEAR_G256_TAIL=0 timeout 300 python tools/g256p_check.py save
for g in 16 24 40 64 128; do
  echo "== PGRID $g"; QLINEAR_GEMM_256_MIN_BLOCKS=1 QLINEAR_G256_PGRID=$g timeout 300 python tools/g256p_check.py check
done
for i in 1 2 3; do
  echo "== 16x16x32, last round as half tiles (product)"; timeout 600 python tools/gemm_yardstick.py w4 2>&1 | grep -v amdgpu.ids
  echo "== 16x16x32, whole tiles only (peel)"; QLINEAR_G256_TAIL=0 timeout 600 python tools/gemm_yardstick.py w4 2>&1 | grep -v amdgpu.ids
  echo "== 32x32x16, whole tiles only (peel): round 4's kernel"; QLINEAR_G256_MI16=0 timeout 600 python tools/gemm_yardstick.py w4 2>&1 | grep -v amdgpu.ids
done
} 2>&1 | tee gpurun_out/g256p_ab.txt
