#!/bin/bash
# round 5: persistent 256-tile int4 GEMM: bit-equality against the one-tile-per-workgroup kernel, then an interleaved A/B at 8192 rows
export QLINEAR_LIB_PATH=chatglm_q_amd/csrc/libqlinear_hip_dev.so
mkdir -p gpurun_out
{
QLINEAR_GEMM_256_MIN_BLOCKS=1 QLINEAR_G256_PERSIST=0 timeout 300 python tools/g256p_check.py save
for g in 16 24 40 64; do
  echo "== PGRID $g"; QLINEAR_GEMM_256_MIN_BLOCKS=1 QLINEAR_G256_PERSIST=1 QLINEAR_G256_PGRID=$g timeout 300 python tools/g256p_check.py check
done
for i in 1 2 3; do
  echo "== persistent, last round as half tiles"; QLINEAR_G256_PERSIST=1 timeout 600 python tools/gemm_yardstick.py w4 2>&1 | grep -v amdgpu.ids
  echo "== persistent, whole tiles only (peel)"; QLINEAR_G256_PERSIST=1 QLINEAR_G256_TAIL=0 timeout 600 python tools/gemm_yardstick.py w4 2>&1 | grep -v amdgpu.ids
  echo "== one tile per workgroup"; QLINEAR_G256_PERSIST=0 timeout 600 python tools/gemm_yardstick.py w4 2>&1 | grep -v amdgpu.ids
done
} 2>&1 | tee gpurun_out/g256p_ab.txt
