#!/bin/bash
export QLINEAR_LIB_PATH=chatglm_q_amd/csrc/libqlinear_hip_dev.so
{
which rocm-smi amd-smi
for s in o_proj w_in; do
for g in 256 192 128 64; do
  QLINEAR_G256_PERSIST=1 QLINEAR_G256_PGRID=$g timeout 120 python tools/g256p_power.py $s 2>&1 | grep -v amdgpu.ids
done
done
} 2>&1 | tee gpurun_out/g256p_power.txt
