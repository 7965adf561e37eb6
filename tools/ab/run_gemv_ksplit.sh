#!/bin/bash
export QLINEAR_LIB_PATH=chatglm_q_amd/csrc/libqlinear_hip_dev.so
{
for ks in 0 1 2 4; do QLINEAR_W4_KSPLIT=$ks timeout 300 python tools/gemv_ksplit_sweep.py 2>&1 | grep -v amdgpu.ids; done
for v in 1 2 4 5 6 7; do QL_VARIANT=$v timeout 300 python tools/gemv_ksplit_sweep.py 2>&1 | grep -v amdgpu.ids | grep "w_out\|qkv"; done
} | tee gpurun_out/r05_gemv_ksplit.txt
