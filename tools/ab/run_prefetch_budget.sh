#!/bin/bash
# round 5: the attention launch's prefetch workgroups (next launch's weights into the caches) retuned for the shorter attention chain:
# budget (MB) x start delay, on attention + o_proj (tools/attention_prefetch.py) and on the decode loop
mkdir -p gpurun_out
{
for i in 1 2; do
for v in chatglm_q_amd/csrc/libqlinear_hip.so tools/ab/libqlinear_hip_pf5.so tools/ab/libqlinear_hip_pf6.so tools/ab/libqlinear_hip_pf8.so tools/ab/libqlinear_hip_pf7s4.so tools/ab/libqlinear_hip_pf6s4.so tools/ab/libqlinear_hip_pf9s4.so; do
  echo "== $v"; QLINEAR_LIB_PATH=$v timeout 300 python tools/attention_prefetch.py 2>&1 | grep "capacity   256"
  QLINEAR_LIB_PATH=$v timeout 300 python tools/profile_decode.py 64 2>&1 | grep -o "gen_tok_per_s.: [0-9.]*"
done; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/prefetch_budget.txt
