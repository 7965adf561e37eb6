#!/bin/bash
# round 5: the decode attention launch - round 2's group kernel (QLINEAR_ATTENTION_R2=1, developer library) beside the instruction-count rebuild
# (product), both wave counts, the timeline of the stamped build, and the decode loop
mkdir -p gpurun_out
{
export QLINEAR_LIB_PATH=chatglm_q_amd/csrc/libqlinear_hip_dev.so
for i in 1 2; do
  echo "== round 5 kernel, 8 waves"; CAPS=256,1152,4224 timeout 300 python tools/attention_sweep.py
  echo "== round 5 kernel, 4 waves"; QLINEAR_ATTENTION_WAVES=4 CAPS=256,1152,4224 timeout 300 python tools/attention_sweep.py
  echo "== round 2 kernel, 8 waves"; QLINEAR_ATTENTION_R2=1 CAPS=256,1152,4224 timeout 300 python tools/attention_sweep.py
done
echo "== decode, round 5 kernel"; timeout 300 python tools/profile_decode.py 64
echo "== decode, round 2 kernel"; QLINEAR_ATTENTION_R2=1 timeout 300 python tools/profile_decode.py 64
echo "== decode, round 5 kernel"; timeout 300 python tools/profile_decode.py 64
echo "== decode, round 2 kernel"; QLINEAR_ATTENTION_R2=1 timeout 300 python tools/profile_decode.py 64
export QLINEAR_LIB_PATH=tools/ab/libqlinear_hip_attstamps.so
echo "== timeline, round 5 kernel"; timeout 200 python tools/attention_timeline.py; IN_DECODE=1 timeout 300 python tools/attention_timeline.py
echo "== timeline, round 5 kernel, 4 waves"; QLINEAR_ATTENTION_WAVES=4 timeout 200 python tools/attention_timeline.py
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/attention_ab.txt
