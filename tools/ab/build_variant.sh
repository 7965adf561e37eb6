#!/bin/bash
# tools/ab/build_variant.sh NAME -DFLAG...: a library whose w4a8.hip is compiled with the given flags (A/B experiments)
set -e
name=$1; shift
cd "$(dirname "$0")/../../chatglm_q_amd/csrc"
make -j8 libqlinear_hip.so > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -fno-slp-vectorize "$@" -c w4a8.hip -o /tmp/w4a8_$name.o
objs=$(ls *.o | grep -v -e '^w4a8.o$' -e span)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libqlinear_hip_$name.so $objs /tmp/w4a8_$name.o
echo built $name
