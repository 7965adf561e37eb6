#!/bin/bash
# tools/ab/build_variant.sh NAME SRC -DFLAG...: a library whose SRC (e.g. w4a8.hip) is compiled with the given flags
set -e
name=$1; src=$2; shift 2
base=${src%.hip}
cd "$(dirname "$0")/../../chatglm_q_amd/csrc"
make -j8 libqlinear_hip.so > /dev/null
extra=""; [ $base = w4a8 ] && extra="-fno-slp-vectorize"; [ $base = w4_rows4 ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc -DQL_DEV_TUNING -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 $extra "$@" -c $src -o /tmp/${base}_$name.o
objs=$(ls *.o | grep -v -e "^$base.o\$" -e span -e "^dev_" -e trace -e nomath)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libqlinear_hip_$name.so $objs /tmp/${base}_$name.o
echo built $name
