#!/bin/bash
# Measured timeline of ONE launch of the W8A8 tile-major GEMM (config 3): builds w8a8.hip with -DQL_W8A8_STAMPS into
# tools/microbench/libql_stamps.so (build step, no GPU), runs it on a GPU box (run step).
set -e
cd "$(dirname "$0")/.."
CS=chatglm_q_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16"
if [ "$1" = build ]; then
  make -C $CS >/dev/null
  /opt/rocm/bin/hipcc $FLAGS -DQL_W8A8_STAMPS -c $CS/w8a8.hip -o /tmp/w8a8_stamps.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/microbench/libql_stamps.so $(ls $CS/*.o | grep -v -e "/w8a8.o" -e "_span.o" -e "/dev_" -e "_trace.o" -e "probe_kernels.o" -e nomath) /tmp/w8a8_stamps.o
else
  shift || true
  QLINEAR_LIB_PATH=$PWD/tools/microbench/libql_stamps.so python tools/w8a8_timeline.py "$@"
fi
