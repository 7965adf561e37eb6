#!/bin/bash
# tools/ab/build_g256_variant.sh NAME "-DQL_G256_X=.."  -> tools/microbench/libql_g256_NAME.so (w4_gemm256.hip rebuilt with the flags)
set -e
cd "$(dirname "$0")/../.."
CS=chatglm_q_amd/csrc
FLAGS="-DQL_DEV_TUNING -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16"
/opt/rocm/bin/hipcc $FLAGS $2 -c $CS/w4_gemm256.hip -o /tmp/w4_gemm256_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/microbench/libql_g256_$1.so $(ls $CS/*.o | grep -v "w4_gemm256.o\|_span.o\|_trace.o\|_nomath.o\|dev_") /tmp/w4_gemm256_$1.o
