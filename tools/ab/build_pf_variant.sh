#!/bin/bash
# tools/ab/build_pf_variant.sh NAME "-DQL_PF_X=.."  -> tools/microbench/libql_pf_NAME.so (prefill_attention.hip rebuilt with the flags;
# run with QLINEAR_LIB_PATH=tools/microbench/libql_pf_NAME.so python tools/prefill_attention_bench.py)
set -e
cd "$(dirname "$0")/../.."
CS=chatglm_q_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16"
/opt/rocm/bin/hipcc $FLAGS $2 -c $CS/prefill_attention.hip -o /tmp/prefill_attention_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/microbench/libql_pf_$1.so $(ls $CS/*.o | grep -v -e "/prefill_attention.o" -e "_span.o" -e "_trace.o" -e "_nomath.o" -e "/dev_" -e "probe_kernels.o") /tmp/prefill_attention_$1.o
