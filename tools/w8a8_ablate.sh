#!/bin/bash
# Builds ablation variants of w8a8.hip (QL_W8A8_ABLATE bit set, see the file) into tools/microbench/libql_abl<N>.so and,
# on a GPU box, times BASELINE config 3's GEMM with each:  tools/w8a8_ablate.sh build | run
set -e
cd "$(dirname "$0")/.."
CS=chatglm_q_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16"
VARIANTS="${VARIANTS:-1 2 4 6 8 16 32 40 63 64 127 128 191 255}"
if [ "$1" = build ]; then
  make -C $CS >/dev/null
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc $FLAGS -DQL_W8A8_ABLATE=$v -c $CS/w8a8.hip -o /tmp/w8a8_abl$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/microbench/libql_abl$v.so $(ls $CS/*.o | grep -v "w8a8.o\|_span.o\|_trace.o\|_nomath.o") /tmp/w8a8_abl$v.o
  done
else
  echo "variant 0 (product)"; python tools/w8a8_config3.py --gemm-only
  for v in $VARIANTS; do
    echo "variant $v"; QLINEAR_LIB_PATH=$PWD/tools/microbench/libql_abl$v.so python tools/w8a8_config3.py --gemm-only
  done
fi
