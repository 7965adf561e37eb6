for v in r4base r4yp1 r4yp2; do echo == $v; QLINEAR_LIB_PATH=tools/microbench/libql_g256_$v.so python tools/gemm_yardstick.py w4 2>&1 | grep -v amdgpu.ids; done
