# interleaved A/B of two builds on the int4g32 yardstick: product vs tools/ab/libqlinear_hip_$1.so
for i in 1 2 3; do
  echo "== product"; python tools/gemm_yardstick.py w4 2>&1 | grep -v amdgpu.ids
  echo "== $1"; QLINEAR_LIB_PATH=tools/ab/libqlinear_hip_$1.so python tools/gemm_yardstick.py w4 2>&1 | grep -v amdgpu.ids
done
