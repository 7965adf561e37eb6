# grouped tile order: rows per group (QLINEAR_GEMM_SY) and off (QLINEAR_GEMM_SUPER=0), developer library, 8192-row yardstick
export QLINEAR_LIB_PATH=chatglm_q_amd/csrc/libqlinear_hip_dev.so
for cfg in "SUPER=0" "SY=2" "SY=4" "SY=8"; do
  echo "== $cfg"
  case $cfg in SUPER=0) env QLINEAR_GEMM_SUPER=0 python tools/gemm_yardstick.py w4 i8 2>&1 | grep -v amdgpu.ids;; *) env QLINEAR_GEMM_SY=${cfg#SY=} python tools/gemm_yardstick.py w4 i8 2>&1 | grep -v amdgpu.ids;; esac
done
