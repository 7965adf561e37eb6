"""Developer tool (round 5): the one-row int4g32 GEMV per layer shape beside the pure streaming read of the same bytes (bench_extras.per_shape),
under the developer library's knobs - QLINEAR_W4_KSPLIT (K slices per column quad: 1 / 2 / 4; 0 = the shape rule), QL_VARIANT (ablations).

  QLINEAR_LIB_PATH=chatglm_q_amd/csrc/libqlinear_hip_dev.so QLINEAR_W4_KSPLIT=2 python tools/gemv_ksplit_sweep.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_extras  # noqa: E402

out = bench_extras.per_shape(torch, torch.device("cuda:0"))
tag = f"KSPLIT={os.environ.get('QLINEAR_W4_KSPLIT', 'rule')} VARIANT={os.environ.get('QL_VARIANT', '0')}"
for name, r in out.items():
    if "us" in r:
        print(f"{tag:28s} {name:9s} {r['us']:7.3f} us   pure read {r.get('pure_read_us')} us   frac {r.get('frac_of_pure_read')}", flush=True)
