#!/usr/bin/env python3
"""Developer tool: bench_extras.prefill_gemm (int4g32 GEMM at 8192 rows, the four layer shapes) - run once per build / switch
setting for A/B comparisons, e.g.  QLINEAR_GEMM_256=0 python tools/prefill_gemm_ab.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_extras  # noqa: E402

r = bench_extras.prefill_gemm(torch, torch.device("cuda:0"))
print(" ".join(f"{k}: {v['TFLOPs']:.0f} TF ({v['ms'] * 1e3:.0f} us)" for k, v in r.items() if isinstance(v, dict) and "TFLOPs" in v))
