#!/usr/bin/env python3
"""A few launches of the W4A8 GEMM (and the W8A8 GEMM) at M x 4096 x 4096 for the profilers."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_extras import _w4_layer  # noqa: E402
from chatglm_q_amd.int4 import hip_ops as h4  # noqa: E402
from chatglm_q_amd.dev import experiments as X  # noqa: E402  (recorded experiments: developer library)
from chatglm_q_amd.int8 import hip_ops as h8  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(17)
K, N = 4096, 4096
layer = _w4_layer(torch, dev, K, N, False, gen)
a8 = X.pack_w4a8(layer.weight, layer.weight_scale)
x = torch.randn(M, K, device=dev, dtype=torch.float16)
a_q, a_s = h8.act_quant_rowwise(x)
for _ in range(12):
    X.w4a8_gemm(a_q, a_s, a8, N, torch.float16)
with torch.no_grad():
    for _ in range(12):
        layer(x)
torch.cuda.synchronize()
