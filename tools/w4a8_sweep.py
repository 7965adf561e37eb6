#!/usr/bin/env python3
"""BASELINE config 5's QLinear calls (M = 8192 rows) through the three int4 paths side by side: f16 MFMA with in-register
dequant (W4A16, reference rounding), W4A8 (i8 MFMA, act-quant + GEMM), and the GEMM of W4A8 alone."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_extras import _graph_time, _w4_layer  # noqa: E402
from chatglm_q_amd.int4 import hip_ops as h4  # noqa: E402
from chatglm_q_amd.dev import experiments as X  # noqa: E402  (recorded experiments: developer library)
from chatglm_q_amd.int8 import hip_ops as h8  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(17)
    rows = [int(v) for v in sys.argv[1:]] or [8192]
    out = {}
    for M in rows:
        for name, K, N in [("qkv_proj", 4096, 4608), ("o_proj", 4096, 4096), ("w_in", 4096, 27392), ("w_out", 13696, 4096)]:
            layer = _w4_layer(torch, dev, K, N, False, gen)
            a8 = X.pack_w4a8(layer.weight, layer.weight_scale)
            x = torch.randn(M, K, device=dev, dtype=torch.float16)
            a_q, a_s = h8.act_quant_rowwise(x)
            flops = 2.0 * M * N * K
            with torch.no_grad():
                ms16 = _graph_time(torch, dev, lambda: layer(x))
            ms8 = _graph_time(torch, dev, lambda: X.w4a8_forward(x, a8, N))
            ms8g = _graph_time(torch, dev, lambda: X.w4a8_gemm(a_q, a_s, a8, N, torch.float16))
            out[f"{name} M={M}"] = {"w4a16_f16_mfma_ms": round(ms16, 4), "w4a16_TFLOPs": round(flops / ms16 / 1e9, 1),
                                    "w4a8_linear_ms": round(ms8, 4), "w4a8_linear_TOPs": round(flops / ms8 / 1e9, 1),
                                    "w4a8_gemm_ms": round(ms8g, 4), "w4a8_gemm_TOPs": round(flops / ms8g / 1e9, 1)}
            del layer, a8, x
            torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
