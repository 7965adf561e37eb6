#!/usr/bin/env python3
"""BASELINE config 3 (int8 weights x int8-quantised activations, 512 x 4096 -> 4096): the two kernels timed apart and
together, round-1 path (row-major W through LDS) beside the tile-major path; other row counts for orientation.
Same protocol as bench.py: 20 distinct weight sets, one HIP graph, HIP events on the launch stream."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_extras import _graph_time  # noqa: E402
from chatglm_q_amd.int8 import hip_ops as h8  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    if "--gemm-only" in sys.argv:                     # ablation runs: results are wrong by construction, only time counts
        g = torch.Generator(device=dev).manual_seed(13)
        for M, K, N in [(512, 4096, 4096), (8192, 4096, 4096)]:
            nsets = int(os.environ.get("W8A8_NSETS", 20 if M == 512 else 4))     # 1: the same operands every launch (cache-hot)
            tiled = [h8.tile_w8(torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)) for _ in range(nsets)]
            sc = (torch.rand(N, device=dev, generator=g) * 0.01 + 0.001).half()
            a_q, a_s = h8.act_quant_rowwise(torch.randn(M, K, device=dev, dtype=torch.float16))
            nl = max(nsets, 20 if M == 512 else 4)
            us = _graph_time(torch, dev, lambda: [h8.w8a8_gemm_tiled(a_q, a_s, tiled[i % nsets], N, sc) for i in range(nl)]) / nl * 1e3
            print(f"  {M}x{K}x{N}: gemm {us:.2f} us ({nsets} weight sets)")
        return
    g = torch.Generator(device=dev).manual_seed(13)
    out = {}
    for M, K, N in [(512, 4096, 4096), (128, 4096, 4096), (2048, 4096, 4096), (8192, 4096, 4096), (8192, 4096, 27392 // 2)]:
        nsets = max(2, min(20, (600 << 20) // (N * K)))
        ws = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g) for _ in range(nsets)]
        tiled = [h8.tile_w8(w) for w in ws]
        sc = (torch.rand(N, device=dev, generator=g) * 0.01 + 0.001).half()
        a = torch.randn(M, K, device=dev, dtype=torch.float16)
        a_q, a_s = h8.act_quant_rowwise(a)
        ops = 2.0 * M * N * K
        r = {}
        r["act_quant_us"] = _graph_time(torch, dev, lambda: [h8.act_quant_rowwise(a) for _ in range(nsets)]) / nsets * 1e3
        r["gemm_tiled_us"] = _graph_time(torch, dev, lambda: [h8.w8a8_gemm_tiled(a_q, a_s, t, N, sc) for t in tiled]) / nsets * 1e3
        r["fused_tiled_us"] = _graph_time(torch, dev, lambda: [h8.w8a8_forward_tiled(a, t, N, sc) for t in tiled]) / nsets * 1e3
        r["fused_rowmajor_r1_us"] = _graph_time(torch, dev, lambda: [h8.w8a8_forward(a, w, sc) for w in ws]) / nsets * 1e3
        r["gemm_tiled_TOPs"] = ops / r["gemm_tiled_us"] / 1e6
        r["fused_tiled_TOPs"] = ops / r["fused_tiled_us"] / 1e6
        r["fused_frac_of_5POPs"] = r["fused_tiled_TOPs"] / 5000.0
        out[f"{M}x{K}x{N}"] = {k: round(v, 3) for k, v in r.items()}
        del ws, tiled
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
