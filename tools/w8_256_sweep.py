#!/usr/bin/env python3
"""Developer tool: int8 weight-only GEMM at prefill row counts - the 256 x 256-tile kernel (qlinear_w8_fwd_tiled256) beside what the
dispatch of qlinear_w8_fwd_tiled picks (QLINEAR_W8_256=0: the 128-row-tile kernel)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_extras import _graph_time  # noqa: E402
from chatglm_q_amd.int8 import hip_ops as h8  # noqa: E402
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(13)
for M, K, N in [(8192, 4096, 4608), (8192, 4096, 4096), (8192, 4096, 27392), (8192, 13696, 4096), (4096, 4096, 4096)]:
    nsets = 3
    tiled = [h8.tile_w8(torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)) for _ in range(nsets)]
    sc = (torch.rand(N, device=dev, generator=g) * 0.01 + 0.001).half()
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    us_new = _graph_time(torch, dev, lambda: [h8.w8_gemm256(a, t, N, sc) for t in tiled]) / nsets * 1e3
    us_auto = _graph_time(torch, dev, lambda: [h8.w8_forward_tiled(a, t, N, sc) for t in tiled]) / nsets * 1e3
    fl = 2.0 * M * N * K
    print(f"{M}x{K}x{N}: 256-tile kernel {us_new:.1f} us = {fl / us_new / 1e6:.0f} TF | dispatch (QLINEAR_W8_256={os.environ.get('QLINEAR_W8_256', 'auto')}) "
          f"{us_auto:.1f} us = {fl / us_auto / 1e6:.0f} TF")
    del tiled
    torch.cuda.empty_cache()
