#!/usr/bin/env python3
"""int4g32 forward time at 1, 2 (default; or argv) rows per ChatGLM2-6B layer shape (rotating weight sets, one HIP graph)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_extras import _graph_time, _w4_layer  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(5)
for name, K, N in [("qkv_proj", 4096, 4608), ("o_proj", 4096, 4096), ("w_in", 4096, 27392), ("w_out", 13696, 4096)]:
    n = max(4, min(40, (700 << 20) // (K * N // 2)))
    layers = [_w4_layer(torch, dev, K, N, False, gen) for _ in range(n)]
    row = [name]
    for M in ([int(v) for v in sys.argv[1:]] or [1, 2]):
        x = torch.randn(M, K, device=dev, dtype=torch.float16)
        def f():
            with torch.no_grad():
                for l in layers:
                    l(x)
        row.append(round(_graph_time(torch, dev, f) / n * 1e3, 2))
    print(*row)
    del layers
    torch.cuda.empty_cache()
