"""Developer tool (round 4): this library's many-row kernels beside the vendor's plain GEMMs under ONE protocol (bench_extras._graph_time:
4 launches on 4 weight sets per graph, best of 3 replays), random operands, M = 8192, the four layer shapes.

  python tools/gemm_yardstick.py [w4] [i8] [vendor]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_extras  # noqa: E402
from bench_extras import _graph_time  # noqa: E402

dev = torch.device("cuda:0")
what = set(sys.argv[1:]) or {"w4", "i8", "vendor"}
M = int(os.environ.get("M", 8192))
g = torch.Generator(device=dev).manual_seed(5)
shapes = [("qkv_proj", 4096, 4608), ("o_proj", 4096, 4096), ("w_in", 4096, 27392), ("w_out", 13696, 4096)]
for name, K, N in shapes:
    line = [f"{name:9s} {M}x{K}x{N}:"]
    flops = 2.0 * M * N * K
    if "vendor" in what:
        a = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
        ws = [torch.randn(N, K, device=dev, dtype=torch.float16, generator=g) * 0.05 for _ in range(4)]
        us = _graph_time(torch, dev, lambda: [a @ w.t() for w in ws]) / 4 * 1e3
        line.append(f"vendor f16 {us:7.1f} us {flops / us / 1e6:6.0f} TF |")
        del ws
    if "w4" in what:
        layers = [bench_extras._w4_layer(torch, dev, K, N, False, g) for _ in range(4)]
        x = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
        with torch.no_grad():
            us = _graph_time(torch, dev, lambda: [l(x) for l in layers]) / 4 * 1e3
        line.append(f"ours int4g32 {us:7.1f} us {flops / us / 1e6:6.0f} TF |")
        del layers
    if "i8" in what or "vendor" in what:
        K8 = K - K % 128
        ops = 2.0 * M * N * K8
        aq = torch.randint(-127, 128, (M, K8), dtype=torch.int8, device=dev, generator=g)
        wq = [torch.randint(-127, 128, (N, K8), dtype=torch.int8, device=dev, generator=g) for _ in range(4)]
        if "vendor" in what and N % 8 == 0:
            us = _graph_time(torch, dev, lambda: [torch._int_mm(aq, w.t()) for w in wq]) / 4 * 1e3
            line.append(f"vendor i8 {us:7.1f} us {ops / us / 1e6:6.0f} TOP/s |")
        if "i8" in what:
            from chatglm_q_amd.int8 import hip_ops as h8
            tiled = [h8.tile_w8(w) for w in wq]
            sc = (torch.rand(N, device=dev, generator=g) * 0.01 + 0.001).half()
            a_s = torch.rand(M, device=dev, generator=g) * 0.01 + 0.001
            us = _graph_time(torch, dev, lambda: [h8.w8a8_gemm_tiled(aq, a_s, t, N, sc) for t in tiled]) / 4 * 1e3
            line.append(f"ours i8xi8 {us:7.1f} us {ops / us / 1e6:6.0f} TOP/s")
            del tiled
        del wq
    print(" ".join(line), flush=True)
    torch.cuda.empty_cache()
