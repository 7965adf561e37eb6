import os, sys, torch
sys.path.insert(0, '.')
from bench_extras import _graph_time
from chatglm_q_amd.int8 import hip_ops as h8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(13)
M, K, N = 8192, 4096, 4096
for name, wfn, afn in [("random", lambda: torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g), lambda: torch.randn(M, K, device=dev, dtype=torch.float16)),
                       ("zeros", lambda: torch.zeros((N, K), dtype=torch.int8, device=dev), lambda: torch.zeros(M, K, device=dev, dtype=torch.float16)),
                       ("small (-1..1)", lambda: torch.randint(-1, 2, (N, K), dtype=torch.int8, device=dev, generator=g), lambda: torch.randn(M, K, device=dev, dtype=torch.float16))]:
    tiled = [h8.tile_w8(wfn()) for _ in range(4)]
    sc = (torch.rand(N, device=dev, generator=g) * 0.01 + 0.001).half()
    a = afn()
    a_q, a_s = h8.act_quant_rowwise(a)
    us = _graph_time(torch, dev, lambda: [h8.w8a8_gemm256(a_q, a_s, t, N, sc) for t in tiled]) / 4 * 1e3
    usw = _graph_time(torch, dev, lambda: [h8.w8_gemm256(a, t, N, sc) for t in tiled]) / 4 * 1e3
    print(f"{name}: int8-act 256 kernel {us:.1f} us = {2.0*M*N*K/us/1e6:.0f} TOP/s; weight-only 256 kernel {usw:.1f} us = {2.0*M*N*K/usw/1e6:.0f} TF")
