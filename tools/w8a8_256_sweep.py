import os, sys, torch
sys.path.insert(0, '.')
from bench_extras import _graph_time
from chatglm_q_amd.int8 import hip_ops as h8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(13)
for M, K, N in [(8192, 4096, 4096), (8192, 4096, 13696), (8192, 13696, 4096), (4096, 4096, 4096), (2048, 4096, 13696)]:
    nsets = 4
    if K % 128: K -= K % 128
    tiled = [h8.tile_w8(torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)) for _ in range(nsets)]
    sc = (torch.rand(N, device=dev, generator=g) * 0.01 + 0.001).half()
    a_q, a_s = h8.act_quant_rowwise(torch.randn(M, K, device=dev, dtype=torch.float16))
    us_new = _graph_time(torch, dev, lambda: [h8.w8a8_gemm256(a_q, a_s, t, N, sc) for t in tiled]) / nsets * 1e3
    os.environ["X"] = "1"
    us_auto = _graph_time(torch, dev, lambda: [h8.w8a8_gemm_tiled(a_q, a_s, t, N, sc) for t in tiled]) / nsets * 1e3
    ops = 2.0 * M * N * K
    print(f"{M}x{K}x{N}: 256-tile kernel {us_new:.1f} us = {ops / us_new / 1e6:.0f} TOP/s | dispatch ({os.environ.get('QLINEAR_W8A8_256', 'auto')}) {us_auto:.1f} us = {ops / us_auto / 1e6:.0f} TOP/s")
    del tiled
    torch.cuda.empty_cache()
