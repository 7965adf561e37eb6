"""Developer tool (round 4): the two-launch many-row weight-only GEMM (dequantise once per call into a 16-bit image, dense ring GEMM;
csrc/w4_dense256.hip, developer library) beside the fused 256-tile kernel: parity against it and time, bench protocol."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_extras  # noqa: E402
from bench_extras import _graph_time  # noqa: E402
from chatglm_q_amd.dev import experiments as X  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
M = int(os.environ.get("M", 8192))
for name, K, N in [("qkv_proj", 4096, 4608), ("o_proj", 4096, 4096), ("w_in", 4096, 27392), ("w_out", 13696, 4096)]:
    layers = [bench_extras._w4_layer(torch, dev, K, N, name == "qkv_proj", g) for _ in range(4)]
    x = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
    with torch.no_grad():
        want = layers[0](x)
        img = X.dense256_image(layers[0].tiled(), N, K, torch.float16)
        got = X.dense256_forward(x, img, N, layers[0].bias)
        rel = ((got.float() - want.float()).norm() / want.float().norm()).item()
        scratch = torch.empty(img.numel(), dtype=torch.uint8, device=dev)
        tiled = [l.tiled() for l in layers]
        us_fused = _graph_time(torch, dev, lambda: [l(x) for l in layers]) / 4 * 1e3
        us_exp = _graph_time(torch, dev, lambda: [X.dense256_image(t, N, K, torch.float16, out=scratch) for t in tiled]) / 4 * 1e3
        imgs = [X.dense256_image(t, N, K, torch.float16) for t in tiled]
        us_gemm = _graph_time(torch, dev, lambda: [X.dense256_forward(x, im, N, l.bias) for im, l in zip(imgs, layers)]) / 4 * 1e3
        us_both = _graph_time(torch, dev, lambda: [X.dense256_forward(x, X.dense256_image(t, N, K, torch.float16, out=scratch), N, l.bias)
                                                   for t, l in zip(tiled, layers)]) / 4 * 1e3
    fl = 2.0 * M * N * K
    print(f"{name:9s} {M}x{K}x{N}: fused {us_fused:7.1f} us {fl / us_fused / 1e6:5.0f} TF | expand {us_exp:6.1f} us + dense GEMM {us_gemm:7.1f} us "
          f"({fl / us_gemm / 1e6:5.0f} TF) | both in one stream {us_both:7.1f} us {fl / us_both / 1e6:5.0f} TF | rel diff vs fused {rel:.2e}", flush=True)
    del layers, imgs, tiled
    torch.cuda.empty_cache()
