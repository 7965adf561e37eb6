#!/usr/bin/env python3
"""Config 3's GEMM (int8 x int8, 512 x 4096 x 4096) with / without the grid-level K split of the developer library (SPLITK=0 / 1:
128 x 128 tiles x 2 K slices, w8a8_tiled_kernel<.., SK = 2> behind qlinear_dev_w8a8_fwd_tiled_splitk), one process per setting:
  SPLITK=1 python tools/w8a8_splitk_ab.py
Prints the time per launch over 20 weight sets (graph replay, HIP events) and checks the output bit for bit against the integer
reference (torch._int_mm of the same int8 operands, the kernel's epilogue formula in fp32)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_extras import _graph_time  # noqa: E402
from chatglm_q_amd.int8 import hip_ops as h8  # noqa: E402
from chatglm_q_amd.dev import experiments as X  # noqa: E402

USE = os.environ.get("SPLITK", "1") != "0"


def gemm(a_q, a_s, t, N, sc, bias=None):
    if USE and X.w8a8_splitk_serves(a_q.shape[0], N, a_q.shape[1]):
        return X.w8a8_gemm_tiled_splitk(a_q, a_s, t, N, sc, bias)
    return h8.w8a8_gemm_tiled(a_q, a_s, t, N, sc, bias)


dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(13)
out = {"splitk": int(USE)}
for M, K, N in [(512, 4096, 4096), (256, 4096, 4096), (384, 4096, 4096), (1024, 4096, 4096), (512, 4096, 4608)]:
    nsets = 20
    ws = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g) for _ in range(nsets)]
    tiled = [h8.tile_w8(w) for w in ws]
    sc = (torch.rand(N, device=dev, generator=g) * 0.01 + 0.001).half()
    bias = (torch.randn(N, device=dev, generator=g) * 0.1).half()
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    a_q, a_s = h8.act_quant_rowwise(a)
    exact = True
    for rep in range(3):                                  # repeated calls: the tickets' parity / epoch scheme
        for i in (0, 7, 19):
            got = gemm(a_q, a_s, tiled[i], N, sc, bias)
            want = h8.w8a8_gemm_tiled(a_q, a_s, tiled[i], N, sc, bias)      # the product's 64 x 128-tile kernel
            exact = exact and bool(torch.equal(got, want))
    us = _graph_time(torch, dev, lambda: [gemm(a_q, a_s, t, N, sc) for t in tiled]) / nsets * 1e3
    us1 = _graph_time(torch, dev, lambda: [gemm(a_q, a_s, tiled[0], N, sc) for _ in range(nsets)]) / nsets * 1e3
    fused = _graph_time(torch, dev, lambda: [gemm(*h8.act_quant_rowwise(a), t, N, sc) for t in tiled]) / nsets * 1e3
    out[f"{M}x{K}x{N}"] = {"gemm_us": round(us, 2), "gemm_us_one_weight_set": round(us1, 2), "fused_us": round(fused, 2), "bit_equal_to_unsplit_kernel": exact, "served_by_splitk": bool(USE and X.w8a8_splitk_serves(M, N, K))}
    del ws, tiled
    torch.cuda.empty_cache()
print(json.dumps(out))
