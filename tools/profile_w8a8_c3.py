"""Developer tool: BASELINE config 3 alone (act-quant + int8 x int8 GEMM, 512 x 4096 -> 4096, 20 weight sets) for rocprofv3 --kernel-trace
--stats: the kernel averages land in profiles/rNN_summary.json (w8a8_config3_under_rocprofv3), from which bench_extras recomputes
extras.w8a8_config3.frac_rocprof."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatglm_q_amd.int8 import hip_ops as h8  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(13)
M, K, N = 512, 4096, 4096
tiled = [h8.tile_w8(torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)) for _ in range(20)]
sc = (torch.rand(N, device=dev, generator=g) * 0.01 + 0.001).half()
a = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
for it in range(15):
    for t in tiled:
        a_q, a_s = h8.act_quant_rowwise(a)
        h8.w8a8_gemm_tiled(a_q, a_s, t, N, sc)
torch.cuda.synchronize()
