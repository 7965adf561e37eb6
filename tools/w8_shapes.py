"""Developer tool: int8 weight-only GEMV time at the ChatGLM2-6B layer shapes (M = 1, fp16), rotating weight sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras
from chatglm_q_amd.int8.qlinear import DynamicQuantizeLinear as L8
dev = torch.device("cuda:0")
for name, K, N, NL in (("qkv", 4096, 4608, 16), ("o", 4096, 4096, 16), ("w_in", 4096, 27392, 4), ("w_out", 13696, 4096, 6), ("lm_head", 4096, 65024, 2)):
    layers = []
    for _ in range(NL):
        l = L8(K, N, bias=False, dtype=torch.float16, device=dev)
        l.weight.copy_(torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev))
        l.weight_scale.copy_((torch.rand(N, device=dev) * 0.01).half())
        layers.append(l)
    for M in (1, 2, 4):
        a = torch.randn(M, K, device=dev, dtype=torch.float16)
        def f():
            with torch.no_grad():
                for l in layers:
                    l(a)
        t = bench_extras._graph_time(torch, dev, f) / NL
        print(f"{name} {K}->{N} M={M}: {t*1e3:.1f} us  {(N*K + 2*N + 2*M*(K+N))/t/1e9:.0f} GB/s", flush=True)
    del layers
