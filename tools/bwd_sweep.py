"""Developer tool: backward (grad_A) kernel times, 4096 x 4096 weights, fp16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras
from chatglm_q_amd.int4 import hip_ops as h4
from chatglm_q_amd.int8 import hip_ops as h8

dev = torch.device("cuda:0")
K = N = 4096
NL = 12
q4 = [torch.randint(0, 256, (K // 2, N), dtype=torch.uint8, device=dev) for _ in range(NL)]
s4 = (torch.rand(K // 32, N, device=dev) * 0.01 + 0.001).half()
w8 = [torch.randint(-127, 128, (K, N), dtype=torch.int8, device=dev) for _ in range(NL)]     # (K, N) contiguous
s8 = (torch.rand(N, device=dev) * 0.01 + 0.001).half()
for M in (64, 512, 2048, 8192):
    g = torch.randn(M, N, device=dev, dtype=torch.float16)
    def f4():
        for q in q4:
            h4.w4_grad_input(g, q, s4)
    def f8():
        for w in w8:
            h8.w8_grad_input(g, w, s8)
    t4 = bench_extras._graph_time(torch, dev, f4) / NL
    t8 = bench_extras._graph_time(torch, dev, f8) / NL
    print(f"M={M}: int4 bwd {t4*1e3:.1f} us ({2*M*N*K/t4/1e9:.0f} TF)   int8 bwd {t8*1e3:.1f} us ({2*M*N*K/t8/1e9:.0f} TF)")
