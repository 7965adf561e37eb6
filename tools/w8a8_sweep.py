"""Developer tool: W8A8 fused op (act-quant + i8 MFMA GEMM) time for M x 4096 -> 4096; QLINEAR_W8A8_MT / _KSPLIT override."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras
from chatglm_q_amd.int8 import hip_ops as h8

dev = torch.device("cuda:0")
N = K = 4096
NL = 24
ws = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev) for _ in range(NL)]
sc = (torch.rand(N, device=dev) * 0.01).half()
out = []
for M in [int(x) for x in os.environ.get("SWEEP_M", "128,512,2048,8192").split(",")]:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    aq, asc = h8.act_quant_rowwise(a)
    def fused():
        for w in ws:
            h8.w8a8_forward(a, w, sc)
    def quant_only():
        for w in ws:
            h8.act_quant_rowwise(a)
    t = bench_extras._graph_time(torch, dev, fused) / NL
    tq = bench_extras._graph_time(torch, dev, quant_only) / NL
    out.append((M, round(t * 1e3, 1), round(tq * 1e3, 1), round(2 * M * N * K / (t - tq) / 1e9, 1), round(2 * M * N * K / t / 1e9, 1)))
print("MT=%s KS=%s   M  fused_us  quant_us  gemm_TOPs  fused_TOPs" % (os.environ.get("QLINEAR_W8A8_MT", "auto"), os.environ.get("QLINEAR_W8A8_KSPLIT", "auto")))
for r in out:
    print("  ", *r)
