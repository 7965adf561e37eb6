"""Developer tool: MFMA GEMM time vs M for the int4g32 and int8 weight-only kernels (4096 -> 4096, fp16).
QLINEAR_GEMM_MT=1|2|4 forces the tile height; unset = the library's heuristic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras
from chatglm_q_amd.int8.qlinear import DynamicQuantizeLinear as L8

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
N = int(os.environ.get("SWEEP_N", 4096)); K = int(os.environ.get("SWEEP_K", 4096))
NL = int(os.environ.get("SWEEP_LAYERS", 36))     # rotating weight sets: > 256 MB of weights, so none is cache-resident
l4s = [bench_extras._w4_layer(torch, dev, K, N, False, gen) for _ in range(NL)]
for l in l4s:
    l.weight = None if False else l.weight       # canonical copy stays (module state); the kernels read the packed one
l8s = []
for _ in range(NL):
    l8 = L8(K, N, bias=False, dtype=torch.float16, device=dev)
    l8.weight.copy_(torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev))
    l8.weight_scale.copy_((torch.rand(N, device=dev) * 0.01).half())
    l8s.append(l8)
out = []
for M in [int(x) for x in os.environ.get('SWEEP_M', '8,16,32,64,128,256,512,1024,2048').split(',')]:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    row = [M]
    for layers in (l4s, l8s):
        def f():
            with torch.no_grad():
                for l in layers:
                    l(a)
        ms = bench_extras._graph_time(torch, dev, f) / NL
        row += [round(ms * 1e3, 1), round(2 * M * N * K / ms / 1e9, 1)]
    out.append(row)
print("MT=%s KS=%s  M  w4_us w4_TF  w8_us w8_TF" % (os.environ.get("QLINEAR_GEMM_MT", "auto"), os.environ.get("QLINEAR_GEMM_KSPLIT", "auto")))
for r in out:
    print("  ", *r)
