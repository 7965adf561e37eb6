"""Developer tool: a few launches of the int4g32 MFMA GEMM (M=8192, 4096->4096) for rocprofv3 PMC passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
layer = bench_extras._w4_layer(torch, dev, 4096, 4096, False, gen)
x = torch.randn(8192, 4096, device=dev, dtype=torch.float16)
with torch.no_grad():
    for _ in range(6):
        layer(x)
torch.cuda.synchronize()
