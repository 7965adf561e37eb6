import os, sys
sys.path.insert(0, "/root/repo")
import torch, bench_extras
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
layers = [bench_extras._w4_layer(torch, dev, 4096, 27392, False, gen) for _ in range(6)]
a = torch.randn(8, 4096, device=dev, dtype=torch.float16)
with torch.no_grad():
    for _ in range(5):
        for l in layers:
            l(a)
torch.cuda.synchronize()
