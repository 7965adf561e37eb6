"""Developer tool: host-side cost of one eager QLinear call (no sync inside the loop), and a cProfile of it."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
layer = bench_extras._w4_layer(torch, dev, 4096, 4096, True, gen)
x = torch.randn(1, 1, 4096, device=dev, dtype=torch.float16)
lin = torch.nn.Linear(4096, 4096, bias=True, device=dev, dtype=torch.float16)
with torch.no_grad():
    for f, name in ((layer, "QLinear int4 (HIP)"), (lin, "nn.Linear fp16 (rocBLAS)")):
        for _ in range(50):
            f(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000):
            f(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name}: host {(t1 - t0) / 2000 * 1e6:.1f} us per call, incl. drain {(t2 - t0) / 2000 * 1e6:.1f} us")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        layer(x)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
