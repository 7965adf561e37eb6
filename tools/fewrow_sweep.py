"""Developer tool: int4g32 forward time for M = 1..8 at the four ChatGLM2-6B layer shapes; QLINEAR_GEMV_MAX_ROWS selects
where the GEMV kernel hands over to the split-K MFMA GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
print("GEMV_MAX_ROWS =", os.environ.get("QLINEAR_GEMV_MAX_ROWS", "4"))
for K, N, NL in ((4096, 4608, 24), (4096, 4096, 24), (4096, 27392, 6), (13696, 4096, 10)):
    layers = [bench_extras._w4_layer(torch, dev, K, N, False, gen) for _ in range(NL)]
    row = []
    for M in (1, 2, 3, 4, 5, 8, 16):
        a = torch.randn(M, K, device=dev, dtype=torch.float16)
        def f():
            with torch.no_grad():
                for l in layers:
                    l(a)
        row.append(f"M{M}:{bench_extras._graph_time(torch, dev, f) / NL * 1e3:.1f}")
    print(f"{K}->{N}: " + "  ".join(row), flush=True)
    del layers
