"""Developer tool (round 4): the vendor's plain GEMMs beside this library's many-row kernels, for rocprofv3.

  python tools/vendor_probe.py [i8|f16|ours_i8|ours_w4|all]

Each leg launches 6 GEMMs at 8192 x 4096 x 4096 on random operands (the bench_extras protocol's shape); run under
`rocprofv3 --kernel-trace --stats` for kernel names / LDS / registers / grid, or under tools/prof_pmc.sh for counters.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

dev = torch.device("cuda:0")
what = sys.argv[1] if len(sys.argv) > 1 else "all"
M, K, N = 8192, 4096, 4096
g = torch.Generator(device=dev).manual_seed(5)
reps = int(os.environ.get("REPS", "6"))

if what in ("i8", "all"):
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
    w = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
    for _ in range(reps):
        torch._int_mm(a, w.t())
    torch.cuda.synchronize()

if what in ("f16", "all"):
    a = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
    w = torch.randn(N, K, device=dev, dtype=torch.float16, generator=g)
    for _ in range(reps):
        a @ w.t()
    torch.cuda.synchronize()

if what in ("ours_i8", "all"):
    from chatglm_q_amd.int8 import hip_ops as h8
    t = h8.tile_w8(torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g))
    sc = (torch.rand(N, device=dev, generator=g) * 0.01 + 0.001).half()
    a_q, a_s = h8.act_quant_rowwise(torch.randn(M, K, device=dev, dtype=torch.float16, generator=g))
    for _ in range(reps):
        h8.w8a8_gemm256(a_q, a_s, t, N, sc)
    torch.cuda.synchronize()

if what in ("ours_w4", "all"):
    import bench_extras
    layer = bench_extras._w4_layer(torch, dev, K, N, False, g)
    x = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
    with torch.no_grad():
        for _ in range(reps):
            layer(x)
    torch.cuda.synchronize()
