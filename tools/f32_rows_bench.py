"""Developer tool (round 5): fp32 activations at many rows - the fp32 matrix-instruction kernel (wq_gemm_f32.hip) against the VALU kernels
that served those calls before (QLINEAR_DISPATCH=nof32mfma), int4g32 and int8 per channel.  python tools/f32_rows_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench_extras import _graph_time  # noqa: E402
from chatglm_q_amd import _lib  # noqa: E402
from chatglm_q_amd.int4 import hip_ops as h4  # noqa: E402
from chatglm_q_amd.int8 import hip_ops as h8  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.get_lib()
g = torch.Generator(device=dev).manual_seed(3)
for M, K, N in [(64, 4096, 4096), (96, 4096, 4096), (128, 4096, 4096), (256, 4096, 4096), (512, 4096, 4096), (2048, 4096, 4096), (512, 4096, 27392), (512, 13696, 4096)]:
    a = torch.randn(M, K, device=dev, generator=g)
    qw = torch.randint(0, 256, (K // 2, N), dtype=torch.uint8, device=dev, generator=g)
    sc = torch.rand(K // 32, N, device=dev, generator=g) * 0.02 + 0.002
    w8 = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
    s8 = torch.rand(N, device=dev, generator=g) * 0.01 + 0.001
    row = [f"{M:5d} x {K:5d} x {N:5d}:"]
    for tag, env in (("mfma", ""), ("valu", "nof32mfma")):
        if env:
            os.environ["QLINEAR_DISPATCH"] = env
        else:
            os.environ.pop("QLINEAR_DISPATCH", None)
        lib.qlinear_dispatch_reload()
        us4 = _graph_time(torch, dev, lambda: h4.w4_forward(a, qw, sc)) * 1e3
        us8 = _graph_time(torch, dev, lambda: h8.w8_forward(a, w8.t(), s8)) * 1e3
        row.append(f"{tag}: int4 {us4:9.1f} us {2.0 * M * N * K / us4 / 1e6:6.1f} TF | int8 {us8:9.1f} us {2.0 * M * N * K / us8 / 1e6:6.1f} TF |")
    print(" ".join(row), flush=True)
os.environ.pop("QLINEAR_DISPATCH", None)
lib.qlinear_dispatch_reload()
