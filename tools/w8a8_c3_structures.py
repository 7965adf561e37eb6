"""Developer tool (round 5): BASELINE config 3 (int8 x int8, 512 x 4096 -> 4096) on the round-4 ring structure.

What a K split of the 256 x 256-tile ring kernel could reach, bounded from below WITHOUT writing it:
  (a) the product's GEMM (w8a8_tiled_kernel: 64-row tiles, 256 workgroups, no split), weights rotated through HBM;
  (b) the ring kernel itself at 512 rows (qlinear_w8a8_fwd_tiled256: 32 tiles = 32 of 256 CUs, K = 4096 each);
  (c) PROXIES of a split by S: the ring kernel on (512 S) x (4096 / S) x 4096 - the same number of workgroups (32 S), the same K
      extent per workgroup, the same MFMA work in total and 16-bit outputs of (512 S) x 4096 x 2 bytes, i.e. HALF the bytes the S int32
      partial tiles of a real split would write (and none of their second pass) and an S-times smaller weight set: a LOWER bound of
      what the main phase of a split-S kernel costs.
  python tools/w8a8_c3_structures.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench_extras import _graph_time  # noqa: E402
from chatglm_q_amd.int8 import hip_ops as h8  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(31)
N = 4096
NSETS = int(os.environ.get("NSETS", 24))        # 24 x 16 MB of weights: out of the 256 MB memory-side cache


def run(M, K, label, fn_name):
    nsets = max(4, min(NSETS * 4096 // K, 96))
    tiled = [h8.tile_w8(torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)) for _ in range(nsets)]
    sc = (torch.rand(N, device=dev, generator=g) * 0.01 + 0.001).half()
    a_q, a_s = h8.act_quant_rowwise(torch.randn(M, K, device=dev, dtype=torch.float16, generator=g))
    fn = getattr(h8, fn_name)
    us = _graph_time(torch, dev, lambda: [fn(a_q, a_s, t, N, sc) for t in tiled]) / nsets * 1e3
    ops = 2.0 * M * N * K
    print(f"{label:62s} {M:5d} x {K:5d} x {N}: {us:7.2f} us  {ops / us / 1e6:6.0f} TOP/s   ({nsets} weight sets)", flush=True)
    del tiled
    torch.cuda.empty_cache()
    return us


base = run(512, 4096, "(a) product GEMM (64-row tiles, 256 workgroups)", "w8a8_gemm_tiled")
ring = run(512, 4096, "(b) ring kernel, 256 x 256 tiles, no split (32 workgroups)", "w8a8_gemm256")
for S in (2, 4, 8):
    us = run(512 * S, 4096 // S, f"(c) proxy of a split by {S}: {32 * S} workgroups, K = {4096 // S} each", "w8a8_gemm256")
    part = S * 512 * N * 4
    print(f"      + what the proxy leaves out: {part / 1e6:.0f} MB of int32 partial tiles written and read once more "
          f"(>= {2 * part / 4.5e12 * 1e6:.1f} us at the 4.5 TB/s this box copies at) -> split-{S} >= {us + 2 * part / 4.5e12 * 1e6:.1f} us")
print(f"product {base:.2f} us; the ring structure without a split {ring:.2f} us")
