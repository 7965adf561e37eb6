#!/usr/bin/env python3
"""Reader of the QL_I256_STAMPS build of w8a8_gemm256.hip: shader cycles per K tile of the 256 x 256 int8 GEMM's loop."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatglm_q_amd import _lib  # noqa: E402
from chatglm_q_amd.int8 import hip_ops as h8  # noqa: E402
M, K, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (8192, 4096, 4096)))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
tiled = [h8.tile_w8(torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)) for _ in range(4)]
sc = (torch.rand(N, device=dev, generator=g) * 0.01 + 0.001).half()
a_q, a_s = h8.act_quant_rowwise(torch.randn(M, K, device=dev, dtype=torch.float16))
for _ in range(5):
    for t in tiled:
        h8.w8a8_gemm_tiled(a_q, a_s, t, N, sc)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for t in tiled:
    h8.w8a8_gemm_tiled(a_q, a_s, t, N, sc)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / len(tiled)
lib = _lib.get_lib()
lib.qlinear_i256_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
blocks = min(((N + 255) // 256) * ((M + 255) // 256), 8192)
buf = np.zeros((blocks, 2, 4), dtype=np.uint64)
assert lib.qlinear_i256_stamps_read(buf.ctypes.data, blocks) == 0
b = buf.astype(np.float64)
kt = K // 128
loop = b[:, 0, 3].mean()
rounds = -(-blocks // 256)
print(f"waits per K tile (wave 0 / wave NWN): vmcnt {b[:, 0, 0].mean() / kt:.0f} / {b[:, 1, 0].mean() / kt:.0f}, barrier {b[:, 0, 1].mean() / kt:.0f} / {b[:, 1, 1].mean() / kt:.0f} cycles"
      f" (QL_I256_STAMPS=2 builds; 0 otherwise)")
t0 = b[:, 0, 2]
order = np.argsort(t0)
print("block start spread (cycles, sorted starts, every 32nd):", [int(t0[i] - t0[order[0]]) for i in order[::32]][:20])
print(f"{M}x{K}x{N}: {us:.1f} us per launch; K loop {loop / kt:.0f} cycles per K tile (pure MFMA issue: 2048), {loop:.0f} per block; "
      f"{rounds} rounds -> implied clock >= {rounds * loop / us / 1e3:.2f} GHz")
