#!/usr/bin/env python3
"""Reader of the QL_G256_STAMPS build of w4_gemm256.hip (tools/ab/build_g256_variant.sh stamps -DQL_G256_STAMPS): where the K loop of
the 256 x 256 int4 GEMM parks - cycles in the W wait, the A wait and the barrier per wave, against the whole loop."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_extras  # noqa: E402
from chatglm_q_amd import _lib  # noqa: E402

M, K, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (8192, 4096, 4096)))
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
layers = [bench_extras._w4_layer(torch, dev, K, N, False, gen) for _ in range(8)]
x = torch.randn(M, K, device=dev, dtype=torch.float16)
with torch.no_grad():
    for _ in range(3):
        for l in layers:
            l(x)
torch.cuda.synchronize()
lib = _lib.get_lib()
lib.qlinear_g256_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
blocks = min(((N + 255) // 256) * ((M + 255) // 256), 8192)
buf = np.zeros((blocks, 2, 4), dtype=np.uint64)
assert lib.qlinear_g256_stamps_read(buf.ctypes.data, blocks) == 0
b = buf.astype(np.float64)
ksteps = K // 64
print(f"{M}x{K}x{N}: {blocks} blocks, {ksteps} K tiles; shader-clock cycles per K tile, mean over blocks (wave 0 / wave 4)")
print(f"  K loop, per K tile   {b[:, 0, 3].mean() / ksteps:8.0f} {b[:, 1, 3].mean() / ksteps:8.0f}")
print(f"  per block: prologue {b[:, 0, 2].mean():8.0f}  K loop {b[:, 0, 3].mean():8.0f}  epilogue (issue) {b[:, 0, 1].mean():8.0f} cycles")
