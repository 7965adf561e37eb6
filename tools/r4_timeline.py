#!/usr/bin/env python3
"""Reader of the QL_R4_STAMPS build of w4_gemm256.hip (tools/ab/build_g256_variant.sh r4s -DQL_R4_STAMPS): per block of the ring-of-4
int4g32 GEMM the wall time (s_memrealtime, 100 MHz) and the shader cycles of prologue / K loop / epilogue, and per CU the gap between
one block's end and the next block's start.   QLINEAR_LIB_PATH=tools/microbench/libql_g256_r4s.so python tools/r4_timeline.py [M K N]"""
import collections
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
g = torch.Generator(device=dev).manual_seed(3)
layers = [bench_extras._w4_layer(torch, dev, K, N, False, g) for _ in range(4)]
x = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
with torch.no_grad():
    for _ in range(3):
        for l in layers:
            l(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for l in layers:
        l(x)
    e1.record()
    torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / len(layers)
lib = _lib.get_lib()
lib.qlinear_r4_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
blocks = min(((N + 255) // 256) * ((M + 255) // 256), 16384)
buf = np.zeros((blocks, 12), dtype=np.uint64)
assert lib.qlinear_r4_stamps_read(buf.ctypes.data, blocks) == 0
rt = buf[:, 0:4].astype(np.float64) * 0.01           # us
ck = buf[:, 4:8].astype(np.float64)
t0 = rt[:, 0].min()
names = ["prologue", "K loop", "epilogue"]
print(f"{M}x{K}x{N}: {us:.1f} us per launch (eager), {blocks} blocks; launch span by stamps {rt[:, 3].max() - t0:.1f} us")
for i, n in enumerate(names):
    d, c = rt[:, i + 1] - rt[:, i], ck[:, i + 1] - ck[:, i]
    print(f"  {n:9s}: {d.mean():7.2f} us (p10 {np.percentile(d, 10):.2f}, p90 {np.percentile(d, 90):.2f}), {c.mean():9.0f} cycles -> {c.mean() / max(d.mean(), 1e-9) / 1e3:.2f} GHz")
steps = K // 32
loop_c = (ck[:, 2] - ck[:, 1]).mean()
print(f"  K loop: {loop_c / steps:.0f} cycles per 32-deep step ({steps} steps; MFMA issue alone: 1024 per SIMD)")
print(f"  waits of wave 0 per step (QL_R4_STAMPS=2 builds): vmcnt {buf[:, 10].astype(np.float64).mean() / steps:.0f}, barrier {buf[:, 11].astype(np.float64).mean() / steps:.0f} cycles")
per_cu = collections.defaultdict(list)
for b in range(blocks):
    per_cu[(int(buf[b, 9]), int(buf[b, 8]) & 0xFFFFF0)].append((rt[b, 0], rt[b, 3]))   # XCC id, HW_ID without the wave slot
gaps, firsts = [], []
for v in per_cu.values():
    v.sort()
    firsts.append(v[0][0] - t0)
    gaps += [b[0] - a[1] for a, b in zip(v, v[1:])]
print(f"  {len(per_cu)} distinct CU ids; first block start after launch: mean {np.mean(firsts):.2f} us, max {np.max(firsts):.2f}; "
      f"gap between blocks on a CU: mean {np.mean(gaps) if gaps else float('nan'):.2f} us, p90 {np.percentile(gaps, 90) if gaps else float('nan'):.2f}")
ends = np.sort(rt[:, 3] - t0)
print(f"  block ends: first {ends[0]:.1f}, median {np.median(ends):.1f}, last {ends[-1]:.1f} us")
