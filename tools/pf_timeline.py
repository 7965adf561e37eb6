#!/usr/bin/env python3
"""Reader of the QL_PF_STAMPS build of prefill_attention.hip (tools/ab/build_pf_variant.sh stamps -DQL_PF_STAMPS; run with
QLINEAR_LIB_PATH=tools/microbench/libql_pf_stamps.so): shader-clock ticks per slot of workgroup 0 (the longest block of sequence 0,
group 0) - arithmetic, staging stores, barrier wait - for a wave of group A (0) and its SIMD partner of group B (4)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatglm_q_amd import _lib, fused_ops as F_  # noqa: E402

H, G, D, B = 32, 2, 128, 4
S, T = 1024, 2048
dev = "cuda"
q = torch.randn(B, S, H * D, device=dev).half()
k = torch.randn(B, T, G, D, device=dev).half()
v = torch.randn(B, T, G, D, device=dev).half()
t = torch.arange(T, device=dev)
rows = torch.arange(T - S, T, device=dev)
mask = ((t[None, None, :] > rows[None, :, None]).expand(B, S, T).float() * -1e10).contiguous()
flags = F_.attention_tile_flags(mask)
for _ in range(3):
    F_.prefill_attention(q, k, v, mask, flags, T, H, G, D)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
F_.prefill_attention(q, k, v, mask, flags, T, H, G, D)
e1.record()
torch.cuda.synchronize()
lib = _lib.get_lib()
lib.qlinear_pf_stamps_read.argtypes = [ctypes.c_void_p]
buf = np.zeros((8, 160, 4), dtype=np.uint64)
assert lib.qlinear_pf_stamps_read(buf.ctypes.data) == 0
b = buf.astype(np.int64)
nslots = int((b[0, :, 0] > 0).sum())
print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us; workgroup 0: {nslots} slots; ticks from slot start: arithmetic done / stores done / behind the barrier")
t00 = b[0, 0, 0]
for w in (0, 4):
    print(f"wave {w}:")
    for p in range(min(nslots, 12)):
        st = b[w, p]
        print(f"  slot {p:3d} start {st[0] - t00:7d}  +{st[1] - st[0]:6d} +{st[2] - st[1]:6d} +{st[3] - st[2]:6d}   slot total {st[3] - st[0]:6d}")
    tot = b[w, nslots - 1, 3] - b[w, 0, 0]
    ar = (b[w, :nslots, 1] - b[w, :nslots, 0]).sum()
    sto = (b[w, :nslots, 2] - b[w, :nslots, 1]).sum()
    bar = (b[w, :nslots, 3] - b[w, :nslots, 2]).sum()
    print(f"  all slots: {tot} ticks = arithmetic {ar} + stores {sto} + barrier wait {bar}; per tile {tot / (nslots / 2):.0f}")
    even = (b[w, 0:nslots:2, 1] - b[w, 0:nslots:2, 0])
    odd = (b[w, 1:nslots:2, 1] - b[w, 1:nslots:2, 0])
    print(f"  arithmetic per even slot {even.mean():.0f}, per odd slot {odd.mean():.0f}")
