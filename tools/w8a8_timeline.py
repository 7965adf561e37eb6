#!/usr/bin/env python3
"""Reader of the QL_W8A8_STAMPS build (tools/w8a8_timeline.sh): per-block timestamps of one GEMM launch, printed as
microseconds after the earliest block start: min / median / max over the blocks."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatglm_q_amd import _lib  # noqa: E402
from chatglm_q_amd.int8 import hip_ops as h8  # noqa: E402

NAMES = ["wave start", "setup done, prologue loads issued", "first A tiles in LDS (barrier)", "main loop done", "K loop done",
         "K-parity groups combined", "output stored", "address setup done (first load issues)"]


def main():
    M, K, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (512, 4096, 4096)))
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    tiled = [h8.tile_w8(torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)) for _ in range(12)]
    sc = (torch.rand(N, device=dev, generator=g) * 0.01 + 0.001).half()
    a_q, a_s = h8.act_quant_rowwise(torch.randn(M, K, device=dev, dtype=torch.float16))
    lib = _lib.get_dev_lib() if os.environ.get("SPLITK") == "1" else _lib.get_lib()
    lib.qlinear_w8a8_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    mt = 2 if M <= 512 else 4
    blocks = ((N + 127) // 128) * ((M + 32 * mt - 1) // (32 * mt))
    splitk = os.environ.get("SPLITK") == "1"                 # round 6 experiment: needs a stamps build WITH -DQL_DEV_EXPERIMENTS as the dev library
    if splitk:
        from chatglm_q_amd.dev import experiments as X
        gemm = X.w8a8_gemm_tiled_splitk
        blocks = (M // 128) * (N // 128) * 2
    else:
        gemm = h8.w8a8_gemm_tiled
    blocks = min(blocks, 4096)
    rows = []
    for rep in range(6):
        for t in tiled[:-1]:
            gemm(a_q, a_s, t, N, sc)                        # keep the chip busy and clocked up
        gemm(a_q, a_s, tiled[-1], N, sc)                    # the launch that is read (stamps of the LAST launch survive)
        torch.cuda.synchronize()
        buf = np.zeros((blocks, 8), dtype=np.uint64)
        assert lib.qlinear_w8a8_stamps_read(buf.ctypes.data, blocks) == 0
        t = buf[:, :8].astype(np.int64)
        t0 = t[:, 0].min()
        rows.append((t - t0) * 0.01)
    r = np.median(np.stack(rows), axis=0)                   # median over repetitions, per block and point
    print(f"{M}x{K}x{N}: {blocks} blocks; microseconds after the earliest block start (min / median / max over blocks)")
    for i in (0, 7, 1, 2, 3, 4, 5, 6):
        col = r[:, i]
        print(f"  {NAMES[i]:38s} {col.min():7.2f} {np.median(col):7.2f} {col.max():7.2f}")
    if splitk:
        # publishers (even ticket) leave right after their flag store, finishers wait for it: split the last two points by role
        # (a finisher's "output stored" is later than its partner's; blocks are paired by position, not by id: compare the distributions)
        d = r[:, 6] - r[:, 5]
        order = np.argsort(d)
        half = len(d) // 2
        print(f"  hand-off + epilogue (point 6 - point 5): shorter half (publishers) median {np.median(d[order[:half]]):.2f} us, "
              f"longer half (finishers) median {np.median(d[order[half:]]):.2f} us")


if __name__ == "__main__":
    main()
