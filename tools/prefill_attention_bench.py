"""One-launch prefill attention (qlinear_prefill_attention) against the GEMM route (two batched GEMMs around
qlinear_masked_softmax, chatglm_q_amd.model.ChatGLM2Attention.core) on config 5's chunks: batch 4, chunks of CH positions of a
2048-token prompt.  Prints ms per layer-chunk and effective TFLOP/s (causal FLOPs: 4 * B * H * D * visible (query, key) pairs).
    gpurun -- 'python tools/prefill_attention_bench.py'"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatglm_q_amd import fused_ops as F_          # noqa: E402
from chatglm_q_amd import model as M                # noqa: E402

H, G, D, B, SEQ = 32, 2, 128, 4, 2048
CH = int(os.environ.get("CHUNK", 1024))
dev = "cuda"


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dt = torch.float16
    attn = M.ChatGLM2Attention(H * D, H, D, G, 0, dtype=dt)
    k = torch.randn(B, SEQ, G, D, device=dev).to(dt)
    v = torch.randn(B, SEQ, G, D, device=dev).to(dt)
    t = torch.arange(SEQ, device=dev)
    total_old = total_new = 0.0
    for s0 in range(0, SEQ, CH):
        S, T = CH, s0 + CH
        q = torch.randn(B, S, H * D, device=dev).to(dt)
        rows = torch.arange(s0, s0 + S, device=dev)
        mask = ((t[None, None, :T] > rows[None, :, None]).expand(B, S, T).float() * -1e10).contiguous()
        flags = F_.attention_tile_flags(mask)
        pairs = sum(min(r + 1, T) for r in range(s0, s0 + S))
        flops = 4.0 * B * H * D * pairs
        old = timed(lambda: attn.core(dt, q.view(B, S, G, H // G, D), k[:, :T], v[:, :T], mask))
        new = timed(lambda: F_.prefill_attention(q, k, v, mask, flags, T, H, G, D))
        noflag = timed(lambda: F_.prefill_attention(q, k, v, mask, None, T, H, G, D))
        fl = timed(lambda: F_.attention_tile_flags(mask))
        a = attn.core(dt, q.view(B, S, G, H // G, D), k[:, :T], v[:, :T], mask)
        b = F_.prefill_attention(q, k, v, mask, flags, T, H, G, D)
        rel = float((a.float() - b.float()).norm() / a.float().norm())
        print(f"chunk rows {s0}..{s0 + S} vs {T} keys: GEMM route {old:.3f} ms ({flops / old / 1e9:.0f} TFLOP/s causal-effective), "
              f"one launch {new:.3f} ms ({flops / new / 1e9:.0f}), without tile flags {noflag:.3f} ms, flags {fl:.3f} ms once per chunk; "
              f"rel-L2 between the routes {rel:.2e}")
        total_old += old
        total_new += new
    print(f"per layer (all chunks): {total_old:.3f} -> {total_new:.3f} ms; x 28 layers: {28 * total_old:.1f} -> {28 * total_new:.1f} ms")


if __name__ == "__main__":
    main()
