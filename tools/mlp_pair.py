#!/usr/bin/env python3
"""The MLP of a one-row decode step (ChatGLM2-6B dims) as two launches and as ONE launch (qlinear_w4g32_mlp_pair): bit
equality and time per layer, rotating weight sets, one HIP graph."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_extras import _graph_time, _w4_layer  # noqa: E402
from chatglm_q_amd import _lib  # noqa: E402
from chatglm_q_amd.int4 import hip_ops as H4  # noqa: E402
from chatglm_q_amd.dev import experiments as X  # noqa: E402  (recorded experiments: developer library)

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(3)
K, HID, NL = 4096, 13696, 8
dtype = torch.float16
ins = [_w4_layer(torch, dev, K, 2 * HID, False, gen) for _ in range(NL)]
outs = [_w4_layer(torch, dev, HID, K, False, gen) for _ in range(NL)]
gated = [l.gated_packed(HID) for l in ins]
ln = (1 + 0.1 * torch.randn(K, device=dev, generator=gen)).to(dtype)
h = torch.randn(1, 1, K, device=dev, generator=gen).to(dtype)


def two(i, x):
    gp, gb = gated[i]
    y = H4.w4_forward_fused(_lib.PRO_ADDNORM | _lib.EPI_SILU_GATE, x, gp, 2 * HID, gb, None, ln, None, 1e-5)
    return H4.w4_forward_residual(y, outs[i].prepare()._packed, K, None, x)


def one(i, x):
    gp, gb = gated[i]
    return X.w4_mlp_pair(x, ln, 1e-5, gp, gb, 2 * HID, outs[i].prepare()._packed, None, K, x)


for i in range(0 if not os.environ.get("PAIR_NOCHECK") else NL, NL):
    a, b = two(i, h), one(i, h)
    assert b is not None, "pair not served"
    torch.cuda.synchronize()
    assert not X.mlp_pair_timed_out(dev), "a consumer gave up waiting"
    assert torch.equal(a, b), (i, (a.float() - b.float()).abs().max().item())
print("bit-equal on", NL, "weight sets")


def chain(f):
    def run():
        x = h
        for r in range(3):
            for i in range(NL):
                x = f(i, x) * 0 + h                       # keep the value bounded; the launches stay dependent
    return run


def chain_plain(f):
    def run():
        for r in range(3):
            for i in range(NL):
                f(i, h)
    return run


for name, f in (("two launches", two), ("one launch", one)):
    ms = _graph_time(torch, dev, chain_plain(f))
    print(f"{name}: {ms / (3 * NL) * 1e3:.2f} us per MLP")
torch.cuda.synchronize()
print("timed out:", X.mlp_pair_timed_out(dev))
