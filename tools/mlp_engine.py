#!/usr/bin/env python3
"""The MLP of a one-row decode step (ChatGLM2-6B dims, int4g32) as two fused launches and as ONE persistent launch
(qlinear_w4g32_mlp_engine: LDS-DMA loader wave + 7 consumer waves per CU, granule hand-off): bit equality, time per MLP over
rotating weight sets replayed from one HIP graph, and - with the trace build (make -C chatglm_q_amd/csrc trace;
QLINEAR_LIB_PATH=.../libqlinear_hip_trace.so) - the timeline of one launch from per-workgroup s_memrealtime stamps.

    python tools/mlp_engine.py [f16|bf16] [--no-check]
"""
import ctypes
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
dtype = torch.bfloat16 if "bf16" in sys.argv else torch.float16
ins = [_w4_layer(torch, dev, K, 2 * HID, False, gen) for _ in range(NL)]
outs = [_w4_layer(torch, dev, HID, K, False, gen) for _ in range(NL)]
if dtype != torch.float16:
    ins = [l.to(dtype) for l in ins]
    outs = [l.to(dtype) for l in outs]
gated = [l.gated_packed(HID) for l in ins]
packed_out = [l.prepare()._packed for l in outs]
ln = (1 + 0.1 * torch.randn(K, device=dev, generator=gen)).to(dtype)
h = torch.randn(1, 1, K, device=dev, generator=gen).to(dtype)
ws = X.mlp_engine_workspace(2 * HID, dev)
print("engine supported:", X.mlp_engine_supported(2 * HID, K, K), "| dtype", dtype, "| strict", _lib.strict_for(dtype))


def two(i, x):
    gp, gb = gated[i]
    y = H4.w4_forward_fused(_lib.PRO_ADDNORM | _lib.EPI_SILU_GATE, x, gp, 2 * HID, gb, None, ln, None, 1e-5)
    return H4.w4_forward_residual(y, packed_out[i], K, None, x)


def one(i, x):
    gp, gb = gated[i]
    return X.w4_mlp_engine(x, ln, 1e-5, gp, gb, 2 * HID, packed_out[i], None, K, ws)


if "--no-check" not in sys.argv:
    for rep in range(3):
        for i in range(NL):
            x = torch.randn(1, 1, K, device=dev, generator=gen).to(dtype)
            a, b = two(i, x), one(i, x)
            assert b is not None, "engine did not serve the shape"
            torch.cuda.synchronize()
            assert X.mlp_engine_error(ws) == 0, f"a bounded wait gave up (code {X.mlp_engine_error(ws)})"
            assert torch.equal(a, b), (rep, i, (a.float() - b.float()).abs().max().item(), int((a != b).sum()))
    print("bit-equal on", NL, "weight sets x 3 inputs")


def chain_plain(f):
    def run():
        for r in range(3):
            for i in range(NL):
                f(i, h)
    return run


for name, f in (("two launches", two), ("one persistent launch", one), ("two launches", two), ("one persistent launch", one)):
    ms = _graph_time(torch, dev, chain_plain(f))
    print(f"{name}: {ms / (3 * NL) * 1e3:.2f} us per MLP")
torch.cuda.synchronize()
print("error word:", X.mlp_engine_error(ws))

# ---- timeline (trace build only) -------------------------------------------------------------------------------------------
lib = _lib.get_lib()
if hasattr(lib, "qlinear_w4g32_mlp_engine_trace"):
    fn = lib.qlinear_w4g32_mlp_engine_trace
    fn.restype = ctypes.c_int
    from ctypes import c_float, c_int, c_int64, c_void_p
    fn.argtypes = [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int,
                   c_int, c_void_p, c_void_p]
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    trace = torch.zeros(ncu * 16, dtype=torch.int64, device=dev)
    out = torch.empty(1, 1, K, device=dev, dtype=dtype)
    rows = []
    for rep in range(6):
        i = rep % NL
        gp, gb = gated[i]
        trace.zero_()
        for _ in range(2):                           # the stamped launch runs behind another one (not from an idle chip)
            st = fn(h.data_ptr(), ln.data_ptr(), 1e-5, gp.data_ptr(), None, 2 * HID, packed_out[i].data_ptr(), None, K, K, out.data_ptr(),
                    ws.data_ptr(), _lib.dtype_code(dtype), _lib.FLAG_STRICT_ROUNDING if _lib.strict_for(dtype) else 0, trace.data_ptr(),
                    _lib.stream_ptr(dev))
            assert st == 0, st
        torch.cuda.synchronize()
        t = trace.view(ncu, 16).cpu().double()
        t0 = t[:, 10].min()                          # first consumer 0 past the workgroup barrier
        us = (t - t0) / 100.0                         # 100 MHz counter
        names = ["loader: last task issued", "loader done (all tasks landed)", "x row normalised", "phase A done (consumer 0)",
                 "gather done (all 7 consumers)", "phase B done (consumer 0)", "end", "phase A done (last consumer)",
                 "consumer 0's own sweep complete"]
        rows.append([(us[:, k].min().item(), us[:, k].median().item(), us[:, k].max().item()) for k in range(9)])
        sweeps = t[:, 9]
    print("timeline of one launch, us after the first workgroup passed its start barrier: min / median / max over the workgroups (last of 6 launches)")
    for k, nm in enumerate(names):
        lo, md, hi = rows[-1][k]
        print(f"  {nm:34s} {lo:7.2f} {md:7.2f} {hi:7.2f}")
    for k, nm in ((11, "consumer 0: waiting for landed slots, phase A"), (12, "consumer 0: waiting for landed slots, phase B"),
                  (15, "loader: blocked on a full ring")):
        v = t[:, k] / 100.0
        print(f"  {nm:46s} {v.min().item():7.2f} {v.median().item():7.2f} {v.max().item():7.2f} us")
    print(f"  tasks run by consumer 0: phase A median {t[:, 13].median().item():.0f} (of {27}), phase B median {t[:, 14].median().item():.0f} (of 16)")
    print(f"  sweeps of consumer 0 until every tag matched: min {sweeps.min().item():.0f} median {sweeps.median().item():.0f} max {sweeps.max().item():.0f}")
