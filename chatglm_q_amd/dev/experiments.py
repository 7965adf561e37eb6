"""Recorded experiments (libqlinear_hip_dev.so, include/qlinear_hip_dev.h) - NOT part of the product path.

Each of these was built to the product's conventions, measured on MI355X, lost to what the default dispatch does, and is kept
with its parity test (pytest marker ``dev``) so that the measurement can be repeated (LABNOTES.md has the numbers):

* ``w4_mlp_pair``      round 2: both MLP projections of a one-row decode step in one launch (chained grids): 27.5 vs 23.0 us
* ``w4_mlp_engine``    round 3: the same as ONE persistent launch on an LDS-DMA ring: 32.0 vs 23.0 us
* ``w4a8_*``           int4g32 weights x int8 activations on the i8 matrix cores: <= half the i8 rate by construction

``enable_mlp_engine()`` / ``enable_mlp_pair()`` plug the one-launch MLPs into ``ChatGLM2Model.decode_step`` through its
``MLP_HOOK``; ``enable_w4a8(module)`` switches an int4 module to the W4A8 path.  Everything here needs the developer library
(``make -C chatglm_q_amd/csrc dev``) and raises ``QLinearLibraryMissing`` without it.
"""
from __future__ import annotations

import torch
from torch import Tensor

from .. import _lib
from ..int4 import hip_ops as H4


def w4a8_supported(a: Tensor, b: Tensor, b_scale: Tensor) -> bool:
    """Shapes / dtypes served by the int8-activation path (qlinear_w4a8_*): fp16 / bf16, group 32."""
    return (a.is_cuda and a.dtype in (torch.float16, torch.bfloat16) and b_scale.dtype == a.dtype and b.dtype == torch.uint8
            and b_scale.shape[0] * 32 == b.shape[0] * 2 and b.is_contiguous() and b_scale.is_contiguous())


def pack_w4a8(b: Tensor, b_scale: Tensor) -> Tensor:
    """Derived "a8" layout of canonical int4g32 buffers for the int8-activation GEMM (``qlinear_w4a8_pack``): a cache like
    ``repack_w4g32``'s, never part of a state_dict."""
    lib = _lib.get_dev_lib()
    K, N = b.shape[0] * 2, b.shape[1]
    nbytes = int(lib.qlinear_w4a8_packed_bytes(N, K, 32, _lib.dtype_code(b_scale.dtype)))
    if nbytes == 0:
        raise ValueError(f"no W4A8 layout for N={N}, K={K}, dtype={b_scale.dtype} (group 32, fp16 / bf16)")
    out = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
    with torch.cuda.device(b.device):
        st = lib.qlinear_w4a8_pack(b.contiguous().data_ptr(), b_scale.contiguous().data_ptr(), out.data_ptr(), N, K, 32,
                                   _lib.dtype_code(b_scale.dtype), _lib.stream_ptr(b.device))
    _lib.check(st, "qlinear_w4a8_pack")
    return out


def w4a8_forward(a: Tensor, packed_a8: Tensor, n_out: int, bias: Tensor | None = None, per_tensor: bool = False) -> Tensor:
    """int8-quantised activations x int4g32 weights on the i8 matrix cores (``qlinear_w4a8_linear``: activation
    quantiser + GEMM in one library call).  NOT bit-compatible with the weight-only path: see include/qlinear_hip.h."""
    lib = _lib.get_dev_lib()
    a2 = H4._rows(a)
    M, K = a2.shape
    H4._check_row_operands("w4a8_forward", a, K, bias=bias)
    if packed_a8.device != a.device or packed_a8.numel() != int(lib.qlinear_w4a8_packed_bytes(n_out, K, 32, _lib.dtype_code(a.dtype))):
        raise AssertionError(f"w4a8_forward: packed buffer does not belong to a ({K}, {n_out}) {a.dtype} weight on {a.device}")
    c = torch.empty((M, n_out), device=a.device, dtype=a.dtype)
    if M:
        if bias is not None:
            bias = bias.contiguous()
        with torch.cuda.device(a.device):
            ws_bytes = int(lib.qlinear_workspace_bytes(_lib.OP_W4A8_LINEAR, M, n_out, K, 32))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device)
            st = lib.qlinear_w4a8_linear(a2.data_ptr(), packed_a8.data_ptr(), _lib.ptr(bias), c.data_ptr(), M, n_out, K,
                                         a2.stride(0) if M > 1 else K, n_out, _lib.dtype_code(a.dtype),
                                         _lib.FLAG_ACT_PER_TENSOR if per_tensor else 0, ws.data_ptr(), ws_bytes,
                                         _lib.stream_ptr(a.device))
        _lib.check(st, "qlinear_w4a8_linear")
    return c.reshape(*a.shape[:-1], n_out)


def w4a8_gemm(a_q: Tensor, a_s: Tensor, packed_a8: Tensor, n_out: int, dtype: torch.dtype, bias: Tensor | None = None) -> Tensor:
    """Second step alone (``qlinear_w4a8_fwd``): pre-quantised int8 rows (M, K) and their fp32 scales."""
    lib = _lib.get_dev_lib()
    M, K = a_q.shape
    c = torch.empty((M, n_out), device=a_q.device, dtype=dtype)
    if M:
        with torch.cuda.device(a_q.device):
            st = lib.qlinear_w4a8_fwd(a_q.data_ptr(), a_s.data_ptr(), packed_a8.data_ptr(), _lib.ptr(bias), c.data_ptr(), M, n_out,
                                      K, n_out, _lib.dtype_code(dtype), _lib.stream_ptr(a_q.device))
        _lib.check(st, "qlinear_w4a8_fwd")
    return c


def mlp_engine_supported(n_in: int, K: int, n_out: int) -> bool:
    """True when ``qlinear_w4g32_mlp_engine`` serves a (K -> n_in -> n_out) gated MLP (one persistent launch)."""
    return bool(_lib.get_dev_lib().qlinear_w4g32_mlp_engine_supported(n_in, K, n_out))


def mlp_engine_workspace(n_in: int, device) -> Tensor:
    """Zeroed workspace of the persistent MLP launch (launch epoch, error word, hand-off granules): one per decode session /
    stream, reused by every layer's launch of that session (the launches of a stream are ordered)."""
    nbytes = int(_lib.get_dev_lib().qlinear_w4g32_mlp_engine_workspace_bytes(n_in))
    ws = torch.zeros(nbytes + 64, dtype=torch.uint8, device=device)
    off = (-ws.data_ptr()) % 64
    return ws[off: off + nbytes]


def mlp_engine_error(ws: Tensor) -> int:
    """Host-synchronous read of the workspace's error word: non-zero when a bounded wait of an earlier launch gave up."""
    return int(ws[8:12].view(torch.int32).item())


def w4_mlp_engine(x: Tensor, ln_weight: Tensor, eps: float, gated_packed: Tensor, bias_in: Tensor | None, n_in: int,
                  packed_out: Tensor, bias_out: Tensor | None, n_out: int, ws: Tensor, strict: bool | None = None,
                  plan_out: list | None = None, guards=()) -> Tensor | None:
    """The MLP of a one-row decode step in ONE persistent launch (``qlinear_w4g32_mlp_engine``): ``round(w_out(silu(h) * gate)
    + x)`` with ``(h | gate) = w_in(rmsnorm(x) * ln_weight)``; bit-equal to ``w4_forward_fused(PRO_ADDNORM | EPI_SILU_GATE)``
    followed by ``w4_forward_residual(residual=x)``.  None when the library does not serve the shape that way."""
    lib = _lib.get_dev_lib()
    K = x.shape[-1]
    if x.numel() != K:
        raise ValueError("the MLP engine serves exactly one row")
    H4._check_row_operands("w4_mlp_engine", x, K, ln_weight=ln_weight, bias_in=bias_in, bias_out=bias_out)
    if n_out != K or not mlp_engine_supported(n_in, K, n_out):
        return None
    if gated_packed.numel() < H4.gemv_nbytes(n_in, K, x.dtype) or packed_out.numel() < H4.gemv_nbytes(n_out, n_in // 2, x.dtype):
        raise AssertionError("w4_mlp_engine: derived buffers too small for the two projections")
    if ws.device != x.device or ws.data_ptr() % 64 or ws.numel() < int(lib.qlinear_w4g32_mlp_engine_workspace_bytes(n_in)):
        raise AssertionError("w4_mlp_engine: workspace on another device, misaligned or too small")
    flags = _lib.FLAG_STRICT_ROUNDING if (_lib.strict_for(x.dtype) if strict is None else strict) else 0
    x = x.contiguous()
    out = torch.empty((*x.shape[:-1], n_out), device=x.device, dtype=x.dtype)
    code = _lib.dtype_code(x.dtype)
    with torch.cuda.device(x.device):
        st = lib.qlinear_w4g32_mlp_engine(x.data_ptr(), ln_weight.data_ptr(), float(eps), gated_packed.data_ptr(), _lib.ptr(bias_in), n_in,
                                          packed_out.data_ptr(), _lib.ptr(bias_out), n_out, K, out.data_ptr(), ws.data_ptr(), code, flags,
                                          _lib.stream_ptr(x.device))
    if st == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(st, "qlinear_w4g32_mlp_engine")
    if plan_out is not None:
        plan_out.append(_lib.make_plan(
            "qlinear_w4g32_mlp_engine", (None, ln_weight.data_ptr(), float(eps), gated_packed.data_ptr(), _lib.ptr(bias_in), n_in,
                                         packed_out.data_ptr(), _lib.ptr(bias_out), n_out, K, None, ws.data_ptr(), code, flags, None),
            0, 10, 14, 1, K, n_out, x.dtype, x.device, (*guards, ln_weight, bias_in, bias_out),
            keep=(gated_packed, packed_out, bias_in, bias_out, ln_weight, ws), dev=True))
    return out


_PAIR_WS: dict = {}


def w4_mlp_pair(x: Tensor, ln_weight: Tensor, eps: float, gated_packed: Tensor, bias_in: Tensor | None, n_in: int,
                packed_out: Tensor, bias_out: Tensor | None, n_out: int, residual: Tensor) -> Tensor | None:
    """EXPERIMENT (``qlinear_w4g32_mlp_pair``): the MLP of a one-row decode step in ONE launch - RMSNorm + w_in + SiLU * gate
    and w_out + residual, the second projection's workgroups waiting inside the launch for the first one's row.  Bit-equal
    to ``w4_forward_fused(PRO_ADDNORM | EPI_SILU_GATE)`` followed by ``w4_forward_residual``.  None when the library does
    not serve the shape (the caller then issues the two launches)."""
    lib = _lib.get_dev_lib()
    K = x.shape[-1]
    if x.numel() != K or residual.numel() != n_out:
        raise ValueError("the MLP pair serves exactly one row")
    H4._check_row_operands("w4_mlp_pair", x, K, ln_weight=ln_weight, bias_in=bias_in, bias_out=bias_out, residual=residual)
    if gated_packed.numel() < H4.gemv_nbytes(n_in, K, x.dtype) or packed_out.numel() < H4.gemv_nbytes(n_out, n_in // 2, x.dtype):
        raise AssertionError("w4_mlp_pair: derived buffers too small for the two projections")
    ws_key = (x.device, torch.cuda.current_stream(x.device).cuda_stream)   # two launches in flight on two streams must not share counters
    ws = _PAIR_WS.get(ws_key)
    if ws is None:                                        # arrival counters: zeroed once, reset by the kernel itself
        ws = _PAIR_WS[ws_key] = torch.zeros(int(lib.qlinear_w4g32_mlp_pair_workspace_bytes()) + 64, dtype=torch.uint8, device=x.device)
    off = (-ws.data_ptr()) % 64
    x = x.contiguous()
    residual = residual.contiguous()
    mid = torch.empty(n_in // 2, device=x.device, dtype=x.dtype)
    out = torch.empty((*x.shape[:-1], n_out), device=x.device, dtype=x.dtype)
    with torch.cuda.device(x.device):
        st = lib.qlinear_w4g32_mlp_pair(x.data_ptr(), ln_weight.data_ptr(), float(eps), gated_packed.data_ptr(), _lib.ptr(bias_in), n_in,
                                        packed_out.data_ptr(), _lib.ptr(bias_out), n_out, K, residual.data_ptr(), mid.data_ptr(),
                                        out.data_ptr(), ws.data_ptr() + off, _lib.dtype_code(x.dtype), _lib.stream_ptr(x.device))
    if st == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(st, "qlinear_w4g32_mlp_pair")
    return out


def mlp_pair_timed_out(device) -> bool:
    """True when a consumer workgroup of an earlier ``w4_mlp_pair`` launch gave up waiting (host-synchronous read of the
    error word; the results of that launch are then garbage)."""
    hit = False
    for (dev, _stream), ws in list(_PAIR_WS.items()):
        if dev == torch.device(device):
            off = (-ws.data_ptr()) % 64
            hit |= bool(ws[off + 4 * 16 * 65: off + 4 * 16 * 65 + 4].view(torch.int32).item())
    return hit


# ---- hooks into the product's Python (no experiment code lives there) ---------------------------------------------------------
def _engine_hook(model, cache, ff, ln, x):
    w_in, w_out = ff.w_in, ff.w_out
    plan = w_in._fast.get("engine")
    if plan is not None and plan[0] is cache:
        out = plan[1](x)
        if out is not None:
            return out
    ws = cache.__dict__.get("engine_ws")
    if ws is None:
        if not mlp_engine_supported(w_in.out_features, w_in.in_features, w_out.out_features):
            return None
        ws = cache.engine_ws = mlp_engine_workspace(w_in.out_features, x.device)
    gp, gb = w_in.gated_packed(ff.hidden_dim)
    plan_out = []
    out = w4_mlp_engine(x, ln.weight, ln.eps, gp, gb, w_in.out_features, w_out.prepare()._packed, w_out.bias,
                        w_out.out_features, ws, plan_out=plan_out,
                        guards=(w_in.weight, w_in.weight_scale, w_out.weight, w_out.weight_scale))
    if out is not None and plan_out and plan_out[0] is not None:
        w_in._fast["engine"] = (cache, plan_out[0])     # the plan bakes this session's workspace in
    return out


def _pair_hook(model, cache, ff, ln, x):
    if _lib.strict_for(x.dtype):                        # the chained-grid kernel exists in the exact-dequant arithmetic only
        return None
    gp, gb = ff.w_in.gated_packed(ff.hidden_dim)
    return w4_mlp_pair(x, ln.weight, ln.eps, gp, gb, ff.w_in.out_features, ff.w_out.prepare()._packed, ff.w_out.bias,
                       ff.w_out.out_features, x)


def _error_check(sess):
    """After a generation: a bounded wait inside a one-launch MLP that gave up leaves an error word (host-synchronous read)."""
    ws = sess.cache.__dict__.get("engine_ws")
    if ws is not None and mlp_engine_error(ws):
        raise RuntimeError("persistent MLP engine: a bounded wait gave up; this generation's tokens are invalid")
    if mlp_pair_timed_out(sess.device):
        raise RuntimeError("one-launch MLP pair: a workgroup timed out waiting; this generation's tokens are invalid")


def _set_hook(fn):
    from .. import decoder as D
    from .. import model as M
    M.MLP_HOOK = fn
    D.POST_GENERATE_CHECK = _error_check if fn is not None else None
    _lib.bump_layout_epoch()                            # captured graphs baked the other MLP route in


def enable_mlp_engine():
    """Int4 one-row decode steps run their MLP as ONE persistent launch (4 launches per layer instead of 5)."""
    _lib.get_dev_lib()
    _set_hook(_engine_hook)


def enable_mlp_pair():
    """Int4 one-row decode steps run both MLP projections in one launch of chained grids (round 2's experiment)."""
    _lib.get_dev_lib()
    _set_hook(_pair_hook)


def disable_mlp_experiments():
    _set_hook(None)


def enable_w4a8():
    """Install the W4A8 route as the forward of int4 modules whose ``act_quant`` is set (the product raises without it)."""
    from ..int4 import qlinear as _q4
    _lib.get_dev_lib()
    _q4.ACT_QUANT_FORWARD = w4a8_module_forward


def disable_w4a8():
    from ..int4 import qlinear as _q4
    _q4.ACT_QUANT_FORWARD = None


def w4a8_module_forward(mod, input: Tensor) -> Tensor | None:
    """``DynamicQuantizeLinear.forward`` of an int4 module whose ``act_quant`` is set (False | True | "per_tensor"): activation
    quantiser + W4A8 GEMM through the developer library.  None when the shape is not served (the caller continues on the
    weight-only path)."""
    if not (w4a8_supported(input, mod.weight, mod.weight_scale) and mod.group_size == 32):
        return None
    key = mod._canonical_key()
    if mod._a8 is None or mod._a8_key != key:
        mod._a8, mod._a8_key = pack_w4a8(mod.weight, mod.weight_scale), key
        _lib.bump_layout_epoch()
    return w4a8_forward(input, mod._a8, mod.out_features, mod.bias, per_tensor=mod.act_quant == "per_tensor")


# ---- round 4: dequantise once per call + dense 16-bit ring GEMM (csrc/w4_dense256.hip) ------------------------------------------
def dense256_image(tiled: Tensor, n_out: int, k_in: int, dtype: torch.dtype, scale: Tensor | None = None, out: Tensor | None = None) -> Tensor:
    """The 16-bit fragment-major image of a weight (``qlinear_dev_dense256_expand``): ``tiled`` = part 2 of the int4 layout, or the
    tile-major int8 copy with its per-channel ``scale``.  ``out``: a scratch buffer to reuse (any layer's image fits the largest)."""
    lib = _lib.get_dev_lib()
    need = int(lib.qlinear_dev_dense256_image_bytes(n_out, k_in))
    img = out if out is not None and out.numel() >= need else torch.empty(need, dtype=torch.uint8, device=tiled.device)
    with torch.cuda.device(tiled.device):
        st = lib.qlinear_dev_dense256_expand(tiled.data_ptr(), _lib.ptr(scale), img.data_ptr(), n_out, k_in, _lib.dtype_code(dtype),
                                             4 if scale is None else 8, _lib.stream_ptr(tiled.device))
    _lib.check(st, "qlinear_dev_dense256_expand")
    return img


def dense256_forward(a: Tensor, image: Tensor, n_out: int, bias: Tensor | None = None, residual: Tensor | None = None,
                     gate: bool = False) -> Tensor:
    """``a @ dequant(W) (+ bias) (+ residual)`` or, with ``gate``, SiLU * gate of it (gate-interleaved weight copy): the dense ring GEMM
    on an image built by ``dense256_image``."""
    lib = _lib.get_dev_lib()
    K = a.shape[-1]
    a2 = a.reshape(-1, K)
    if a2.stride(1) != 1 or a2.stride(0) % 8:
        a2 = a2.contiguous()
    M = a2.shape[0]
    cols = n_out // 2 if gate else n_out
    c = torch.empty((M, cols), device=a.device, dtype=a.dtype)
    r2 = residual.reshape(M, n_out) if residual is not None else None
    with torch.cuda.device(a.device):
        st = lib.qlinear_dev_dense256_fwd(a2.data_ptr(), image.data_ptr(), _lib.ptr(bias), _lib.ptr(r2), c.data_ptr(), M, n_out, K, a2.stride(0),
                                          cols, r2.stride(0) if r2 is not None else 0, _lib.dtype_code(a.dtype), 1 if gate else 0,
                                          _lib.stream_ptr(a.device))
    _lib.check(st, "qlinear_dev_dense256_fwd")
    return c.reshape(*a.shape[:-1], cols)


# ---- round 6: config 3's GEMM on 128 x 128 tiles with a grid-level K split (measured slower: include/qlinear_hip_dev.h) --------------
# The kernel keeps per-tile ticket counters in a workspace its owner zeroes ONCE: one per (device, stream) here.
_splitk_ws: dict = {}


def w8a8_splitk_serves(M: int, N: int, K: int) -> int:
    """Workspace bytes of the 128 x 128-tile x 2-K-slice kernel for the shape, 0 when it does not serve it."""
    return int(_lib.get_dev_lib().qlinear_dev_w8a8_splitk_workspace_bytes(M, N, K))


def w8a8_gemm_tiled_splitk(a_q: Tensor, a_s: Tensor, tiled: Tensor, n_out: int, w_scale: Tensor, bias: Tensor | None = None) -> Tensor:
    """``qlinear_dev_w8a8_fwd_tiled_splitk``: bit-equal to int8.hip_ops.w8a8_gemm_tiled (integer sums, the same epilogue)."""
    lib = _lib.get_dev_lib()
    M, K = a_q.shape
    nbytes = w8a8_splitk_serves(M, n_out, K)
    if not nbytes:
        raise ValueError(f"the split-K kernel does not serve {M} x {K} x {n_out}")
    dev = a_q.device
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    ws = _splitk_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _splitk_ws[key] = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    c = torch.empty((M, n_out), device=dev, dtype=w_scale.dtype)
    with torch.cuda.device(dev):
        st = lib.qlinear_dev_w8a8_fwd_tiled_splitk(a_q.data_ptr(), a_s.data_ptr(), tiled.data_ptr(), w_scale.contiguous().data_ptr(),
                                                   _lib.ptr(bias), c.data_ptr(), M, n_out, K, n_out, _lib.dtype_code(w_scale.dtype),
                                                   ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(st, "qlinear_dev_w8a8_fwd_tiled_splitk")
    return c
