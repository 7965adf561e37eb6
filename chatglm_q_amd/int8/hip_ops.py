"""Host wrappers for the int8 per-channel HIP kernels.

Mirrors chatglm_q/int8/triton_ops.py:9-10,87-127 (``check_input``, ``dynamic_quant_matmul``) and adds
the int8-activation path (``dynamic_quant_matmul_a8``) whose semantic the reference states in
chatglm_q/int8/qlinear.py:56-70 and chatglm_q/int8/quantizer.py:11-19.
"""
from __future__ import annotations

import os

import torch
from torch import Tensor

from .. import _lib
from ..int4.hip_ops import _check_row_operands, _rows


def check_input(a: Tensor) -> bool:
    return a.get_device() >= 0


def _check_w8_args(a: Tensor, b: Tensor, b_scale: Tensor):
    # chatglm_q/int8/triton_ops.py:97-105
    if b.dim() != 2:
        raise AssertionError(f"weight must be 2-D, got {tuple(b.shape)}")
    if b_scale.dim() != 1:
        raise AssertionError(f"scale must be 1-D, got {tuple(b_scale.shape)}")
    if a.shape[-1] != b.shape[0]:
        raise AssertionError(f"K mismatch: {a.shape[-1]} vs {b.shape[0]}")
    if b.shape[1] != b_scale.shape[0]:
        raise AssertionError(f"N mismatch: {b.shape[1]} vs {b_scale.shape[0]}")
    if b.dtype != torch.int8:
        raise AssertionError(f"weight must be int8, got {b.dtype}")
    if a.dtype != b_scale.dtype:
        raise AssertionError(f"activation dtype {a.dtype} != scale dtype {b_scale.dtype}")
    if a.get_device() < 0:
        raise AssertionError("activations must be on a GPU")
    if b.device != a.device:
        raise AssertionError(f"b.device={b.device}, a.device={a.device}")
    if b_scale.device != a.device:
        raise AssertionError(f"b_scale.device={b_scale.device}, a.device={a.device}")


def w8_forward(a: Tensor, b: Tensor, b_scale: Tensor, bias: Tensor | None = None,
               strict: bool | None = None, plan_out: list | None = None, guards=()) -> Tensor:
    """``a @ (b * b_scale) (+ bias)``; ``b`` is the logical (K, N) int8 matrix with ANY strides - the
    module passes ``weight.t()`` (strides (1, K)), the reference test a contiguous (K, N)."""
    _check_w8_args(a, b, b_scale)
    _check_row_operands("w8_forward", a, a.shape[-1], bias=bias)
    lib = _lib.get_lib()
    out_shape = (*a.shape[:-1], b.shape[1])
    a2 = _rows(a)
    M, K = a2.shape
    N = b.shape[1]
    # same rule as the reference wrapper: copy only when both strides exceed 1 (triton_ops.py:109-110)
    if b.stride(0) > 1 and b.stride(1) > 1:
        b = b.contiguous()
    c = torch.empty((M, N), device=a.device, dtype=a.dtype)
    if M == 0:
        return c.reshape(out_shape)
    b_scale = b_scale.contiguous()
    if bias is not None:
        bias = bias.contiguous()
    ldw_k = b.stride(0) if K > 1 else 1
    ldw_n = b.stride(1) if N > 1 else max(K, 1)
    flags = _lib.FLAG_STRICT_ROUNDING if (_lib.strict_for(a.dtype) if strict is None else strict) else 0
    code = _lib.dtype_code(a.dtype)
    with torch.cuda.device(a.device):
        # few-row GEMMs split K over workgroups into an fp32 workspace (0 bytes for M <= 4 and for large M)
        ws_bytes = int(lib.qlinear_workspace_bytes(_lib.OP_W8_FWD, M, N, K, 0)) if M > 4 and ldw_k == 1 else 0
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device) if ws_bytes else None
        st = lib.qlinear_w8_fwd(a2.data_ptr(), b.data_ptr(), b_scale.data_ptr(), _lib.ptr(bias), c.data_ptr(), M, N, K,
                                ldw_k, ldw_n, a2.stride(0) if M > 1 else K, N, code, flags,
                                _lib.ptr(ws), ws_bytes, _lib.stream_ptr(a.device))
    _lib.check(st, "qlinear_w8_fwd")
    if plan_out is not None and a.is_contiguous() and a.data_ptr() % 16 == 0:
        plan_out.append(_lib.make_plan(
            "qlinear_w8_fwd", (None, b.data_ptr(), b_scale.data_ptr(), _lib.ptr(bias), None, M, N, K, ldw_k, ldw_n, K, N, code, flags,
                               None, ws_bytes, None),
            0, 4, 16, M, K, N, a.dtype, a.device, (*guards, b_scale, bias), ws_slot=14 if ws_bytes else None, ws_bytes=ws_bytes,
            keep=(b, b_scale, bias)))
    return c.reshape(out_shape)


def dynamic_quant_matmul(a: Tensor, b: Tensor, b_scale: Tensor, allow_tf32: bool | None = None) -> Tensor:
    """Same contract as the reference wrapper (chatglm_q/int8/triton_ops.py:87-127); ``allow_tf32`` is
    accepted and ignored (no TF32 on CDNA4; IEEE fp32 FMA)."""
    del allow_tf32
    return w8_forward(a, b, b_scale)


def tile_w8(weight_nk: Tensor) -> Tensor:
    """Tile-major derived copy of an (N, K) int8 weight for the MFMA kernels (qlinear_w8_tile)."""
    lib = _lib.get_lib()
    N, K = weight_nk.shape
    weight_nk = weight_nk.contiguous()
    out = torch.empty(int(lib.qlinear_w8_tiled_bytes(N, K)), dtype=torch.uint8, device=weight_nk.device)
    with torch.cuda.device(weight_nk.device):
        st = lib.qlinear_w8_tile(weight_nk.data_ptr(), out.data_ptr(), N, K, K, _lib.stream_ptr(weight_nk.device))
    _lib.check(st, "qlinear_w8_tile")
    return out


def w8_tiled_supported(a: Tensor, weight_nk: Tensor) -> bool:
    return (a.is_cuda and a.dtype in (torch.float16, torch.bfloat16) and weight_nk.shape[1] % 16 == 0
            and a.numel() // max(a.shape[-1], 1) > 2)


def w8_gemm256(a: Tensor, tiled: Tensor, n_out: int, w_scale: Tensor, bias: Tensor | None = None, out: Tensor | None = None) -> Tensor:
    """The many-row weight-only kernel alone (``qlinear_w8_fwd_tiled256``: 256 x 256 tiles) for any row count - ``w8_forward_tiled``
    picks it by itself at prefill row counts.  ``out``: optional (M, >= n_out) buffer (its row stride = ldc)."""
    lib = _lib.get_lib()
    a2 = _rows(a)
    M, K = a2.shape
    c = torch.empty((M, n_out), device=a.device, dtype=a.dtype) if out is None else out
    with torch.cuda.device(a.device):
        st = lib.qlinear_w8_fwd_tiled256(a2.data_ptr(), tiled.data_ptr(), w_scale.contiguous().data_ptr(), _lib.ptr(bias), c.data_ptr(),
                                         M, n_out, K, a2.stride(0) if M > 1 else K, c.stride(0) if M > 1 else n_out,
                                         _lib.dtype_code(a.dtype), _lib.stream_ptr(a.device))
    _lib.check(st, "qlinear_w8_fwd_tiled256")
    return c[:, :n_out]


def w8_forward_tiled(a: Tensor, tiled: Tensor, n_out: int, w_scale: Tensor, bias: Tensor | None = None,
                     plan_out: list | None = None, guards=()) -> Tensor:
    """``a @ (W * scale).T (+ bias)`` for >= 3 rows on the tile-major copy (few-row kernel / tiled MFMA GEMM)."""
    lib = _lib.get_lib()
    a2 = _rows(a)
    M, K = a2.shape
    _check_row_operands("w8_forward_tiled", a, K, w_scale=w_scale, bias=bias)
    if w_scale.shape != (n_out,) or (bias is not None and bias.shape != (n_out,)):
        raise AssertionError(f"w8_forward_tiled: scale / bias must have shape ({n_out},)")
    if tiled.device != a.device or tiled.numel() != int(lib.qlinear_w8_tiled_bytes(n_out, K)):
        raise AssertionError(f"w8_forward_tiled: tiled copy does not belong to a ({n_out}, {K}) weight on {a.device}")
    c = torch.empty((M, n_out), device=a.device, dtype=a.dtype)
    if bias is not None:
        bias = bias.contiguous()
    w_scale = w_scale.contiguous()
    code = _lib.dtype_code(a.dtype)
    with torch.cuda.device(a.device):
        ws_bytes = int(lib.qlinear_workspace_bytes(_lib.OP_W8_FWD_TILED, M, n_out, K, 0))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device) if ws_bytes else None
        st = lib.qlinear_w8_fwd_tiled(a2.data_ptr(), tiled.data_ptr(), w_scale.data_ptr(), _lib.ptr(bias), c.data_ptr(),
                                      M, n_out, K, a2.stride(0) if M > 1 else K, n_out, code, _lib.ptr(ws),
                                      ws_bytes, _lib.stream_ptr(a.device))
    _lib.check(st, "qlinear_w8_fwd_tiled")
    if plan_out is not None and a.is_contiguous() and a.data_ptr() % 16 == 0:
        plan_out.append(_lib.make_plan(
            "qlinear_w8_fwd_tiled", (None, tiled.data_ptr(), w_scale.data_ptr(), _lib.ptr(bias), None, M, n_out, K, K, n_out, code, None,
                                     ws_bytes, None),
            0, 4, 13, M, K, n_out, a.dtype, a.device, (*guards, w_scale, bias), ws_slot=11 if ws_bytes else None, ws_bytes=ws_bytes,
            keep=(tiled, w_scale, bias)))
    return c.reshape(*a.shape[:-1], n_out)


def _w8_tiled_epilogue_args(what: str, a: Tensor, tiled: Tensor, n_out: int, w_scale: Tensor, bias: Tensor | None):
    lib = _lib.get_lib()
    K = a.shape[-1]
    from ..int4.hip_ops import _check_row_operands
    _check_row_operands(what, a, K, w_scale=w_scale, bias=bias)
    if w_scale.shape != (n_out,) or (bias is not None and bias.shape != (n_out,)):
        raise AssertionError(f"{what}: scale / bias must have shape ({n_out},)")
    if tiled.device != a.device or tiled.numel() != int(lib.qlinear_w8_tiled_bytes(n_out, K)):
        raise AssertionError(f"{what}: tiled copy does not belong to a ({n_out}, {K}) weight on {a.device}")
    a2 = a.reshape(-1, K)
    if a2.stride(1) != 1 or a2.stride(0) % 8:
        a2 = a2.contiguous()
    return lib, K, a2


def w8_forward_tiled_gated(a: Tensor, gated_tiled: Tensor, n_out: int, w_scale: Tensor, bias: Tensor | None = None) -> Tensor | None:
    """Prefill row counts through a gate-interleaved first MLP projection with SiLU * gate in the 256 x 256-tile GEMM's epilogue
    (``qlinear_w8_fwd_tiled_gated``): (..., K) -> (..., n_out / 2); scale / bias in the same permuted order.  None when that kernel
    does not serve the row count (the caller runs the projection and ``silu_mul`` separately)."""
    lib, K, a2 = _w8_tiled_epilogue_args("w8_forward_tiled_gated", a, gated_tiled, n_out, w_scale, bias)
    M = a2.shape[0]
    c = torch.empty((M, n_out // 2), device=a.device, dtype=a.dtype)
    with torch.cuda.device(a.device):
        st = lib.qlinear_w8_fwd_tiled_gated(a2.data_ptr(), gated_tiled.data_ptr(), w_scale.data_ptr(), _lib.ptr(bias), c.data_ptr(), M, n_out,
                                            K, a2.stride(0), n_out // 2, _lib.dtype_code(a.dtype), _lib.stream_ptr(a.device))
    if st == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(st, "qlinear_w8_fwd_tiled_gated")
    return c.reshape(*a.shape[:-1], n_out // 2)


def w8_forward_tiled_residual(a: Tensor, tiled: Tensor, n_out: int, w_scale: Tensor, bias: Tensor | None, residual: Tensor) -> Tensor | None:
    """Prefill row counts: ``round(round(a @ (W * scale).T (+ bias)) + residual)`` with the add in the 256 x 256-tile GEMM's epilogue
    (``qlinear_w8_fwd_tiled_residual``).  None when that kernel does not serve the row count."""
    lib, K, a2 = _w8_tiled_epilogue_args("w8_forward_tiled_residual", a, tiled, n_out, w_scale, bias)
    M = a2.shape[0]
    if residual.dtype != a.dtype or residual.device != a.device:
        raise AssertionError("w8_forward_tiled_residual: residual dtype / device differ from the activations'")
    r2 = residual.reshape(-1, n_out)
    if r2.shape[0] != M:
        raise AssertionError("w8_forward_tiled_residual: residual rows != activation rows")
    if r2.stride(1) != 1 or r2.stride(0) % 8:
        r2 = r2.contiguous()
    c = torch.empty((M, n_out), device=a.device, dtype=a.dtype)
    with torch.cuda.device(a.device):
        st = lib.qlinear_w8_fwd_tiled_residual(a2.data_ptr(), tiled.data_ptr(), w_scale.data_ptr(), _lib.ptr(bias), r2.data_ptr(), c.data_ptr(),
                                               M, n_out, K, a2.stride(0), n_out, r2.stride(0), _lib.dtype_code(a.dtype),
                                               _lib.stream_ptr(a.device))
    if st == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(st, "qlinear_w8_fwd_tiled_residual")
    return c.reshape(*a.shape[:-1], n_out)


def w8_forward_fused(kind: int, a: Tensor, weight_nk: Tensor, w_scale: Tensor, bias: Tensor | None = None,
                     delta: Tensor | None = None, ln_weight: Tensor | None = None, hout: Tensor | None = None,
                     eps: float = 0.0, plan_out: list | None = None, guards=()) -> Tensor:
    """One-row fp16 forward with the add + RMSNorm prologue (``_lib.PRO_ADDNORM``, optionally ``| _lib.EPI_SILU_GATE``
    on gate-interleaved rows, see ``int4.hip_ops.gate_interleave``).  ``weight_nk``: the module's (N, K) buffer."""
    lib = _lib.get_lib()
    N, K = weight_nk.shape
    if a.numel() != K:
        raise ValueError("fused prologues serve exactly one activation row")
    _check_row_operands("w8_forward_fused", a, K, w_scale=w_scale, bias=bias, delta=delta, ln_weight=ln_weight, hout=hout)
    if weight_nk.dtype != torch.int8 or weight_nk.device != a.device or w_scale.numel() != N:
        raise AssertionError("w8_forward_fused: weight must be int8 (N, K) on the activations' device with N scales")
    a = a.contiguous()
    cols = N // 2 if kind & _lib.EPI_SILU_GATE else N
    c = torch.empty((*a.shape[:-1], cols), device=a.device, dtype=a.dtype)
    code = _lib.dtype_code(a.dtype)
    with torch.cuda.device(a.device):
        st = lib.qlinear_w8_fwd_fused(kind, a.data_ptr(), weight_nk.data_ptr(), w_scale.data_ptr(), _lib.ptr(bias), c.data_ptr(),
                                      N, K, weight_nk.stride(0), _lib.ptr(delta), _lib.ptr(ln_weight), _lib.ptr(hout),
                                      float(eps), code, _lib.stream_ptr(a.device))
    _lib.check(st, "qlinear_w8_fwd_fused")
    if plan_out is not None:
        plan_out.append(_lib.make_plan(
            "qlinear_w8_fwd_fused", (kind, None, weight_nk.data_ptr(), w_scale.data_ptr(), _lib.ptr(bias), None, N, K, weight_nk.stride(0),
                                     None, _lib.ptr(ln_weight), None, float(eps), code, None),
            1, 5, 14, 1, K, cols, a.dtype, a.device, (*guards, weight_nk, w_scale, bias, ln_weight),
            keep=(weight_nk, w_scale, bias, ln_weight), extras=((9, K), (11, K))))
    return c


def w8_forward_residual(a: Tensor, weight_nk: Tensor, w_scale: Tensor, bias: Tensor | None, residual: Tensor,
                        plan_out: list | None = None, guards=()) -> Tensor:
    """One-row fp16 forward added to the residual stream in the kernel's epilogue (``qlinear_w8_fwd_residual``)."""
    lib = _lib.get_lib()
    N, K = weight_nk.shape
    if a.numel() != K or residual.numel() != N:
        raise ValueError("the residual epilogue serves exactly one row")
    _check_row_operands("w8_forward_residual", a, K, w_scale=w_scale, bias=bias, residual=residual)
    if weight_nk.dtype != torch.int8 or weight_nk.device != a.device or w_scale.numel() != N:
        raise AssertionError("w8_forward_residual: weight must be int8 (N, K) on the activations' device with N scales")
    a = a.contiguous()
    residual = residual.contiguous()
    c = torch.empty((*a.shape[:-1], N), device=a.device, dtype=a.dtype)
    code = _lib.dtype_code(a.dtype)
    with torch.cuda.device(a.device):
        st = lib.qlinear_w8_fwd_residual(a.data_ptr(), weight_nk.data_ptr(), w_scale.data_ptr(), _lib.ptr(bias),
                                         residual.data_ptr(), c.data_ptr(), N, K, weight_nk.stride(0),
                                         code, _lib.stream_ptr(a.device))
    _lib.check(st, "qlinear_w8_fwd_residual")
    if plan_out is not None:
        plan_out.append(_lib.make_plan(
            "qlinear_w8_fwd_residual", (None, weight_nk.data_ptr(), w_scale.data_ptr(), _lib.ptr(bias), None, None, N, K,
                                        weight_nk.stride(0), code, None),
            0, 5, 10, 1, K, N, a.dtype, a.device, (*guards, weight_nk, w_scale, bias), keep=(weight_nk, w_scale, bias),
            extras=((4, N),)))
    return c


def w8_grad_input_supported(grad_out: Tensor, b: Tensor, b_scale: Tensor) -> bool:
    """Shapes / dtypes served by qlinear_w8_bwd_input (everything else takes the dense torch formula)."""
    return (grad_out.is_cuda and grad_out.dtype in (torch.float16, torch.bfloat16) and b_scale.dtype == grad_out.dtype
            and b.dtype == torch.int8 and b.dim() == 2 and b.shape[1] % 16 == 0 and b.shape[1] >= 16 and _lib.available())


def w8_grad_input(grad_out: Tensor, b: Tensor, b_scale: Tensor) -> Tensor:
    """``grad_out @ (b.t() * b_scale[:, None])``; ``b`` is the logical (K, N) int8 matrix (chatglm_q/int8/qlinear.py:
    41-52).  The kernel wants the contraction index N contiguous: the module's ``weight.t()`` view (strides (1, K)) is
    materialised as a (K, N) row-major copy first - one pass over the int8 weights per backward call."""
    lib = _lib.get_lib()
    K, N = b.shape
    b_kn = b.contiguous()
    g2 = _rows(grad_out)
    M = g2.shape[0]
    out = torch.empty((M, K), device=grad_out.device, dtype=grad_out.dtype)
    if M:
        with torch.cuda.device(grad_out.device):
            st = lib.qlinear_w8_bwd_input(g2.data_ptr(), b_kn.data_ptr(), b_scale.contiguous().data_ptr(), out.data_ptr(), M, N, K,
                                          g2.stride(0) if M > 1 else N, K, _lib.dtype_code(grad_out.dtype),
                                          _lib.stream_ptr(grad_out.device))
        _lib.check(st, "qlinear_w8_bwd_input")
    return out.reshape(*grad_out.shape[:-1], K)


def dynamic_quant_matmul_transposed(a: Tensor, b_T: Tensor, b_scale: Tensor, allow_tf32: bool | None = None) -> Tensor:
    """Same contract as the reference wrapper (chatglm_q/int8/triton_ops.py:205-245): A (..., K), B_T (N, K) int8,
    B_scale (K) -> (..., N)."""
    del allow_tf32
    if a.shape[-1] != b_T.shape[1] or b_T.shape[1] != b_scale.shape[0]:
        raise AssertionError(f"K mismatch: {a.shape[-1]}, {b_T.shape[1]}, {b_scale.shape[0]}")
    if b_T.dtype != torch.int8 or a.dtype != b_scale.dtype:
        raise AssertionError("B_T must be int8 and A / B_scale share a dtype")
    if not w8_grad_input_supported(a, b_T, b_scale):
        raise AssertionError("transposed int8 product: fp16 / bf16 GPU tensors, K % 16 == 0")
    return w8_grad_input(a, b_T, b_scale)


def act_quant_rowwise(a: Tensor, per_tensor: bool = False):
    """Symmetric int8 quantisation of activations in fp32 arithmetic: one scale per row (``quantize_int8``,
    chatglm_q/int8/quantizer.py:11-19) or, with ``per_tensor``, one scale for the whole tensor (the ONNX export's
    second branch, chatglm_q/int8/qlinear.py:64-70).  Returns (a_q (M, K) int8, a_scale (M,) float32)."""
    lib = _lib.get_lib()
    a2 = _rows(a)
    M, K = a2.shape
    a_q = torch.empty((M, K), device=a.device, dtype=torch.int8)
    a_s = torch.empty((M,), device=a.device, dtype=torch.float32)
    if M:
        with torch.cuda.device(a.device):
            st = lib.qlinear_act_quant_i8(a2.data_ptr(), a_q.data_ptr(), a_s.data_ptr(), M, K,
                                          a2.stride(0) if M > 1 else K, _lib.dtype_code(a.dtype),
                                          _lib.FLAG_ACT_PER_TENSOR if per_tensor else 0, _lib.stream_ptr(a.device))
        _lib.check(st, "qlinear_act_quant_i8")
    return a_q, a_s


def w8a8_forward_tiled(a: Tensor, tiled: Tensor, n_out: int, w_scale: Tensor, bias: Tensor | None = None,
                       per_tensor: bool = False) -> Tensor:
    """The int8-activation linear on the tile-major copy of the weights (``tile_w8``): activation quantisation + i8 x i8
    MFMA GEMM in one library call (``qlinear_w8a8_linear_tiled``)."""
    lib = _lib.get_lib()
    a2 = _rows(a)
    M, K = a2.shape
    _check_row_operands("w8a8_forward_tiled", a, K, w_scale=w_scale, bias=bias)
    if w_scale.shape != (n_out,) or (bias is not None and bias.shape != (n_out,)):
        raise AssertionError(f"w8a8_forward_tiled: scale / bias must have shape ({n_out},)")
    if tiled.device != a.device or tiled.numel() != int(lib.qlinear_w8_tiled_bytes(n_out, K)):
        raise AssertionError(f"w8a8_forward_tiled: tiled copy does not belong to a ({n_out}, {K}) weight on {a.device}")
    c = torch.empty((M, n_out), device=a.device, dtype=a.dtype)
    if M:
        if bias is not None:
            bias = bias.contiguous()
        with torch.cuda.device(a.device):
            ws_bytes = int(lib.qlinear_workspace_bytes(_lib.OP_W8A8_LINEAR_TILED, M, n_out, K, 0))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device)
            st = lib.qlinear_w8a8_linear_tiled(a2.data_ptr(), tiled.data_ptr(), w_scale.contiguous().data_ptr(), _lib.ptr(bias),
                                               c.data_ptr(), M, n_out, K, a2.stride(0) if M > 1 else K, n_out,
                                               _lib.dtype_code(a.dtype), _lib.FLAG_ACT_PER_TENSOR if per_tensor else 0,
                                               ws.data_ptr(), ws_bytes, _lib.stream_ptr(a.device))
        _lib.check(st, "qlinear_w8a8_linear_tiled")
    return c.reshape(*a.shape[:-1], n_out)


def w8a8_gemm_tiled(a_q: Tensor, a_s: Tensor, tiled: Tensor, n_out: int, w_scale: Tensor, bias: Tensor | None = None) -> Tensor:
    """Second step alone (``qlinear_w8a8_fwd_tiled``): pre-quantised int8 rows x tile-major int8 weights."""
    lib = _lib.get_lib()
    M, K = a_q.shape
    c = torch.empty((M, n_out), device=a_q.device, dtype=w_scale.dtype)
    if M:
        with torch.cuda.device(a_q.device):
            st = lib.qlinear_w8a8_fwd_tiled(a_q.data_ptr(), a_s.data_ptr(), tiled.data_ptr(), w_scale.contiguous().data_ptr(),
                                            _lib.ptr(bias), c.data_ptr(), M, n_out, K, n_out, _lib.dtype_code(w_scale.dtype),
                                            _lib.stream_ptr(a_q.device))
        _lib.check(st, "qlinear_w8a8_fwd_tiled")
    return c


def w8a8_gemm_tiled_gated(a_q: Tensor, a_s: Tensor, gated_tiled: Tensor, n_out: int, w_scale: Tensor, bias: Tensor | None = None) -> Tensor | None:
    """Pre-quantised int8 rows x the GATE-INTERLEAVED tile-major copy of a first MLP projection with SiLU * gate in the GEMM's epilogue
    (``qlinear_w8a8_fwd_tiled_gated``): (M, K) -> (M, n_out / 2), bit-equal to ``w8a8_gemm_tiled`` + ``silu_mul``.  ``w_scale`` / ``bias``
    in the copy's column order (``DynamicQuantizeLinear.gated_tiled``).  None when the many-row kernel does not serve the row count."""
    lib = _lib.get_lib()
    M, K = a_q.shape
    c = torch.empty((M, n_out // 2), device=a_q.device, dtype=w_scale.dtype)
    with torch.cuda.device(a_q.device):
        st = lib.qlinear_w8a8_fwd_tiled_gated(a_q.data_ptr(), a_s.data_ptr(), gated_tiled.data_ptr(), w_scale.contiguous().data_ptr(),
                                              _lib.ptr(bias), c.data_ptr(), M, n_out, K, n_out // 2, _lib.dtype_code(w_scale.dtype),
                                              _lib.stream_ptr(a_q.device))
    if st == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(st, "qlinear_w8a8_fwd_tiled_gated")
    return c


def w8a8_gemm256(a_q: Tensor, a_s: Tensor, tiled: Tensor, n_out: int, w_scale: Tensor, bias: Tensor | None = None,
                 out: Tensor | None = None) -> Tensor:
    """The many-row kernel alone (``qlinear_w8a8_fwd_tiled256``: 256 x 256 tiles, both operands by LDS-DMA) for any row count -
    ``w8a8_gemm_tiled`` picks it by itself at prefill row counts.  ``out``: optional (M, >= n_out) buffer (its row stride = ldc)."""
    lib = _lib.get_lib()
    M, K = a_q.shape
    c = torch.empty((M, n_out), device=a_q.device, dtype=w_scale.dtype) if out is None else out
    with torch.cuda.device(a_q.device):
        st = lib.qlinear_w8a8_fwd_tiled256(a_q.data_ptr(), a_s.data_ptr(), tiled.data_ptr(), w_scale.contiguous().data_ptr(),
                                           _lib.ptr(bias), c.data_ptr(), M, n_out, K, c.stride(0) if M > 1 else n_out,
                                           _lib.dtype_code(w_scale.dtype), _lib.stream_ptr(a_q.device))
    _lib.check(st, "qlinear_w8a8_fwd_tiled256")
    return c[:, :n_out]


def w8a8_forward(a: Tensor, weight_nk: Tensor, w_scale: Tensor, bias: Tensor | None = None) -> Tensor:
    """fp activations -> int8 rows -> i8 x i8 -> i32 MFMA -> ``acc * a_scale[m] * w_scale[n]`` (+ bias).
    ``weight_nk`` is the module's (N, K) row-major int8 buffer."""
    lib = _lib.get_lib()
    if weight_nk.dtype != torch.int8 or weight_nk.dim() != 2:
        raise AssertionError("weight must be a 2-D int8 (N, K) tensor")
    if a.shape[-1] != weight_nk.shape[1]:
        raise AssertionError(f"K mismatch: {a.shape[-1]} vs {weight_nk.shape[1]}")
    if a.dtype != w_scale.dtype:
        raise AssertionError(f"activation dtype {a.dtype} != scale dtype {w_scale.dtype}")
    weight_nk = weight_nk.contiguous()
    N, K = weight_nk.shape
    out_shape = (*a.shape[:-1], N)
    a_q, a_s = act_quant_rowwise(a)
    M = a_q.shape[0]
    c = torch.empty((M, N), device=a.device, dtype=a.dtype)
    if M:
        if bias is not None:
            bias = bias.contiguous()
        with torch.cuda.device(a.device):
            ws_bytes = int(lib.qlinear_workspace_bytes(_lib.OP_W8A8_FWD, M, N, K, 0))    # int32 split-K slabs, few row tiles
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device) if ws_bytes else None
            st = lib.qlinear_w8a8_fwd(a_q.data_ptr(), a_s.data_ptr(), weight_nk.data_ptr(), w_scale.contiguous().data_ptr(),
                                      _lib.ptr(bias), c.data_ptr(), M, N, K, N, _lib.dtype_code(a.dtype),
                                      _lib.ptr(ws), ws_bytes, _lib.stream_ptr(a.device))
        _lib.check(st, "qlinear_w8a8_fwd")
    return c.reshape(out_shape)


def qembedding_w8(ids: Tensor, weight: Tensor, scale: Tensor) -> Tensor:
    lib = _lib.get_lib()
    V, D = weight.shape
    idx = ids.reshape(-1).to(torch.int64).contiguous()
    out = torch.empty((idx.numel(), D), device=weight.device, dtype=scale.dtype)
    if idx.numel():
        with torch.cuda.device(weight.device):
            st = lib.qlinear_qembedding_w8(idx.data_ptr(), weight.data_ptr(), scale.data_ptr(), out.data_ptr(),
                                           idx.numel(), V, D, _lib.dtype_code(scale.dtype),
                                           _lib.stream_ptr(weight.device))
        _lib.check(st, "qlinear_qembedding_w8")
    return out.reshape(*ids.shape, D)
