"""Host wrappers for the int4 group-quantised HIP kernels.

Mirrors the wrapper layer of the reference (chatglm_q/int4/triton_ops.py:10-11,90-139): same
function names, argument meaning and pre-launch checks, with the Triton launch replaced by a call
into libqlinear_hip.so.  The reference's ``assert``s are kept as real exceptions (asserts vanish
under ``python -O``).
"""
from __future__ import annotations

import functools
import os

import torch
from torch import Tensor

from .. import _lib


def check_input(a: Tensor) -> bool:
    """True when ``a`` lives on a GPU, i.e. the HIP kernels handle it
    (chatglm_q/int4/triton_ops.py:10-11)."""
    return a.get_device() >= 0


def _check_w4_args(a: Tensor, b: Tensor, b_scale: Tensor):
    # chatglm_q/int4/triton_ops.py:102-111
    if b.dim() != 2:
        raise AssertionError(f"qweight must be 2-D, got {tuple(b.shape)}")
    if b_scale.dim() != 2:
        raise AssertionError(f"scale must be 2-D, got {tuple(b_scale.shape)}")
    if a.shape[-1] != b.shape[0] * 2:
        raise AssertionError(f"K mismatch: a has {a.shape[-1]}, packed weight implies {b.shape[0] * 2}")
    if b.shape[1] != b_scale.shape[1]:
        raise AssertionError(f"N mismatch: weight {b.shape[1]} vs scale {b_scale.shape[1]}")
    if b.dtype != torch.uint8:
        raise AssertionError(f"qweight must be uint8, got {b.dtype}")
    if a.dtype != b_scale.dtype:
        raise AssertionError(f"activation dtype {a.dtype} != scale dtype {b_scale.dtype}")
    if b.shape[0] % b_scale.shape[0] != 0:
        raise AssertionError(f"packed rows {b.shape[0]} not divisible by groups {b_scale.shape[0]}")
    if a.get_device() < 0:
        raise AssertionError("activations must be on a GPU")
    if b.device != a.device:
        raise AssertionError(f"b.device={b.device}, a.device={a.device}")
    if b_scale.device != a.device:
        raise AssertionError(f"b_scale.device={b_scale.device}, a.device={a.device}")


def _check_row_operands(what: str, a: Tensor, K: int, **operands: Tensor | None):
    """Checks of the internal fast paths that bypass ``_check_w4_args`` / ``_check_w8_args`` (derived layouts, fused
    one-row launches): the kernels reinterpret scale / bias / residual / norm-weight memory in the ACTIVATION dtype, so
    a dtype or device mismatch would silently compute garbage (the reference asserts the same, triton_ops.py:104-111)."""
    if a.get_device() < 0:
        raise AssertionError(f"{what}: activations must be on a GPU")
    if a.shape[-1] != K:
        raise AssertionError(f"{what}: K mismatch: activations have {a.shape[-1]}, weights {K}")
    for name, t in operands.items():
        if t is None:
            continue
        if t.dtype != a.dtype:
            raise AssertionError(f"{what}: {name} dtype {t.dtype} != activation dtype {a.dtype}")
        if t.device != a.device:
            raise AssertionError(f"{what}: {name}.device={t.device}, a.device={a.device}")


def _rows(a: Tensor) -> Tensor:
    """Flatten leading dims to (M, K) with unit inner stride (the C ABI takes lda explicitly)."""
    a2 = a.reshape(-1, a.shape[-1])
    if a2.stride(1) != 1 or (a2.shape[0] > 1 and a2.stride(0) < a2.shape[1]):
        a2 = a2.contiguous()
    # the vector kernels read rows with 16-byte loads
    if a2.data_ptr() % 16 or (a2.shape[0] > 1 and (a2.stride(0) * a2.element_size()) % 16):
        a2 = a2.clone(memory_format=torch.contiguous_format)
    return a2


@functools.lru_cache(maxsize=None)
def packed_nbytes(N: int, K: int, dtype: torch.dtype, group: int = 32) -> int:
    return int(_lib.get_lib().qlinear_w4g32_packed_bytes(N, K, group, _lib.dtype_code(dtype)))


# rows served by the GEMV (part 1 of the derived layout).  The library reads the same variable (w4_rows_use_gemm,
# csrc/w4_packed.hip); which part a given call needs is the library's decision: rows_on_tiled().
GEMV_MAX_ROWS = int(os.environ.get("QLINEAR_GEMV_MAX_ROWS", "2"))


def _dispatch_key() -> int:
    """What the library parsed from QLINEAR_DISPATCH (qlinear_dispatch_reload parses again): the routing answers cached below depend
    on it - ADVICE r5: after a reload with `norows16` a cached "part 1 serves 5..16 rows" sent a part-1-only buffer to a kernel that
    reads part 2."""
    return int(_lib.get_lib().qlinear_dispatch_flags())


@functools.lru_cache(maxsize=None)
def _rows_on_tiled(M: int, N: int, K: int, dtype: torch.dtype, strict: bool, dispatch_key: int) -> bool:
    flags = _lib.FLAG_STRICT_ROUNDING if strict else 0
    return bool(_lib.get_lib().qlinear_w4g32_rows_on_tiled(M, N, K, _lib.dtype_code(dtype), flags))


def rows_on_tiled(M: int, N: int, K: int, dtype: torch.dtype, strict: bool = False) -> bool:
    """True when ``M`` rows of a (K, N) int4g32 weight are served from part 2 of the derived layout (tile-major: few-row and
    MFMA GEMM kernels), False when part 1 serves them (GEMV, the 4x4x4-MFMA kernel for 2..4 rows, fp32).  The library's answer,
    cached per (shape, dispatch flags)."""
    if dtype not in (torch.float16, torch.bfloat16):
        return False
    return _rows_on_tiled(M, N, K, dtype, strict, _dispatch_key())


@functools.lru_cache(maxsize=None)
def gemv_nbytes(N: int, K: int, dtype: torch.dtype, group: int = 32) -> int:
    """Bytes of part 1 (column-major, the GEMVs) of the derived layout."""
    return int(_lib.get_lib().qlinear_w4g32_gemv_bytes(N, K, group, _lib.dtype_code(dtype)))


@functools.lru_cache(maxsize=None)
def tiled_nbytes(N: int, K: int, dtype: torch.dtype, group: int = 32) -> int:
    """Bytes of part 2 (tile-major, the MFMA kernels); 0 for dtypes without an MFMA path (fp32)."""
    return int(_lib.get_lib().qlinear_w4g32_tiled_bytes(N, K, group, _lib.dtype_code(dtype)))


@functools.lru_cache(maxsize=None)
def _workspace_nbytes_for(op: int, M: int, N: int, K: int, group: int, dispatch_key: int) -> int:
    return int(_lib.get_lib().qlinear_workspace_bytes(op, M, N, K, group))


def _workspace_nbytes(op: int, M: int, N: int, K: int, group: int) -> int:
    return _workspace_nbytes_for(op, M, N, K, group, _dispatch_key())      # which kernel serves (and what it needs) follows the flags


def _repack(entry: str, nbytes: int, b: Tensor, b_scale: Tensor) -> Tensor:
    lib = _lib.get_lib()
    K, N = b.shape[0] * 2, b.shape[1]
    group = K // b_scale.shape[0]
    if nbytes == 0:
        raise ValueError(f"no packed layout for N={N}, K={K}, group={group}")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
    with torch.cuda.device(b.device):
        st = getattr(lib, entry)(b.contiguous().data_ptr(), b_scale.contiguous().data_ptr(), packed.data_ptr(), N, K, group,
                                 _lib.dtype_code(b_scale.dtype), _lib.stream_ptr(b.device))
    _lib.check(st, entry)
    return packed


def repack_w4g32(b: Tensor, b_scale: Tensor) -> Tensor:
    """Build the derived streaming layout (include/qlinear_hip.h), both parts in one buffer, from canonical buffers.
    Returns a uint8 tensor; it is a cache, never part of a state_dict."""
    K, N = b.shape[0] * 2, b.shape[1]
    return _repack("qlinear_w4g32_repack", packed_nbytes(N, K, b_scale.dtype, K // b_scale.shape[0]), b, b_scale)


def repack_w4g32_gemv(b: Tensor, b_scale: Tensor) -> Tensor:
    """Part 1 alone (rows <= GEMV_MAX_ROWS and every fused one-row launch): what the modules build first."""
    K, N = b.shape[0] * 2, b.shape[1]
    return _repack("qlinear_w4g32_repack_gemv", gemv_nbytes(N, K, b_scale.dtype, K // b_scale.shape[0]), b, b_scale)


def unpack_w4g32_gemv(gemv: Tensor, N: int, K: int, dtype: torch.dtype) -> tuple[Tensor, Tensor]:
    """The inverse of ``repack_w4g32_gemv``: part 1 of the derived layout -> the canonical ``(weight (K / 2, N) uint8, weight_scale
    (K / 32, N) dtype)`` buffers, byte for byte (``qlinear_w4g32_unpack_gemv``)."""
    lib = _lib.get_lib()
    if gemv.numel() < gemv_nbytes(N, K, dtype):
        raise AssertionError(f"unpack_w4g32_gemv: buffer of {gemv.numel()} bytes is no part 1 of a ({K}, {N}) weight")
    w = torch.empty((K // 2, N), dtype=torch.uint8, device=gemv.device)
    sc = torch.empty((K // 32, N), dtype=dtype, device=gemv.device)
    with torch.cuda.device(gemv.device):
        st = lib.qlinear_w4g32_unpack_gemv(gemv.data_ptr(), w.data_ptr(), sc.data_ptr(), N, K, 32, _lib.dtype_code(dtype),
                                           _lib.stream_ptr(gemv.device))
    _lib.check(st, "qlinear_w4g32_unpack_gemv")
    return w, sc


def tile_w4g32(gemv: Tensor, N: int, K: int, dtype: torch.dtype) -> Tensor:
    """Part 2 (tile-major, fp16 / bf16) built from part 1: what the modules build on their first forward with more rows."""
    lib = _lib.get_lib()
    nbytes = tiled_nbytes(N, K, dtype)
    if nbytes == 0 or gemv.numel() < gemv_nbytes(N, K, dtype):
        raise ValueError(f"no tile-major layout for N={N}, K={K}, {dtype}, or part 1 too small")
    tiled = torch.empty(nbytes, dtype=torch.uint8, device=gemv.device)
    with torch.cuda.device(gemv.device):
        st = lib.qlinear_w4g32_tile(gemv.data_ptr(), tiled.data_ptr(), N, K, 32, _lib.dtype_code(dtype), _lib.stream_ptr(gemv.device))
    _lib.check(st, "qlinear_w4g32_tile")
    return tiled


def w4_gemm256(a: Tensor, tiled: Tensor, n_out: int, bias: Tensor | None = None, out: Tensor | None = None) -> Tensor:
    """The many-row kernel alone (``qlinear_w4g32_fwd_tiled256``: 256 x 256 tiles) on part 2 of the derived layout, for any row
    count - ``w4_forward`` / the modules pick it by themselves at prefill row counts.  ``out``: optional (M, >= n_out) buffer
    whose row stride becomes ``ldc``."""
    lib = _lib.get_lib()
    a2 = _rows(a)
    M, K = a2.shape
    c = torch.empty((M, n_out), device=a.device, dtype=a.dtype) if out is None else out
    with torch.cuda.device(a.device):
        st = lib.qlinear_w4g32_fwd_tiled256(a2.data_ptr(), tiled.data_ptr(), _lib.ptr(bias), c.data_ptr(), M, n_out, K, 32,
                                            a2.stride(0) if M > 1 else K, c.stride(0) if M > 1 else n_out,
                                            _lib.dtype_code(a.dtype), _lib.stream_ptr(a.device))
    _lib.check(st, "qlinear_w4g32_fwd_tiled256")
    return c[:, :n_out]


def w4_forward(a: Tensor, b: Tensor, b_scale: Tensor, bias: Tensor | None = None,
               packed: Tensor | None = None, strict: bool | None = None, tiled: Tensor | None = None,
               plan_out: list | None = None) -> Tensor:
    """``a @ dequant(b, b_scale) (+ bias)`` on the GPU.  ``packed`` selects the derived-layout kernels: a buffer of
    ``repack_w4g32`` (both parts, any row count) or of ``repack_w4g32_gemv`` (part 1: rows <= GEMV_MAX_ROWS, fp32 any);
    ``tiled`` (part 2, ``tile_w4g32``) serves fp16 / bf16 at any row count.  ``strict`` asks for the reference's per-weight
    rounding bit for bit (the canonical-layout and MFMA kernels always round that way); default: the package policy
    (``_lib.strict_for``: env QLINEAR_STRICT, else strict for bf16 only).  ``plan_out``: a list that receives a pre-bound
    launch (``_lib.make_plan``) for later calls of the same shape when the call went through a derived layout."""
    _check_w4_args(a, b, b_scale)
    _check_row_operands("w4_forward", a, a.shape[-1], bias=bias)
    lib = _lib.get_lib()
    out_shape = (*a.shape[:-1], b.shape[1])
    a2 = _rows(a)
    M, K = a2.shape
    G, N = b_scale.shape
    group = K // G
    c = torch.empty((M, N), device=a.device, dtype=a.dtype)
    if M == 0:
        return c.reshape(out_shape)
    code = _lib.dtype_code(a.dtype)
    flags = _lib.FLAG_STRICT_ROUNDING if (_lib.strict_for(a.dtype) if strict is None else strict) else 0
    if bias is not None:
        bias = bias.contiguous()
    plannable = plan_out is not None and a.is_contiguous() and a.data_ptr() % 16 == 0
    with torch.cuda.device(a.device):
        stream = _lib.stream_ptr(a.device)
        mfma_rows = group == 32 and rows_on_tiled(M, N, K, a.dtype, bool(flags & _lib.FLAG_STRICT_ROUNDING))
        if tiled is not None and (mfma_rows or packed is None):
            if a.dtype not in (torch.float16, torch.bfloat16) or group != 32 or tiled.device != a.device or \
                    tiled.numel() < tiled_nbytes(N, K, a.dtype):
                raise AssertionError("w4_forward: the tile-major buffer serves fp16 / bf16, group 32, and must belong to (K, N)")
            ws_bytes = _workspace_nbytes(_lib.OP_W4G32_FWD_PACKED, max(M, GEMV_MAX_ROWS + 1), N, K, group)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device) if ws_bytes else None
            st = lib.qlinear_w4g32_fwd_tiled(a2.data_ptr(), tiled.data_ptr(), _lib.ptr(bias), c.data_ptr(), M, N, K, group,
                                             a2.stride(0) if M > 1 else K, N, code, _lib.ptr(ws), ws_bytes, stream)
            _lib.check(st, "qlinear_w4g32_fwd_tiled")
            if plannable:
                plan_out.append(_lib.make_plan(
                    "qlinear_w4g32_fwd_tiled", (None, tiled.data_ptr(), _lib.ptr(bias), None, M, N, K, group, K, N, code, None, ws_bytes, None),
                    0, 3, 13, M, K, N, a.dtype, a.device, (b, b_scale, bias), ws_slot=11 if ws_bytes else None, ws_bytes=ws_bytes,
                    keep=(tiled, bias)))
        elif packed is not None:
            need = packed_nbytes(N, K, a.dtype, group) if mfma_rows else gemv_nbytes(N, K, a.dtype, group)
            if packed.device != a.device or packed.numel() < need:
                raise AssertionError(f"w4_forward: derived buffer of {packed.numel()} bytes cannot serve {M} rows of a ({K}, {N}) "
                                     f"weight ({need} needed: part 1 alone serves few rows only, see rows_on_tiled)")
            # few-row GEMMs split K over workgroups into an fp32 workspace (0 bytes for M <= 4 and for large M)
            ws_bytes = _workspace_nbytes(_lib.OP_W4G32_FWD_PACKED, M, N, K, group) if M > 1 else 0
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device) if ws_bytes else None
            st = lib.qlinear_w4g32_fwd_packed(a2.data_ptr(), packed.data_ptr(), _lib.ptr(bias), c.data_ptr(), M, N, K,
                                              group, a2.stride(0) if M > 1 else K, N, code, flags, _lib.ptr(ws),
                                              ws_bytes, stream)
            _lib.check(st, "qlinear_w4g32_fwd_packed")
            if plannable:
                plan_out.append(_lib.make_plan(
                    "qlinear_w4g32_fwd_packed", (None, packed.data_ptr(), _lib.ptr(bias), None, M, N, K, group, K, N, code, flags, None,
                                                 ws_bytes, None),
                    0, 3, 14, M, K, N, a.dtype, a.device, (b, b_scale, bias), ws_slot=12 if ws_bytes else None, ws_bytes=ws_bytes,
                    keep=(packed, bias)))
        else:
            if not b.is_contiguous():
                b = b.contiguous()
            if not b_scale.is_contiguous():
                b_scale = b_scale.contiguous()
            ws_bytes = _workspace_nbytes(_lib.OP_W4G32_FWD, M, N, K, group)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device) if ws_bytes else None
            st = lib.qlinear_w4g32_fwd(a2.data_ptr(), b.data_ptr(), b_scale.data_ptr(), _lib.ptr(bias), c.data_ptr(),
                                       M, N, K, group, a2.stride(0) if M > 1 else K, N, code, _lib.ptr(ws), ws_bytes,
                                       stream)
            _lib.check(st, "qlinear_w4g32_fwd")
    return c.reshape(out_shape)


def w4_grad_input_supported(grad_out: Tensor, b: Tensor, b_scale: Tensor) -> bool:
    """Shapes / dtypes served by qlinear_w4g32_bwd_input (everything else takes the dense torch formula)."""
    K, N = b.shape[0] * 2, b.shape[1]
    return (grad_out.is_cuda and grad_out.dtype in (torch.float16, torch.bfloat16) and b_scale.dtype == grad_out.dtype
            and b.dtype == torch.uint8 and b_scale.shape[0] * 32 == K and N % 16 == 0 and N >= 16
            and b.is_contiguous() and b_scale.is_contiguous() and _lib.available())


def w4_grad_input(grad_out: Tensor, b: Tensor, b_scale: Tensor) -> Tensor:
    """``grad_out @ dequant(b, b_scale).T`` on the canonical layout (reference:
    dynamic_quant_matmul_transposed_s4, chatglm_q/int4/triton_ops.py:212-264)."""
    lib = _lib.get_lib()
    K, N = b.shape[0] * 2, b.shape[1]
    g2 = _rows(grad_out)
    M = g2.shape[0]
    out = torch.empty((M, K), device=grad_out.device, dtype=grad_out.dtype)
    if M:
        with torch.cuda.device(grad_out.device):
            st = lib.qlinear_w4g32_bwd_input(g2.data_ptr(), b.data_ptr(), b_scale.data_ptr(), out.data_ptr(), M, N, K, 32,
                                             g2.stride(0) if M > 1 else N, K, _lib.dtype_code(grad_out.dtype),
                                             _lib.stream_ptr(grad_out.device))
        _lib.check(st, "qlinear_w4g32_bwd_input")
    return out.reshape(*grad_out.shape[:-1], K)


def dynamic_quant_matmul_transposed_s4(a: Tensor, b: Tensor, b_scale: Tensor, allow_tf32: bool | None = None) -> Tensor:
    """Same contract as the reference wrapper (chatglm_q/int4/triton_ops.py:212-264): A (..., K), B (N//2, K) uint8,
    B_scale (N//32, K) -> (..., N).  Unlike the reference kernel it needs no power-of-two sizes (K % 16 == 0)."""
    del allow_tf32
    _check_w4_t_args(a, b, b_scale)
    return w4_grad_input(a, b, b_scale)


def _check_w4_t_args(a: Tensor, b: Tensor, b_scale: Tensor) -> None:
    if b.dim() != 2 or b_scale.dim() != 2:
        raise AssertionError("B and B_scale must be 2-D")
    if a.shape[-1] != b.shape[1] or b.shape[1] != b_scale.shape[1]:
        raise AssertionError(f"K mismatch: {a.shape[-1]}, {b.shape[1]}, {b_scale.shape[1]}")
    if b.dtype != torch.uint8 or a.dtype != b_scale.dtype:
        raise AssertionError("B must be uint8 and A / B_scale share a dtype")
    if not w4_grad_input_supported(a, b, b_scale):
        raise AssertionError("transposed int4 product: fp16 / bf16 GPU tensors, group 32, K % 16 == 0, contiguous B")


def gate_interleave(hidden: int, device=None) -> Tensor:
    """Column order (h_0, h_1, gate_0, gate_1, h_2, h_3, gate_2, gate_3, ...) of a (K, 2 * hidden) first MLP
    projection, as the SiLU * gate epilogue expects it (``hidden`` even)."""
    if hidden % 2:
        raise ValueError("gate interleave needs an even hidden size")
    t = torch.arange(hidden // 2, device=device)
    return torch.stack((2 * t, 2 * t + 1, hidden + 2 * t, hidden + 2 * t + 1), dim=1).reshape(-1)


def w4_forward_fused(kind: int, a: Tensor, packed: Tensor, n_out: int, bias: Tensor | None = None,
                     delta: Tensor | None = None, ln_weight: Tensor | None = None, hout: Tensor | None = None,
                     eps: float = 0.0, strict: bool | None = None, plan_out: list | None = None, guards=()) -> Tensor:
    """One-row forward with an activation prologue (``_lib.PRO_SILU`` / ``_lib.PRO_ADDNORM``, optionally OR-ed with
    ``_lib.EPI_SILU_GATE``) on the derived layout.  ``a``: (..., K) for ADDNORM, (..., 2K) for SILU, exactly one
    row; ``n_out`` = number of packed columns.  Returns (..., n_out), or (..., n_out / 2) with the gate epilogue
    (``packed`` then holds the gate-interleaved column order, see ``gate_interleave``).  ``strict``: as ``w4_forward``.
    ``plan_out`` / ``guards``: a list receiving a pre-bound launch ``run(a, delta, hout)`` valid while ``guards`` (the
    canonical buffers ``packed`` was built from) stay unchanged."""
    lib = _lib.get_lib()
    K = a.shape[-1] // 2 if (kind & 0xFF) == _lib.PRO_SILU else a.shape[-1]
    if a.numel() != a.shape[-1]:
        raise ValueError("fused prologues serve exactly one activation row")
    _check_row_operands("w4_forward_fused", a, a.shape[-1], bias=bias, delta=delta, ln_weight=ln_weight, hout=hout)
    if packed.device != a.device or packed.numel() < gemv_nbytes(n_out, K, a.dtype):
        raise AssertionError("w4_forward_fused: packed buffer on another device or too small for (n_out, K)")
    if _lib.strict_for(a.dtype) if strict is None else strict:
        kind |= _lib.FUSED_STRICT
    a = a.contiguous()
    cols = n_out // 2 if kind & _lib.EPI_SILU_GATE else n_out
    c = torch.empty((*a.shape[:-1], cols), device=a.device, dtype=a.dtype)
    code = _lib.dtype_code(a.dtype)
    with torch.cuda.device(a.device):
        st = lib.qlinear_w4g32_fwd_packed_fused(kind, a.data_ptr(), packed.data_ptr(), _lib.ptr(bias), c.data_ptr(),
                                                n_out, K, _lib.ptr(delta), _lib.ptr(ln_weight), _lib.ptr(hout),
                                                float(eps), code, _lib.stream_ptr(a.device))
    _lib.check(st, "qlinear_w4g32_fwd_packed_fused")
    if plan_out is not None:
        plan_out.append(_lib.make_plan(
            "qlinear_w4g32_fwd_packed_fused", (kind, None, packed.data_ptr(), _lib.ptr(bias), None, n_out, K, None, _lib.ptr(ln_weight),
                                               None, float(eps), code, None),
            1, 4, 12, 1, a.shape[-1], cols, a.dtype, a.device, (*guards, bias, ln_weight), keep=(packed, bias, ln_weight),
            extras=((7, K), (9, K))))
    return c


def w4_forward_gated(a: Tensor, gated: Tensor, n_out: int, bias: Tensor | None, part1: bool = False) -> Tensor | None:
    """Few rows through a gate-interleaved first MLP projection with SiLU * gate in the kernel's epilogue: (..., K) ->
    (..., n_out / 2).  ``gated``: part 2 of the gate-interleaved copy (``DynamicQuantizeLinear.gated_tiled``; 3..32 rows,
    ``qlinear_w4g32_fwd_tiled_gated``) or, with ``part1``, its part 1 (``gated_packed``; the row counts for which
    ``rows_on_tiled`` is False, ``qlinear_w4g32_fwd_packed_gated`` -> the 4x4x4-MFMA kernel at 2..4 rows, the 16x16x32 one-launch kernels at 3..16).  None when the library does
    not serve the shape that way (the caller then runs the projection and ``silu_mul`` separately)."""
    lib = _lib.get_lib()
    K = a.shape[-1]
    _check_row_operands("w4_forward_gated", a, K, bias=bias)
    a2 = a.reshape(-1, K)
    if a2.stride(1) != 1 or a2.stride(0) % 8:
        a2 = a2.contiguous()
    M = a2.shape[0]
    if part1:
        if rows_on_tiled(M, n_out, K, a.dtype) or n_out % 4:
            return None
        need, entry = gemv_nbytes(n_out, K, a.dtype), "qlinear_w4g32_fwd_packed_gated"
    else:
        need, entry = tiled_nbytes(n_out, K, a.dtype), "qlinear_w4g32_fwd_tiled_gated"
    if gated.device != a.device or gated.numel() < need:
        raise AssertionError("w4_forward_gated: derived buffer on another device or too small for (n_out, K)")
    c = torch.empty((M, n_out // 2), device=a.device, dtype=a.dtype)
    with torch.cuda.device(a.device):
        st = getattr(lib, entry)(a2.data_ptr(), gated.data_ptr(), _lib.ptr(bias), c.data_ptr(), M, n_out, K, a2.stride(0),
                                 n_out // 2, _lib.dtype_code(a.dtype), _lib.stream_ptr(a.device))
    if st == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(st, entry)
    return c.reshape(*a.shape[:-1], n_out // 2)


def w4_forward_tiled_residual(a: Tensor, tiled: Tensor, n_out: int, bias: Tensor | None, residual: Tensor) -> Tensor | None:
    """Prefill row counts: ``round(round(a @ dequant(W) (+ bias)) + residual)`` with the add in the 256 x 256-tile GEMM's epilogue
    (``qlinear_w4g32_fwd_tiled_residual``; bit-equal to the projection followed by ``+``).  ``tiled``: part 2 of the derived layout.
    None when that kernel does not serve the row count (the caller then adds the residual itself)."""
    lib = _lib.get_lib()
    K = a.shape[-1]
    _check_row_operands("w4_forward_tiled_residual", a, K, bias=bias, residual=residual)
    a2 = a.reshape(-1, K)
    if a2.stride(1) != 1 or a2.stride(0) % 8:
        a2 = a2.contiguous()
    M = a2.shape[0]
    r2 = residual.reshape(-1, n_out)
    if r2.shape[0] != M or r2.stride(1) != 1 or r2.stride(0) % 8:
        if r2.shape[0] != M:
            raise AssertionError("w4_forward_tiled_residual: residual rows != activation rows")
        r2 = r2.contiguous()
    if tiled.device != a.device or tiled.numel() < tiled_nbytes(n_out, K, a.dtype):
        raise AssertionError("w4_forward_tiled_residual: derived buffer on another device or too small for (n_out, K)")
    c = torch.empty((M, n_out), device=a.device, dtype=a.dtype)
    with torch.cuda.device(a.device):
        st = lib.qlinear_w4g32_fwd_tiled_residual(a2.data_ptr(), tiled.data_ptr(), _lib.ptr(bias), r2.data_ptr(), c.data_ptr(), M, n_out, K,
                                                  a2.stride(0), n_out, r2.stride(0), _lib.dtype_code(a.dtype), _lib.stream_ptr(a.device))
    if st == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(st, "qlinear_w4g32_fwd_tiled_residual")
    return c.reshape(*a.shape[:-1], n_out)


def w4_forward_residual(a: Tensor, packed: Tensor, n_out: int, bias: Tensor | None, residual: Tensor,
                        strict: bool | None = None, plan_out: list | None = None, guards=()) -> Tensor:
    """One-row forward added to the residual stream in the kernel's epilogue: round(y + residual), y = the layer's
    rounded output (``qlinear_w4g32_fwd_packed_residual``).  ``plan_out`` / ``guards``: as ``w4_forward_fused``; the plan is
    ``run(a, residual)``."""
    lib = _lib.get_lib()
    if a.numel() != a.shape[-1] or residual.numel() != n_out:
        raise ValueError("the residual epilogue serves exactly one row")
    _check_row_operands("w4_forward_residual", a, a.shape[-1], bias=bias, residual=residual)
    if packed.device != a.device or packed.numel() < gemv_nbytes(n_out, a.shape[-1], a.dtype):
        raise AssertionError("w4_forward_residual: packed buffer on another device or too small for (n_out, K)")
    flags = _lib.FLAG_STRICT_ROUNDING if (_lib.strict_for(a.dtype) if strict is None else strict) else 0
    a = a.contiguous()
    residual = residual.contiguous()
    c = torch.empty((*a.shape[:-1], n_out), device=a.device, dtype=a.dtype)
    code = _lib.dtype_code(a.dtype)
    with torch.cuda.device(a.device):
        st = lib.qlinear_w4g32_fwd_packed_residual(a.data_ptr(), packed.data_ptr(), _lib.ptr(bias), residual.data_ptr(),
                                                   c.data_ptr(), n_out, a.shape[-1], code, flags, _lib.stream_ptr(a.device))
    _lib.check(st, "qlinear_w4g32_fwd_packed_residual")
    if plan_out is not None:
        plan_out.append(_lib.make_plan(
            "qlinear_w4g32_fwd_packed_residual", (None, packed.data_ptr(), _lib.ptr(bias), None, None, n_out, a.shape[-1], code, flags, None),
            0, 4, 9, 1, a.shape[-1], n_out, a.dtype, a.device, (*guards, bias), keep=(packed, bias), extras=((3, n_out),)))
    return c


def w4_forward_rows_fused(kind: int, a: Tensor, packed: Tensor, n_out: int, bias: Tensor | None, delta: Tensor | None,
                          ln_weight: Tensor, eps: float, want_hout: bool = True):
    """2..4 rows (batched decode) with the residual add + RMSNorm prologue, optionally the SiLU * gate epilogue
    (``kind = _lib.PRO_ADDNORM [| _lib.EPI_SILU_GATE]``), in ONE launch of the 4x4x4-MFMA kernel (``qlinear_w4g32_fwd_rows_fused``):
    bit-equal to ``add_rmsnorm`` (``rmsnorm`` when ``delta`` is None) followed by the projection (and ``silu_mul``).
    ``a``, ``delta``: (..., K) with 2..4 rows in total.  Returns (out, hnew) - hnew is None when ``delta`` is None - or None
    when the library does not serve the shape that way."""
    lib = _lib.get_lib()
    K = a.shape[-1]
    a2 = a.reshape(-1, K).contiguous()
    M = a2.shape[0]
    if M < 2 or K > 8192 or rows_on_tiled(M, n_out, K, a.dtype):
        return None
    _check_row_operands("w4_forward_rows_fused", a, K, bias=bias, delta=delta, ln_weight=ln_weight)
    if packed.device != a.device or packed.numel() < gemv_nbytes(n_out, K, a.dtype):
        raise AssertionError("w4_forward_rows_fused: packed buffer on another device or too small for (n_out, K)")
    d2 = delta.reshape(-1, K).contiguous() if delta is not None else None
    if d2 is not None and d2.shape != a2.shape:
        raise AssertionError("w4_forward_rows_fused: delta shape differs from the activations'")
    hout = torch.empty_like(a2) if (d2 is not None and want_hout) else None
    cols = n_out // 2 if kind & _lib.EPI_SILU_GATE else n_out
    c = torch.empty((M, cols), device=a.device, dtype=a.dtype)
    with torch.cuda.device(a.device):
        st = lib.qlinear_w4g32_fwd_rows_fused(kind, a2.data_ptr(), packed.data_ptr(), _lib.ptr(bias), c.data_ptr(), M, n_out, K,
                                              _lib.ptr(d2), ln_weight.data_ptr(), _lib.ptr(hout), float(eps), _lib.dtype_code(a.dtype),
                                              _lib.stream_ptr(a.device))
    if st == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(st, "qlinear_w4g32_fwd_rows_fused")
    return c.reshape(*a.shape[:-1], cols), (hout.reshape(a.shape) if hout is not None else None)


def dynamic_quant_matmul_s4(a: Tensor, b: Tensor, b_scale: Tensor, allow_tf32: bool | None = None) -> Tensor:
    """Same contract as the reference wrapper (chatglm_q/int4/triton_ops.py:90-139).

    A: (..., K) float; B: (K//2, N) uint8; B_scale: (G, N) float; returns (..., N).
    ``allow_tf32`` is accepted and ignored: CDNA4 has no TF32, fp32 inputs use IEEE fp32 FMA, which
    is what the reference's own tests request (``allow_tf32=False``, tests/test_triton_ops_int4.py:20).
    """
    del allow_tf32
    return w4_forward(a, b, b_scale)


def qembedding_w4(ids: Tensor, weight: Tensor, scale: Tensor, group_size: int) -> Tensor:
    lib = _lib.get_lib()
    V, D = weight.shape[0] * 2, weight.shape[1]
    idx = ids.reshape(-1).to(torch.int64).contiguous()
    out = torch.empty((idx.numel(), D), device=weight.device, dtype=scale.dtype)
    if idx.numel():
        with torch.cuda.device(weight.device):
            st = lib.qlinear_qembedding_w4(idx.data_ptr(), weight.data_ptr(), scale.data_ptr(), out.data_ptr(),
                                           idx.numel(), V, D, group_size, _lib.dtype_code(scale.dtype),
                                           _lib.stream_ptr(weight.device))
        _lib.check(st, "qlinear_qembedding_w4")
    return out.reshape(*ids.shape, D)
