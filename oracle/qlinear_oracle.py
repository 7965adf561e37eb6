"""CPU oracle for the quantized-linear forward path (TEST INFRASTRUCTURE ONLY).

This file is a numpy restatement of the *algorithm* of K024/chatglm-q's
weight-only quantized linear forward.  It is the checker the parity tests,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use; the
product package ``chatglm_q_amd`` never imports it (tests/test_layout.py
enforces that).

Parity status: PINNED.  ``tests/golden/*.npz`` were generated in the build
container by importing the reference itself (torch fallback route and the
reference's ``@triton.jit`` kernels under ``TRITON_INTERPRET=1``; generator
committed as ``tests/golden/make_golden.py``) and ``tests/test_oracle_golden.py``
checks every function below against them.

Each function cites the reference lines it restates (paths relative to the
reference checkout).

Conventions
-----------
* "act dtype" is one of ``"f32"``, ``"f16"``, ``"bf16"``.  numpy has no bf16, so
  bf16 tensors travel as float32 arrays whose values are exactly
  bf16-representable (``round_to(x, "bf16")`` produces those).
* Accumulation is done in float64 and rounded ONCE to the act dtype, which is
  the semantic of the reference kernels (fp32 accumulator, one final cast:
  chatglm_q/int4/triton_ops.py:66-80) up to fp32-vs-fp64 accumulation noise
  (~1e-7 relative), far inside the stated tolerances.
"""
from __future__ import annotations

import numpy as np

DEFAULT_GROUP_SIZE = 32  # chatglm_q/int4/qlinear.py:5
MAX_Q_INT4 = 7           # chatglm_q/int4/quantizer.py:8
MAX_Q_INT8 = 127         # chatglm_q/int8/quantizer.py:7
SCALE_FLOOR = 1e-10      # chatglm_q/int4/quantizer.py:23, chatglm_q/int8/quantizer.py:17


# --------------------------------------------------------------------------
# dtype helpers
# --------------------------------------------------------------------------
def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit pattern (uint16), round-to-nearest-even, NaN kept."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    rounding = ((u >> 16) & 1) + np.uint32(0x7FFF)
    out = ((u + rounding) >> 16).astype(np.uint16)
    nan = np.isnan(x)
    if nan.any():
        out = np.where(nan, np.uint16(0x7FC0), out)
    return out


def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    bits = np.ascontiguousarray(bits, dtype=np.uint16)
    return (bits.astype(np.uint32) << 16).view(np.float32)


def round_to(x: np.ndarray, dtype: str) -> np.ndarray:
    """Round a float array to the act dtype; result container is float16 for
    "f16" and float32 (holding representable values) for "f32"/"bf16"."""
    if dtype == "f32":
        return np.asarray(x, dtype=np.float32)
    if dtype == "f16":
        # float64 -> float16 directly would double-round via nothing: numpy
        # rounds correctly from float64 to float16 in one step.
        return np.asarray(x).astype(np.float16)
    if dtype == "bf16":
        # float64 -> float32 -> bf16 is a double rounding; avoid it by rounding
        # from float64 with sticky information folded into the float32 LSB.
        x64 = np.asarray(x, dtype=np.float64)
        x32 = x64.astype(np.float32)
        # fix up double rounding: if the float32 rounding was inexact and landed
        # exactly on a bf16 tie, nudge one float32 ulp towards the true value.
        u = x32.view(np.uint32).copy()
        is_tie = (u & np.uint32(0xFFFF)) == np.uint32(0x8000)
        err = x64 - x32.astype(np.float64)
        finite = np.isfinite(x32)
        up = is_tie & finite & (((err > 0) & (x32 >= 0)) | ((err < 0) & (x32 < 0)))
        dn = is_tie & finite & (((err < 0) & (x32 >= 0)) | ((err > 0) & (x32 < 0)))
        u = np.where(up, u + np.uint32(1), u)
        u = np.where(dn, u - np.uint32(1), u)
        return bf16_bits_to_f32(f32_to_bf16_bits(u.view(np.float32)))
    raise ValueError(f"unknown act dtype {dtype!r}")


def as_f64(x: np.ndarray) -> np.ndarray:
    return np.asarray(x).astype(np.float64)


def dtype_of(arr: np.ndarray, hint: str | None = None) -> str:
    if hint is not None:
        return hint
    if arr.dtype == np.float16:
        return "f16"
    if arr.dtype == np.float32:
        return "f32"
    raise ValueError("pass dtype= explicitly for bf16 (carried as float32)")


# --------------------------------------------------------------------------
# int4 group-32 format
# --------------------------------------------------------------------------
def unpack_int4_codes(qweight: np.ndarray) -> np.ndarray:
    """(K/2, N) uint8 -> (K, N) int8 in [-8, 7].

    Byte [k//2, n] holds row 2*(k//2) in its low nibble and row 2*(k//2)+1 in
    its high nibble (pack: chatglm_q/int4/quantizer.py:27-28; unpack:
    chatglm_q/int4/qlinear.py:29-31).  Stored nibble = q + 8.
    """
    qweight = np.asarray(qweight)
    assert qweight.dtype == np.uint8 and qweight.ndim == 2
    k2, n = qweight.shape
    out = np.empty((k2 * 2, n), dtype=np.int8)
    out[0::2] = (qweight & 0xF).astype(np.int8) - 8
    out[1::2] = ((qweight >> 4) & 0xF).astype(np.int8) - 8
    return out


def unpack_int4(qweight: np.ndarray, scale: np.ndarray, dtype: str | None = None) -> np.ndarray:
    """Dense dequantised (K, N) weight in the act dtype.

    Restates ``unpack_int4`` (chatglm_q/int4/qlinear.py:20-33): the product
    ``(nibble - 8) * scale[g, n]`` is rounded to the scale's dtype; the Triton
    kernel does the same per element (chatglm_q/int4/triton_ops.py:71-73).
    """
    dt = dtype_of(scale, dtype)
    codes = unpack_int4_codes(qweight)
    k, n = codes.shape
    g = scale.shape[0]
    assert scale.shape == (g, n) and k % g == 0, (scale.shape, codes.shape)
    group_k = k // g
    prod = codes.reshape(g, group_k, n).astype(np.float64) * as_f64(scale)[:, None, :]
    return round_to(prod.reshape(k, n), dt)


def w4_matmul(a: np.ndarray, qweight: np.ndarray, scale: np.ndarray,
              bias: np.ndarray | None = None, dtype: str | None = None) -> np.ndarray:
    """int4g32 QLinear forward: ``out = a @ dequant(qweight, scale) (+ bias)``.

    Restates DynamicQuantizeMatMul.forward (chatglm_q/int4/qlinear.py:44-51)
    followed by DynamicQuantizeLinear.forward's in-place bias add AFTER the cast
    to the act dtype (chatglm_q/int4/qlinear.py:90-94).  Leading dims of ``a``
    are flattened (chatglm_q/int4/triton_ops.py:100-101,139).
    """
    dt = dtype_of(scale, dtype)
    w = as_f64(unpack_int4(qweight, scale, dt))
    lead = a.shape[:-1]
    a2 = as_f64(a).reshape(-1, a.shape[-1])
    assert a2.shape[1] == w.shape[0], (a.shape, w.shape)
    out = round_to(a2 @ w, dt)
    if bias is not None:
        out = round_to(as_f64(out) + as_f64(bias)[None, :], dt)
    return out.reshape(*lead, w.shape[1])


def w4_matmul_grad_input(grad_out: np.ndarray, qweight: np.ndarray, scale: np.ndarray,
                         dtype: str | None = None) -> np.ndarray:
    """Backward of the int4g32 product w.r.t. the activations: ``grad_A = grad_out @ dequant(qweight, scale).T``.

    Restates DynamicQuantizeMatMul.backward (chatglm_q/int4/qlinear.py:53-64): the dense weight is rounded to the
    activation dtype first (unpack_int4), one rounding of the result; the transposed Triton kernel does the same
    per tile (chatglm_q/int4/triton_ops.py:191-197)."""
    dt = dtype_of(scale, dtype)
    w = as_f64(unpack_int4(qweight, scale, dt))
    lead = grad_out.shape[:-1]
    g2 = as_f64(grad_out).reshape(-1, grad_out.shape[-1])
    assert g2.shape[1] == w.shape[1], (grad_out.shape, w.shape)
    return round_to(g2 @ w.T, dt).reshape(*lead, w.shape[0])


def quantize_int4(x: np.ndarray, group_k: int = DEFAULT_GROUP_SIZE, dtype: str = "f32"):
    """RTN int4 group quantiser; x is (K, N) in the act dtype.

    Restates quantize_int4 (chatglm_q/int4/quantizer.py:11-29): per-group
    abs-max / 7 clamped at 1e-10, ``round`` is half-to-even (torch.round),
    clamp to [-7, 7], +8, pack two K rows per byte.  All arithmetic is carried
    out in the tensor's own dtype as torch does (division and the clamp floor
    are rounded to ``dtype``).
    """
    x = np.asarray(x)
    k, n = x.shape
    assert k % group_k == 0
    g = k // group_k
    xg = as_f64(x).reshape(g, group_k, n)
    w_max = np.abs(xg).max(axis=1, keepdims=True)
    scale = round_to(w_max / MAX_Q_INT4, dtype)                    # division rounded to dtype
    floor = as_f64(round_to(np.array(SCALE_FLOOR), dtype))        # 1e-10 in dtype (0 in f16)
    scale = np.maximum(as_f64(scale), floor)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = as_f64(round_to(xg / scale, dtype))                    # x / scale rounded to dtype
    q = np.clip(np.rint(q), -MAX_Q_INT4, MAX_Q_INT4)               # rint = half-to-even
    with np.errstate(invalid="ignore"):                            # 0/0 groups in f16 (floor underflows)
        q = (q + 8).astype(np.uint8).reshape(k, n)
    packed = (q[0::2] & 0xF) | ((q[1::2] & 0xF) << 4)
    return np.ascontiguousarray(packed.astype(np.uint8)), round_to(scale.reshape(g, n), dtype)


# --------------------------------------------------------------------------
# int8 per-output-channel format
# --------------------------------------------------------------------------
def w8_dequant(w_kn: np.ndarray, scale: np.ndarray, dtype: str | None = None) -> np.ndarray:
    """``B * b_scale`` with B (K, N) int8, scale (N,) -> act dtype
    (chatglm_q/int8/qlinear.py:38, chatglm_q/int8/triton_ops.py:70)."""
    dt = dtype_of(scale, dtype)
    assert w_kn.dtype == np.int8 and w_kn.ndim == 2 and scale.shape == (w_kn.shape[1],)
    return round_to(w_kn.astype(np.float64) * as_f64(scale)[None, :], dt)


def w8_matmul(a: np.ndarray, w_kn: np.ndarray, scale: np.ndarray,
              bias: np.ndarray | None = None, dtype: str | None = None) -> np.ndarray:
    """int8 QLinear forward ``a @ (B * b_scale) (+ bias)``; B is the (K, N) view,
    i.e. ``module.weight.t()`` (chatglm_q/int8/qlinear.py:32-39,89-93)."""
    dt = dtype_of(scale, dtype)
    w = as_f64(w8_dequant(w_kn, scale, dt))
    lead = a.shape[:-1]
    a2 = as_f64(a).reshape(-1, a.shape[-1])
    out = round_to(a2 @ w, dt)
    if bias is not None:
        out = round_to(as_f64(out) + as_f64(bias)[None, :], dt)
    return out.reshape(*lead, w.shape[1])


def w8_matmul_grad_input(grad_out: np.ndarray, w_kn: np.ndarray, scale: np.ndarray,
                         dtype: str | None = None) -> np.ndarray:
    """``grad_A = grad_out @ (B.t() * b_scale[:, None])`` (chatglm_q/int8/qlinear.py:41-52); B is the (K, N) view."""
    dt = dtype_of(scale, dtype)
    w = as_f64(w8_dequant(w_kn, scale, dt))
    lead = grad_out.shape[:-1]
    g2 = as_f64(grad_out).reshape(-1, grad_out.shape[-1])
    return round_to(g2 @ w.T, dt).reshape(*lead, w.shape[0])


def quantize_int8(x: np.ndarray, dtype: str = "f32"):
    """Row-wise symmetric int8 quantiser (chatglm_q/int8/quantizer.py:11-19):
    weights (out, in) or activations (..., features); arithmetic in ``dtype``."""
    x = np.asarray(x)
    assert x.ndim == 2
    xf = as_f64(x)
    w_max = np.abs(xf).max(axis=1, keepdims=True)
    scale = round_to(w_max / MAX_Q_INT8, dtype)
    floor = as_f64(round_to(np.array(SCALE_FLOOR), dtype))
    scale = np.maximum(as_f64(scale), floor)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = as_f64(round_to(xf / scale, dtype))
    with np.errstate(invalid="ignore"):
        q = np.clip(np.rint(q), -MAX_Q_INT8, MAX_Q_INT8).astype(np.int8)
    return q, round_to(scale[:, 0], dtype)


# --------------------------------------------------------------------------
# W8A8: int8 activations x int8 weights (semantic defined in SURVEY.md §8a-A7)
# --------------------------------------------------------------------------
def act_quant_rowwise(a: np.ndarray):
    """Row-wise symmetric activation quantisation in float32 arithmetic.

    Follows quantize_int8's formula (chatglm_q/int8/quantizer.py:11-19, whose
    docstring names activations) evaluated in float32 regardless of the act
    dtype: scale = max(max|a_row| / 127, 1e-10), q = clamp(rint(a / scale)).
    Returns (int8 (M, K), float32 (M,)).
    """
    a32 = np.asarray(a).astype(np.float32).reshape(-1, a.shape[-1])
    w_max = np.abs(a32).max(axis=1, keepdims=True)
    scale = np.maximum((w_max / np.float32(MAX_Q_INT8)).astype(np.float32), np.float32(SCALE_FLOOR))
    q = np.clip(np.rint((a32 / scale).astype(np.float32)), -MAX_Q_INT8, MAX_Q_INT8).astype(np.int8)
    return q, scale[:, 0].astype(np.float32)


def act_quant_per_tensor(a: np.ndarray):
    """Per-TENSOR symmetric activation quantisation in float32 arithmetic: the second branch of
    DynamicQuantizeMatMul.symbolic (chatglm_q/int8/qlinear.py:64-70):
    ``A_scale = ReduceMax(Abs(A)) / 127``; ``QuantizeLinear(A, A_scale, zero_point=0)`` = saturate(round-half-even(
    A / A_scale)) to int8.  The reference never executes this in PyTorch (ONNX export only; its comment says it NaNs
    on CUDA: an all-zero tensor gives 0 / 0) - the scale floor of quantize_int8 (1e-10) is applied to keep that out.
    Returns (int8 (M, K), float32 (M,) - the one scale repeated per row, the layout the GEMM epilogue takes)."""
    a32 = np.asarray(a).astype(np.float32).reshape(-1, a.shape[-1])
    scale = np.maximum((np.abs(a32).max() / np.float32(MAX_Q_INT8)).astype(np.float32), np.float32(SCALE_FLOOR))
    q = np.clip(np.rint((a32 / scale).astype(np.float32)), -MAX_Q_INT8, MAX_Q_INT8).astype(np.int8)
    return q, np.full((a32.shape[0],), scale, dtype=np.float32)


def w8a8_acc_i32(a_q: np.ndarray, w_nk: np.ndarray) -> np.ndarray:
    """Exact integer contraction: (M, K) int8 x (N, K) int8 -> (M, N) int32
    (MatMulInteger of the ONNX symbolic, chatglm_q/int8/qlinear.py:60,68)."""
    return a_q.astype(np.int32) @ w_nk.astype(np.int32).T


def w8a8_matmul(a: np.ndarray, w_nk: np.ndarray, w_scale: np.ndarray,
                bias: np.ndarray | None = None, dtype: str | None = None, per_tensor: bool = False) -> np.ndarray:
    """fp in -> row-wise int8 act-quant -> i8 x i8 -> i32 -> ``acc * a_scale[m] *
    w_scale[n]`` in float32 -> act dtype (+ bias after the cast).  Epilogue shape
    follows chatglm_q/int8/qlinear.py:61-62 (Cast then Mul by A_scale*b_scale)."""
    dt = dtype_of(w_scale, dtype)
    lead = a.shape[:-1]
    a_q, a_s = act_quant_per_tensor(a) if per_tensor else act_quant_rowwise(a)
    acc = w8a8_acc_i32(a_q, w_nk).astype(np.float32)
    comb = (a_s[:, None] * np.asarray(w_scale).astype(np.float32)[None, :]).astype(np.float32)
    out = round_to((acc * comb).astype(np.float32), dt)
    if bias is not None:
        out = round_to(as_f64(out) + as_f64(bias)[None, :], dt)
    return out.reshape(*lead, w_nk.shape[0])


def w4a8_group_acc_i32(a_q: np.ndarray, qweight: np.ndarray) -> np.ndarray:
    """Exact integer stage of W4A8: per group of 32 k, (M, K) int8 x (K, N) int4 codes -> (G, M, N) int32.
    Codes = nibble - 8 (chatglm_q/int4/qlinear.py:24-30, chatglm_q/int4/triton_ops.py:71-72)."""
    codes = unpack_int4_codes(qweight).astype(np.int32)                   # (K, N)
    k, n = codes.shape
    g = k // DEFAULT_GROUP_SIZE
    a3 = a_q.astype(np.int32).reshape(a_q.shape[0], g, DEFAULT_GROUP_SIZE)
    return np.einsum("mgk,gkn->gmn", a3, codes.reshape(g, DEFAULT_GROUP_SIZE, n))


def w4a8_matmul(a: np.ndarray, qweight: np.ndarray, scale: np.ndarray, bias: np.ndarray | None = None,
                dtype: str | None = None, per_tensor: bool = False) -> np.ndarray:
    """int4g32 weights x int8-quantised activations (SURVEY.md 8d config 5 "W4A8 act-quant path"): the activation side
    of the int8 path - quantize_int8 per row (chatglm_q/int8/quantizer.py:11-19) or per tensor (chatglm_q/int8/
    qlinear.py:64-70), integer matmul, Cast, Mul by the scales (qlinear.py:60-62) - with the int4 weight decode
    (nibble - 8) * scale[group, column].  out = round(a_scale[m] * sum_g scale[g, n] * acc_i32[g, m, n]) (+ bias after
    the cast, chatglm_q/int4/qlinear.py:92-93); the group sum is evaluated in float64 here."""
    dt = dtype_of(scale, dtype)
    lead = a.shape[:-1]
    a_q, a_s = act_quant_per_tensor(a) if per_tensor else act_quant_rowwise(a)
    acc = w4a8_group_acc_i32(a_q, qweight).astype(np.float64)             # (G, M, N)
    y = np.einsum("gmn,gn->mn", acc, as_f64(scale)) * a_s.astype(np.float64)[:, None]
    out = round_to(y, dt)
    if bias is not None:
        out = round_to(as_f64(out) + as_f64(bias)[None, :], dt)
    return out.reshape(*lead, qweight.shape[1])


# --------------------------------------------------------------------------
# quantized embeddings ("next" row N3)
# --------------------------------------------------------------------------
def qembedding_int4(ids: np.ndarray, qweight: np.ndarray, scale: np.ndarray,
                    group_size: int = DEFAULT_GROUP_SIZE, dtype: str | None = None) -> np.ndarray:
    """int4 QEmbedding.forward (chatglm_q/int4/qlinear.py:122-131): packing runs
    along the vocabulary axis: token t -> byte row t//2, nibble t%2, group t//gs."""
    dt = dtype_of(scale, dtype)
    ids = np.asarray(ids)
    b = qweight[ids // 2].astype(np.int32)
    shift = ((ids % 2) * 4)[..., None]
    codes = ((b >> shift) & 0xF) - 8
    return round_to(codes.astype(np.float64) * as_f64(scale[ids // group_size]), dt)


def qembedding_int8(ids: np.ndarray, weight: np.ndarray, scale: np.ndarray,
                    dtype: str | None = None) -> np.ndarray:
    """int8 QEmbedding.forward (chatglm_q/int8/qlinear.py:118-120)."""
    dt = dtype_of(scale, dtype)
    return round_to(weight[np.asarray(ids)].astype(np.float64) * as_f64(scale)[None, :], dt)


# --------------------------------------------------------------------------
# error metrics used by the tests
# --------------------------------------------------------------------------
def rel_l2(y: np.ndarray, ref: np.ndarray) -> float:
    y = as_f64(y).ravel()
    ref = as_f64(ref).ravel()
    den = np.linalg.norm(ref)
    return float(np.linalg.norm(y - ref) / (den if den > 0 else 1.0))


def max_norm_err(y: np.ndarray, ref: np.ndarray) -> float:
    y = as_f64(y).ravel()
    ref = as_f64(ref).ravel()
    den = np.abs(ref).max()
    return float(np.abs(y - ref).max() / (den if den > 0 else 1.0))


# --------------------------------------------------------------------------
# the decode loop's sampler (caller of the hot path: chatglm_q/decoder.py:12-27)
# --------------------------------------------------------------------------
def top_p_filter(logits: np.ndarray, top_k: int = 100, top_p: float = 0.8, temperature: float = 1.0):
    """The deterministic part of ``top_p_sampling`` (chatglm_q/decoder.py:12-22): fp32 softmax of logits / temperature
    (:14), descending sort cut to top_k (:15-17; ties keep the lower index first - what torch's CPU sort returns and what
    a stable sort guarantees), entries whose PRECEDING cumulative mass exceeds top_p zeroed (:20-21), renormalised (:22).
    Returns (probs float32 (min(top_k, N),), indices int64) - the distribution ``torch.multinomial`` then draws from (:25)
    and the token ids the drawn position maps to (:26)."""
    x = np.asarray(logits, dtype=np.float32) / np.float32(temperature)
    e = np.exp(x - x.max(), dtype=np.float32)
    probs = e / e.sum(dtype=np.float32)
    order = np.argsort(-probs.astype(np.float64), kind="stable")[:top_k]
    p = probs[order].astype(np.float32)
    cum = np.cumsum(p, dtype=np.float32)
    p = np.where((cum - p) > np.float32(top_p), np.float32(0), p)
    return (p / p.sum(dtype=np.float32)).astype(np.float32), order.astype(np.int64)
