"""int8 per-output-channel linear / embedding modules (drop-in for chatglm_q/int8/qlinear.py).

Same public names, constructor signatures and registered buffers as the reference: ``weight``
(out, in) int8, ``weight_scale`` (out,), ``bias``.  GPU activations run the HIP kernels (and raise if
libqlinear_hip.so is missing); CPU activations take the dense formula ``A @ (B * b_scale)``, the
reference's own CPU branch (chatglm_q/int8/qlinear.py:35-38) and BASELINE config 1.

``DynamicQuantizeLinear.act_quant`` (default False) switches the module to the int8-activation MFMA
path: row-wise symmetric activation quantisation followed by a true i8 x i8 -> i32 contraction.  The
reference executes that semantic only in its ONNX export (chatglm_q/int8/qlinear.py:56-70); it is NOT
bit-compatible with the weight-only path (quantisation error ~1e-2 relative) and is opt-in.
"""
from __future__ import annotations

import torch
from typing import Optional

from torch import Tensor, nn
from torch.autograd.function import FunctionCtx

from .. import _lib
from . import hip_ops
from .hip_ops import check_input

KERNEL_IMPL = "hip" if _lib.available() else "none"


class DynamicQuantizeMatMul(torch.autograd.Function):
    """A: (m, k) float; B: (k, n) int8; b_scale: (n,) float (chatglm_q/int8/qlinear.py:19-52)."""

    @staticmethod
    def forward(ctx: FunctionCtx, A: Tensor, B: Tensor, b_scale: Tensor):
        ctx.save_for_backward(A, B, b_scale)
        if check_input(A):
            return hip_ops.dynamic_quant_matmul(A, B, b_scale)
        return A.matmul(B * b_scale)

    @staticmethod
    def backward(ctx: FunctionCtx, grad_out: Tensor):
        A, B, b_scale = ctx.saved_tensors
        grad_A = None
        if ctx.needs_input_grad[0]:
            if check_input(A) and hip_ops.w8_grad_input_supported(grad_out, B, b_scale):
                grad_A = hip_ops.w8_grad_input(grad_out, B, b_scale)      # qlinear_w8_bwd_input
            else:
                grad_A = grad_out.matmul(B.t() * b_scale[:, None])
        return grad_A, None, None


def dynamic_quant_matmul(A: Tensor, B: Tensor, b_scale: Tensor) -> Tensor:
    return DynamicQuantizeMatMul.apply(A, B, b_scale)


def dynamic_quant_matmul_a8(A: Tensor, weight_nk: Tensor, b_scale: Tensor, bias: Tensor | None = None) -> Tensor:
    """int8-activation variant: ``weight_nk`` is the (N, K) buffer (NOT transposed)."""
    if not check_input(A):
        raise RuntimeError("the int8-activation path exists only on the GPU (v_mfma_i32_32x32x32_i8)")
    return hip_ops.w8a8_forward(A, weight_nk, b_scale, bias)


class DynamicQuantizeLinear(nn.Module):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.register_buffer("weight", torch.empty((out_features, in_features), device=device, dtype=torch.int8))
        self.register_buffer("weight_scale", torch.empty(out_features, device=device, dtype=dtype))
        if bias:
            self.register_buffer("bias", torch.empty(out_features, device=device, dtype=dtype))
        else:
            self.register_buffer("bias", None)
        self._tiled, self._tiled_key = None, None
        self._gated, self._gated_key = None, None
        self._gated_tiled, self._gated_tiled_key = None, None
        self._plans: dict = {}            # pre-bound launches (``_lib.make_plan``): forward() by input.numel() ...
        self._fast: dict = {}             # ... and the fused one-row launches of the decode step by call site
        self.act_quant = False

    def invalidate(self):
        """Drop the derived copies (tile-major weights, gate-interleaved rows); rebuilt on the next GPU forward.  Needed
        only after a write the version counter cannot see (``weight.data.copy_``, raw pointers, inference tensors)."""
        self._tiled, self._tiled_key = None, None
        self._gated, self._gated_key = None, None
        self._gated_tiled, self._gated_tiled_key = None, None
        self._plans, self._fast = {}, {}
        _lib.bump_layout_epoch()
        return self

    def release(self, *parts: str):
        """Free derived copies a deployment no longer needs ("tiled", "gated", "gated_tiled"); rebuilt on demand."""
        for part in parts:
            if part not in ("tiled", "gated", "gated_tiled"):
                raise ValueError(f"unknown derived layout {part!r}")
            setattr(self, "_" + part, None)
            setattr(self, "_" + part + "_key", None)
        self._plans, self._fast = {}, {}
        _lib.bump_layout_epoch()
        return self

    def derived_nbytes(self) -> dict:
        def nb(t):
            if isinstance(t, tuple):
                return sum(nb(x) for x in t)
            return 0 if t is None else t.numel() * t.element_size()
        return {"tiled": nb(self._tiled), "gated": nb(self._gated), "gated_tiled": nb(self._gated_tiled),
                "canonical": nb(self.weight) + nb(self.weight_scale)}

    def __setattr__(self, name, value):
        super().__setattr__(name, value)
        if name in ("weight", "weight_scale", "bias", "act_quant") and "_plans" in self.__dict__:
            self._plans, self._fast = {}, {}
            _lib.bump_layout_epoch()

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self.invalidate()

    def _apply(self, fn, *args, **kwargs):      # .to() / .cuda() / .half(): new storage (whose address may be a reused one)
        self.invalidate()
        return super()._apply(fn, *args, **kwargs)

    def forward_quantized(self, a_q: Tensor, a_scale: Tensor) -> Tensor:
        """The int8-activation path (``act_quant``) for rows that arrive ALREADY quantised - int8 ``(rows, in_features)`` +
        one fp32 scale per row, e.g. from ``fused_ops.rmsnorm_quant`` / ``silu_mul_quant`` - : the GEMM launch alone, bit for
        bit what ``forward`` computes from the 16-bit rows those came from (chatglm_q/int8/qlinear.py:56-62)."""
        if a_q.dtype != torch.int8 or a_q.dim() != 2 or a_q.shape[1] != self.in_features or self.in_features % 16:
            raise ValueError("forward_quantized takes int8 rows of in_features (a multiple of 16) values")
        return hip_ops.w8a8_gemm_tiled(a_q, a_scale, self.prepare()._tiled, self.out_features, self.weight_scale, self.bias)

    def forward_quantized_gated(self, a_q: Tensor, a_scale: Tensor, hidden: int) -> Optional[Tensor]:
        """``forward_quantized`` of a first MLP projection (``out_features == 2 * hidden``) followed by SiLU * gate, in ONE launch at
        prefill row counts (the int8 x int8 ring GEMM's gate epilogue on the gate-interleaved copy): (rows, hidden), bit-equal to the
        two launches.  None when the library does not serve the shape that way - asked BEFORE the copy is built."""
        if a_q.dtype != torch.int8 or a_q.dim() != 2 or a_q.shape[1] != self.in_features:
            raise ValueError("forward_quantized_gated takes int8 rows of in_features values")
        from .. import _lib
        if (self.out_features != 2 * hidden or hidden % 2 or self.in_features % 128 or self.in_features < 256 or
                not _lib.get_lib().qlinear_gated_serves(a_q.shape[0], self.out_features, self.in_features,
                                                        _lib.dtype_code(self.weight_scale.dtype), 88)):
            return None
        tiled, s_perm, b_perm = self.gated_tiled(hidden)
        return hip_ops.w8a8_gemm_tiled_gated(a_q, a_scale, tiled, self.out_features, s_perm, b_perm)

    def forward(self, input: Tensor):
        plan = self._plans.get(input.numel())      # pre-bound launch for this row count (re-validates buffers and input)
        if plan is not None:
            out = plan(input)
            if out is not None:
                return out
        if check_input(input) and not (input.requires_grad and torch.is_grad_enabled()):
            if self.act_quant:
                if self.in_features % 16 == 0:
                    # tile-major weights straight into the i8 MFMA operands; act_quant == "per_tensor": one scale for the
                    # whole activation tensor (chatglm_q/int8/qlinear.py:64-70) instead of one per row
                    return hip_ops.w8a8_forward_tiled(input, self.prepare()._tiled, self.out_features, self.weight_scale,
                                                      self.bias, per_tensor=self.act_quant == "per_tensor")
                return hip_ops.w8a8_forward(input, self.weight, self.weight_scale, self.bias)
            plan_out = [] if input.numel() else None
            if hip_ops.w8_tiled_supported(input, self.weight):
                # >= 3 rows: MFMA kernels on the tile-major derived copy (built lazily, keyed on the buffer's version)
                out = hip_ops.w8_forward_tiled(input, self.prepare()._tiled, self.out_features, self.weight_scale, self.bias,
                                               plan_out=plan_out, guards=(self.weight,))
            else:
                # bias fused after the output rounding: same two roundings as qlinear.py:90-93
                out = hip_ops.w8_forward(input, self.weight.t(), self.weight_scale, self.bias, plan_out=plan_out,
                                         guards=(self.weight,))
            if plan_out and plan_out[0] is not None:
                if len(self._plans) >= 16:
                    self._plans.clear()
                self._plans[input.numel()] = plan_out[0]
            return out
        out = dynamic_quant_matmul(input, self.weight.t(), self.weight_scale)
        if self.bias is not None:
            out = out + self.bias              # not in place: the Function's output may be a view
        return out

    @torch.no_grad()
    def prepare(self):
        """Build (or refresh) the tile-major derived copy now, e.g. before capturing a HIP graph."""
        key = _lib.buffer_key(self.weight)
        if self._tiled is None or self._tiled_key != key:
            self._tiled = hip_ops.tile_w8(self.weight) if self.weight.is_cuda and self.in_features % 16 == 0 else None
            self._tiled_key = key
            self._plans, self._fast = {}, {}
            _lib.bump_layout_epoch()
        return self

    @torch.no_grad()
    def gated(self, hidden: int):
        """Row-permuted copy (weight, scale, bias) of a first MLP projection (out_features = 2 * hidden) in
        (h_2t, h_2t+1, gate_2t, gate_2t+1) quads for the fused SiLU * gate epilogue (qlinear_w8_fwd_fused |
        QL_EPI_SILU_GATE).  Cached, keyed on the buffers' identity and version; never part of the state_dict."""
        if self.out_features != 2 * hidden:
            raise ValueError("gated layout needs out_features == 2 * hidden")
        key = _lib.buffer_key(self.weight, self.weight_scale, self.bias)
        if self._gated is None or self._gated_key != key:
            from ..int4.hip_ops import gate_interleave
            perm = gate_interleave(hidden, self.weight.device)
            self._gated = (self.weight.index_select(0, perm).contiguous(), self.weight_scale.index_select(0, perm).contiguous(),
                           None if self.bias is None else self.bias.index_select(0, perm).contiguous())
            self._gated_key = key
            self._gated_tiled, self._gated_tiled_key = None, None
            self._fast = {}
            _lib.bump_layout_epoch()
        return self._gated

    @torch.no_grad()
    def gated_tiled(self, hidden: int):
        """Tile-major copy of the gate-interleaved weights (prefill row counts: qlinear_w8_fwd_tiled_gated) + the permuted scale and
        bias; built on first use from ``gated()``, dropped with it."""
        w, s, b = self.gated(hidden)
        key = self._gated_key
        if getattr(self, "_gated_tiled", None) is None or self._gated_tiled_key != key:
            self._gated_tiled = hip_ops.tile_w8(w)
            self._gated_tiled_key = key
            _lib.bump_layout_epoch()
        return self._gated_tiled, s, b

    @torch.no_grad()
    def apply_weights_(self, q_weight: Tensor, scale: Tensor, bias: Tensor = None):
        self.weight.copy_(q_weight)
        self.weight_scale.copy_(scale)
        if bias is not None:
            self.bias.copy_(bias)
        self.invalidate()

    def extra_repr(self) -> str:
        return "in_features={}, out_features={}, bias={}".format(
            self.in_features, self.out_features, self.bias is not None)

    def reset_parameters(self):
        pass


class QEmbedding(nn.Module):
    def __init__(self, num_embeddings: int, embedding_dim: int, device=None, dtype=None):
        super().__init__()
        self.num_embeddings = num_embeddings
        self.embedding_dim = embedding_dim
        self.register_buffer("weight", torch.empty((num_embeddings, embedding_dim), device=device, dtype=torch.int8))
        self.register_buffer("weight_scale", torch.empty(embedding_dim, device=device, dtype=dtype))

    def forward(self, input: Tensor):
        if check_input(input) and self.weight.is_contiguous():
            return hip_ops.qembedding_w8(input, self.weight, self.weight_scale)
        return self.weight[input] * self.weight_scale

    @torch.no_grad()
    def apply_weights_(self, q_weight: Tensor, scale: Tensor):
        self.weight.copy_(q_weight)
        self.weight_scale.copy_(scale)

    def extra_repr(self) -> str:
        return "num_embeddings={}, embedding_dim={}".format(self.num_embeddings, self.embedding_dim)

    def reset_parameters(self):
        pass
