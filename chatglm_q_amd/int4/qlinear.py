"""int4 group-quantised linear / embedding modules (drop-in for chatglm_q/int4/qlinear.py).

Public surface kept from the reference: ``DEFAULT_GROUP_SIZE``, ``KERNEL_IMPL``, ``check_input``,
``unpack_int4``, ``DynamicQuantizeMatMul``, ``dynamic_quant_matmul``, ``DynamicQuantizeLinear``,
``QEmbedding`` with the same constructor signatures and the same registered buffers
(``weight`` (in/2, out) uint8, ``weight_scale`` (in/group, out), ``bias``), so a checkpoint loader that
``copy_``s into ``state_dict()`` (chatglm_q/loader.py:90-104) works unchanged.

Dispatch rule (same predicate as the reference, chatglm_q/int4/qlinear.py:47-50): GPU activations go
to the HIP kernels - and raise if libqlinear_hip.so is missing, there is no silent fallback on a
GPU - while CPU activations take the dense formula ``A @ unpack_int4(B, scale)``, which is the
reference's own CPU branch and BASELINE config 1's plumbing path.
"""
from __future__ import annotations

import os

import torch
from torch import Tensor, nn
from torch.autograd.function import FunctionCtx

from .. import _lib
from . import hip_ops
from .hip_ops import check_input

DEFAULT_GROUP_SIZE = 32  # chatglm_q/int4/qlinear.py:5

# "hip" when the C-ABI library is loadable, "none" otherwise (the reference's values are
# "triton" / "none", chatglm_q/int4/qlinear.py:13,17).
KERNEL_IMPL = "hip" if _lib.available() else "none"
# forward of an int4 module whose ``act_quant`` is set: None in the product; ``chatglm_q_amd.dev.experiments.enable_w4a8()`` installs
# the developer library's W4A8 route here (a recorded experiment: the product's forward never imports ``chatglm_q_amd.dev``)
ACT_QUANT_FORWARD = None

if KERNEL_IMPL == "none":
    print("libqlinear_hip.so not found: GPU tensors will raise; CPU tensors use the dense torch formula.")

# fp32 activations: largest row count served by the derived-layout GEMV kernel (above it the canonical
# kernel runs).  fp16 / bf16 use the derived layout for every row count (GEMV up to 4 rows, MFMA GEMM above).
PACKED_MAX_ROWS = int(os.environ.get("QLINEAR_PACKED_MAX_ROWS", "4"))
# "auto" (packed for decode shapes), "canonical" (never repack) - for A/B measurements.
W4_LAYOUT = os.environ.get("QLINEAR_W4_LAYOUT", "auto")


@torch.no_grad()
def unpack_int4(x: Tensor, x_scale: Tensor) -> Tensor:
    """Dense (K, N) dequantised weight in the scale's dtype (chatglm_q/int4/qlinear.py:20-33).

    Byte [k//2, n]: low nibble = row 2*(k//2), high nibble = the next row; value = (nibble - 8) * scale
    of the row's group.
    """
    K = x.shape[0] * 2
    G, N = x_scale.shape
    if x.shape[1] != N:
        raise AssertionError(f"N mismatch: {x.shape[1]} vs {N}")
    if K % G != 0:
        raise AssertionError(f"K={K}, G={G}")
    codes = torch.stack(((x & 0xF), (x >> 4)), dim=1).reshape(K, N).to(torch.int8) - 8
    return (codes.reshape(G, K // G, N) * x_scale[:, None, :]).reshape(K, N)


def _dense_matmul(A: Tensor, W: Tensor) -> Tensor:
    """The CPU branch's product (chatglm_q/int4/qlinear.py:50).  A function of its own so that the real-dimension CPU
    tests can swap in an fp32-accumulating twin: torch's fp16 CPU GEMM runs a 32 x 4096 x 27392 product in 40 s."""
    return A.matmul(W)


class DynamicQuantizeMatMul(torch.autograd.Function):
    """A: (m, k) float; B: (k//2, n) uint8; b_scale: (g, n) float (chatglm_q/int4/qlinear.py:36-68).

    Forward runs the HIP kernel for GPU tensors.  Backward: ``grad_A = grad_out @ dequant(B).T`` through
    qlinear_w4g32_bwd_input (fp16 / bf16 GPU tensors, group 32; the reference's transposed Triton kernel,
    chatglm_q/int4/triton_ops.py:142-264), the dense formula otherwise.
    """

    @staticmethod
    def forward(ctx: FunctionCtx, A: Tensor, B: Tensor, b_scale: Tensor):
        ctx.save_for_backward(A, B, b_scale)
        if check_input(A):
            return hip_ops.dynamic_quant_matmul_s4(A, B, b_scale)
        return _dense_matmul(A, unpack_int4(B, b_scale))

    @staticmethod
    def backward(ctx: FunctionCtx, grad_out: Tensor):
        A, B, b_scale = ctx.saved_tensors
        grad_A = None
        if ctx.needs_input_grad[0]:
            if check_input(A) and hip_ops.w4_grad_input_supported(grad_out, B, b_scale):
                grad_A = hip_ops.w4_grad_input(grad_out, B, b_scale)
            else:
                grad_A = grad_out.matmul(unpack_int4(B, b_scale).t())
        return grad_A, None, None

    @staticmethod
    def symbolic(g, A, B, b_scale):
        raise NotImplementedError()


def dynamic_quant_matmul(A: Tensor, B: Tensor, b_scale: Tensor) -> Tensor:
    return DynamicQuantizeMatMul.apply(A, B, b_scale)


class DynamicQuantizeLinear(nn.Module):
    def __init__(self, in_features: int, out_features: int, bias=True, group_size=DEFAULT_GROUP_SIZE,
                 device=None, dtype=None):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        if in_features % group_size != 0:
            raise AssertionError(f"in_features={in_features}, group_size={group_size}")
        self.group_size = group_size
        self.groups = in_features // group_size
        self.register_buffer("weight", torch.empty((in_features // 2, out_features), device=device, dtype=torch.uint8))
        self.register_buffer("weight_scale", torch.empty((self.groups, out_features), device=device, dtype=dtype))
        if bias:
            self.register_buffer("bias", torch.empty(out_features, device=device, dtype=dtype))
        else:
            self.register_buffer("bias", None)
        # derived streaming layout: a cache keyed on the canonical buffers' identity + version,
        # never registered, never saved (SURVEY.md 8b "Buffer ownership")
        # part 1 (column-major, the GEMVs) is built on the first GPU forward; part 2 (tile-major, the MFMA kernels) on the
        # first forward with more than hip_ops.GEMV_MAX_ROWS rows - a decode-only session never allocates it
        self._packed: Tensor | None = None
        self._packed_key = None
        self._tiled, self._tiled_key = None, None
        self._gated, self._gated_key = None, None
        self._gated_tiled, self._gated_tiled_key = None, None
        self._a8, self._a8_key = None, None
        # pre-bound launches (``_lib.make_plan``): forward() by input.numel(), the fused one-row launches of the decode step by
        # call site; built by the checked path after it served a call, dropped with the derived layouts
        self._plans: dict = {}
        self._fast: dict = {}
        # opt-in int8-activation path (W4A8, i8 MFMA): False | True (row-wise scales) | "per_tensor".  NOT bit-compatible
        # with the weight-only path (activation quantisation error ~1e-2 relative) and, since round 4, a developer experiment:
        # it needs libqlinear_hip_dev.so (include/qlinear_hip_dev.h; measured behind the weight-only GEMM at every shape)
        self.act_quant = False
        # low-footprint mode (round 5): drop_canonical() frees the canonical GPU buffers and keeps serving from the derived layouts;
        # part 1 (or the gate-interleaved part 1) is then the copy state_dict() is rebuilt from.  None = canonical buffers resident.
        self._dropped_key = None

    # -- derived layout -----------------------------------------------------------------------
    def _canonical_key(self):
        if self._dropped_key is not None:            # dropped: nobody can have written the canonical buffers since
            return self._dropped_key
        return _lib.buffer_key(self.weight, self.weight_scale)

    # -- low-footprint mode -------------------------------------------------------------------
    @property
    def canonical_dropped(self) -> bool:
        return self.__dict__.get("_dropped_key") is not None

    @torch.no_grad()
    def drop_canonical(self):
        """Free the canonical ``weight`` / ``weight_scale`` GPU buffers (the reference's 6 GB-class footprint, readme.md:72: with
        them AND the derived layouts resident a ChatGLM2-6B int4g32 model holds the weights 2 - 3.5 times).  The module keeps
        serving every GPU call from its derived layouts; the buffers become shape-only stand-ins (one element of storage), and
        ``state_dict()`` / ``save_pretrained`` rebuild the canonical tensors byte for byte from part 1 - the repack is a bijection
        (``qlinear_w4g32_unpack_gemv``) - so checkpoints are unchanged (chatglm_q/loader.py:90-104).  Whatever needs the canonical
        data itself (backward, the canonical-layout kernel for fp32 rows > 4, ``.to()``, loading) restores it first
        (``restore_canonical``).  Returns the bytes freed (0 when nothing was resident or the layout is not supported)."""
        if self.canonical_dropped or not self._packed_supported():
            return 0
        if self._packed is None and self._gated is None:
            self._build_part1()
        elif self._packed is not None and self._packed_key != self._canonical_key():
            self._build_part1()                      # a stale part 1 must not become the only copy
        elif self._packed is None and self._gated_key[0] != self._canonical_key():
            # only a gate-interleaved part 1 is resident (decode-only release) and the canonical buffers were written since it was
            # built: it must not become the only copy either - rebuild the plain part 1 from the current weights and let
            # the next fused step rebuild the gated copy
            self._gated = None
            self._build_part1()
        freed = self.weight.numel() * self.weight.element_size() + self.weight_scale.numel() * self.weight_scale.element_size()
        key = self._canonical_key()
        dev, sdt = self.weight.device, self.weight_scale.dtype
        shape_w, shape_s = tuple(self.weight.shape), tuple(self.weight_scale.shape)
        self.weight = torch.empty(1, dtype=torch.uint8, device=dev).expand(shape_w)
        self.weight_scale = torch.empty(1, dtype=sdt, device=dev).expand(shape_s)
        self._dropped_key = key
        return freed

    @torch.no_grad()
    def _rebuild_canonical(self):
        """(weight, weight_scale) as fresh tensors from the resident derived layout."""
        K, N, dt = self.in_features, self.out_features, self.weight_scale.dtype
        if self._packed is not None:
            return hip_ops.unpack_w4g32_gemv(self._packed, N, K, dt)
        if self._gated is not None:                  # a first MLP projection whose plain part 1 was released: undo the column order
            perm = hip_ops.gate_interleave(N // 2, self._gated[0].device)
            w, sc = hip_ops.unpack_w4g32_gemv(self._gated[0], N, K, dt)
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(N, device=perm.device)
            return w.index_select(1, inv), sc.index_select(1, inv)
        raise RuntimeError("canonical buffers were dropped and no part 1 is resident to rebuild them from")

    @torch.no_grad()
    def restore_canonical(self):
        """Undo ``drop_canonical``: the canonical buffers are rebuilt from part 1 and resident again (derived layouts stay valid)."""
        if not self.canonical_dropped:
            return self
        w, sc = self._rebuild_canonical()
        key_before = self._dropped_key
        self._dropped_key = None
        self.weight, self.weight_scale = w, sc
        # the derived layouts were built from exactly these bytes: re-key them to the new buffers instead of rebuilding
        key = self._canonical_key()
        for part in ("packed", "tiled", "a8"):
            if getattr(self, "_" + part) is not None and getattr(self, "_" + part + "_key") == key_before:
                setattr(self, "_" + part + "_key", key)
        if self._gated is not None and self._gated_key[0] == key_before:
            same = self._gated_tiled_key == self._gated_key
            self._gated_key = (key, self._gated_key[1])
            if same:
                self._gated_tiled_key = self._gated_key
        return self

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        if not self.canonical_dropped:
            return super()._save_to_state_dict(destination, prefix, keep_vars)
        w, sc = self._rebuild_canonical()            # temporaries: the module stays in low-footprint mode
        destination[prefix + "weight"] = w
        destination[prefix + "weight_scale"] = sc
        if self.bias is not None:
            destination[prefix + "bias"] = self.bias if keep_vars else self.bias.detach()

    def invalidate(self):
        """Drop every derived layout; the next GPU forward rebuilds them from the canonical buffers.  Needed only
        after a write the version counter cannot see (``weight.data.copy_``, raw pointers, inference tensors)."""
        self.restore_canonical()                    # low-footprint mode: part 1 is the only copy - the canonical buffers come back first
        self._packed, self._packed_key = None, None
        self._tiled, self._tiled_key = None, None
        self._gated, self._gated_key = None, None
        self._gated_tiled, self._gated_tiled_key = None, None
        self._a8, self._a8_key = None, None
        self._plans, self._fast = {}, {}
        _lib.bump_layout_epoch()
        return self

    def release(self, *parts: str):
        """Free derived layouts a deployment no longer needs (they are caches: anything released is rebuilt on demand).
        ``parts``: any of "packed" (part 1, the GEMVs), "tiled" (part 2, the MFMA kernels), "gated" (gate-interleaved part 1 of
        a first MLP projection), "gated_tiled", "a8" (W4A8 copy).  E.g. a decode-only session keeps "packed" for
        qkv_proj / o_proj / w_out / lm_head and only "gated" for w_in: ``w_in.release("packed", "tiled", "gated_tiled")``."""
        if self.canonical_dropped and not any(getattr(self, "_" + p) is not None and p not in parts for p in ("packed", "gated")):
            raise ValueError("canonical buffers are dropped: part 1 (or the gate-interleaved part 1) is the only copy of the weights; "
                             "restore_canonical() first")
        for part in parts:
            if part not in ("packed", "tiled", "gated", "gated_tiled", "a8"):
                raise ValueError(f"unknown derived layout {part!r}")
            setattr(self, "_" + part, None)
            setattr(self, "_" + part + "_key", None)
        self._plans, self._fast = {}, {}
        _lib.bump_layout_epoch()
        return self

    def derived_nbytes(self) -> dict:
        """Resident bytes of each derived layout (0 when not built) next to the canonical buffers'."""
        def nb(t):
            t = t[0] if isinstance(t, tuple) else t
            return 0 if t is None else t.numel() * t.element_size()
        out = {k: nb(getattr(self, "_" + k)) for k in ("packed", "tiled", "gated", "gated_tiled", "a8")}
        out["canonical"] = 0 if self.canonical_dropped else nb(self.weight) + nb(self.weight_scale)
        return out

    def __setattr__(self, name, value):
        super().__setattr__(name, value)
        if name in ("weight", "weight_scale", "bias", "act_quant") and "_plans" in self.__dict__:
            self._plans, self._fast = {}, {}        # a replaced buffer / another path: the pre-bound launches are stale
            _lib.bump_layout_epoch()

    def _load_from_state_dict(self, *args, **kwargs):
        self.restore_canonical()                # real buffers to copy into (a partial load keeps the rest of the data)
        super()._load_from_state_dict(*args, **kwargs)
        self.invalidate()

    def _apply(self, fn, *args, **kwargs):      # .to() / .cuda() / .half(): new storage (whose address may be a reused one)
        self.restore_canonical()                # the derived layouts do not travel: the canonical data must
        self.invalidate()
        return super()._apply(fn, *args, **kwargs)

    def _packed_supported(self) -> bool:
        if self.canonical_dropped:              # only a supported layout can have been dropped
            return True
        return (W4_LAYOUT != "canonical" and self.group_size == 32 and self.weight.is_cuda
                and self.weight.is_contiguous() and self.weight_scale.is_contiguous())

    @torch.no_grad()
    def prepare(self, rows: int = 1):
        """Build (or refresh) the derived layout now, e.g. before capturing a HIP graph: part 1, and part 2 as well when
        ``rows`` (the row count the module is about to serve) needs the MFMA kernels."""
        if not self._packed_supported():
            self._packed, self._packed_key = None, None
            self._tiled, self._tiled_key = None, None
            return self
        self._build_part1()
        dt = self.weight_scale.dtype
        if hip_ops.rows_on_tiled(rows, self.out_features, self.in_features, dt, _lib.strict_for(dt)):
            self.tiled()
        return self

    def _build_part1(self):
        key = self._canonical_key()
        if self._packed is None or self._packed_key != key:
            # low-footprint mode with only the gate-interleaved part 1 resident: the canonical bytes are rebuilt as temporaries
            w, sc = self._rebuild_canonical() if self.canonical_dropped else (self.weight, self.weight_scale)
            self._packed = hip_ops.repack_w4g32_gemv(w, sc)
            self._packed_key = key
            self._plans, self._fast = {}, {}
            _lib.bump_layout_epoch()

    @torch.no_grad()
    def tiled(self) -> Tensor:
        """Part 2 of the derived layout (tile-major, the few-row and MFMA GEMM kernels; fp16 / bf16), built from part 1
        on first use and cached like it."""
        key = self._canonical_key()
        if self._tiled is None or self._tiled_key != key:
            self._build_part1()                     # not prepare(): with QLINEAR_GEMV_MAX_ROWS=0 that would call tiled() again
            self._tiled = hip_ops.tile_w4g32(self._packed, self.out_features, self.in_features, self.weight_scale.dtype)
            self._tiled_key = key
            _lib.bump_layout_epoch()
        return self._tiled

    @torch.no_grad()
    def gated_packed(self, hidden: int):
        """Derived layout of a first MLP projection (out_features = 2 * hidden) with its columns reordered to
        (h_2t, h_2t+1, gate_2t, gate_2t+1) quads for the fused SiLU * gate epilogue
        (qlinear_w4g32_fwd_packed_fused | QL_EPI_SILU_GATE).  Returns (packed, bias_or_None); cached like
        ``_packed`` and never part of the state_dict."""
        if self.out_features != 2 * hidden or not self._packed_supported():
            raise ValueError("gated layout needs a supported (K, 2 * hidden) int4g32 weight")
        key = (self._canonical_key(), _lib.buffer_key(self.bias))
        if getattr(self, "_gated", None) is None or self._gated_key != key:
            if self.canonical_dropped:              # the permuted copy is built from the canonical columns: bring them back for the build
                self.restore_canonical()
                try:
                    return self.gated_packed(hidden)
                finally:
                    self.drop_canonical()
            perm = hip_ops.gate_interleave(hidden, self.weight.device)
            packed = hip_ops.repack_w4g32_gemv(self.weight.index_select(1, perm), self.weight_scale.index_select(1, perm))
            bias = self.bias.index_select(0, perm) if self.bias is not None else None
            self._gated, self._gated_key = (packed, bias), key
            self._gated_tiled, self._gated_tiled_key = None, None
            self._fast = {}
            _lib.bump_layout_epoch()
        return self._gated

    @torch.no_grad()
    def gated_tiled(self, hidden: int):
        """Part 2 of the gate-interleaved copy (the row counts ``rows_on_tiled`` gives to part 2: qlinear_w4g32_fwd_tiled_gated), built on first
        use.  Returns (tiled, bias_or_None)."""
        packed, bias = self.gated_packed(hidden)
        key = self._gated_key
        if self._gated_tiled is None or self._gated_tiled_key != key:
            self._gated_tiled = hip_ops.tile_w4g32(packed, self.out_features, self.in_features, self.weight_scale.dtype)
            self._gated_tiled_key = key
            _lib.bump_layout_epoch()
        return self._gated_tiled, bias

    def forward(self, input: Tensor):
        # fast path: a pre-bound launch for this row count (built below by the checked path); it re-validates input layout,
        # buffer identity and version counters itself and returns None when anything moved
        plan = self._plans.get(input.numel())
        if plan is not None:
            out = plan(input)
            if out is not None:
                return out
        if check_input(input):
            if input.requires_grad and torch.is_grad_enabled():
                self.restore_canonical()        # the autograd function saves (and its backward reads) the canonical buffers
                out = dynamic_quant_matmul(input, self.weight, self.weight_scale)
                if self.bias is not None:
                    out = out + self.bias      # not in place: the Function's output may be a view
                return out
            if self.act_quant:            # W4A8: a recorded experiment of the developer library; the product never imports it
                if ACT_QUANT_FORWARD is None:
                    raise RuntimeError("int4 DynamicQuantizeLinear.act_quant (W4A8) is served by the developer library only: call "
                                       "chatglm_q_amd.dev.experiments.enable_w4a8() first, or leave act_quant = False")
                out = ACT_QUANT_FORWARD(self, input)
                if out is not None:
                    return out
            rows = input.numel() // max(input.shape[-1], 1)
            packed = tiled = None
            half = input.dtype in (torch.float16, torch.bfloat16)
            if (rows <= PACKED_MAX_ROWS or half) and self._packed_supported() and input.dtype == self.weight_scale.dtype:
                if half and hip_ops.rows_on_tiled(rows, self.out_features, self.in_features, input.dtype, _lib.strict_for(input.dtype)):
                    tiled = self.tiled()
                else:
                    packed = self.prepare()._packed
            # bias is added inside the kernel epilogue AFTER the rounding to the output dtype,
            # i.e. the same two roundings as "out = matmul(); out += bias" (qlinear.py:90-94)
            if packed is None and tiled is None:
                self.restore_canonical()        # the canonical-layout kernel (fp32 rows > 4, odd shapes) reads the canonical buffers
            plan_out = [] if (packed is not None or tiled is not None) and rows > 0 and not self.act_quant else None
            out = hip_ops.w4_forward(input, self.weight, self.weight_scale, self.bias, packed, tiled=tiled, plan_out=plan_out)
            if plan_out and plan_out[0] is not None:
                if len(self._plans) >= 16:                 # row counts seen so far (prefill chunks): keep the table small
                    self._plans.clear()
                self._plans[input.numel()] = plan_out[0]
            return out
        out = dynamic_quant_matmul(input, self.weight, self.weight_scale)
        if self.bias is not None:
            out += self.bias
        return out

    @torch.no_grad()
    def apply_weights_(self, q_weight: Tensor, scale: Tensor, bias: Tensor = None):
        self.restore_canonical()
        self.weight.copy_(q_weight)
        self.weight_scale.copy_(scale)
        if bias is not None:
            self.bias.copy_(bias)
        self.invalidate()

    def extra_repr(self) -> str:
        return "in_features={}, out_features={}, group_size={}, bias={}".format(
            self.in_features, self.out_features, self.group_size, self.bias is not None)

    def reset_parameters(self):
        pass


class QEmbedding(nn.Module):
    """int4 embedding; packing runs along the vocabulary axis (chatglm_q/int4/qlinear.py:111-142)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, group_size=DEFAULT_GROUP_SIZE, device=None, dtype=None):
        super().__init__()
        self.num_embeddings = num_embeddings
        self.embedding_dim = embedding_dim
        if num_embeddings % group_size != 0:
            raise AssertionError(f"num_embeddings={num_embeddings}, group_size={group_size}")
        self.group_size = group_size
        self.groups = num_embeddings // group_size
        self.register_buffer("weight", torch.empty((num_embeddings // 2, embedding_dim), device=device, dtype=torch.uint8))
        self.register_buffer("weight_scale", torch.empty((self.groups, embedding_dim), device=device, dtype=dtype))

    def forward(self, input: Tensor):
        if check_input(input) and self.weight.is_contiguous() and self.weight_scale.is_contiguous():
            return hip_ops.qembedding_w4(input, self.weight, self.weight_scale, self.group_size)
        rows = self.weight[input // 2]
        shifts = ((input % 2) * 4)[..., None].to(rows.dtype)
        codes = ((rows >> shifts) & 0xF).to(torch.int8) - 8
        return codes * self.weight_scale[input // self.group_size]

    @torch.no_grad()
    def apply_weights_(self, q_weight: Tensor, scale: Tensor):
        self.weight.copy_(q_weight)
        self.weight_scale.copy_(scale)

    def extra_repr(self) -> str:
        return "num_embeddings={}, embedding_dim={}, group_size={}".format(
            self.num_embeddings, self.embedding_dim, self.group_size)

    def reset_parameters(self):
        pass
