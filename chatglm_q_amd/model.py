"""ChatGLM2 decoder graph - the CALLER of the quantized-linear path (harness for BASELINE configs 4 and 5).

This is the build's own counterpart of the reference's model graph (chatglm_q/model.py:90-392); it exists so
that the QLinear kernels can be measured where they are used: 28 x (qkv_proj, o_proj, w_in, w_out) +
lm_head = 113 calls per decoded token.  Numerical semantics follow the reference (cited per block) and
are pinned by tests/golden/tiny_model.npz; parameter / buffer names match it, so a reference checkpoint's
state_dict loads key for key.

Differences in structure (not in results): one preallocated key/value cache per layer that a decode step
writes in place (the reference grows its cache with torch.cat every token, chatglm_q/model.py:151-155),
which makes a decode step shape-static and HIP-graph capturable (`chatglm_q_amd/decoder.py`).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn


# Route switches of the decode / prefill graph.  Plain module attributes (no environment variables): every one has a fallback that
# computes the same function, the tests flip them with monkeypatch to compare the two routes, tools/ set them for A/B timings.
FUSED_DECODE_OPS = True     # False: every op around the QLinear calls in plain torch
ROWS_FUSED_MAX = 2          # batched decode at 2..ROWS_FUSED_MAX rows: residual add + RMSNorm inside the projections' launches
PREQUANT = True             # int8-activation modules: RMSNorm / SiLU * gate emit the int8 rows + scales themselves
PREFILL_ATTENTION = True    # many-position attention as one launch per layer (False: two batched GEMMs around masked_softmax)
GATED_PREFILL = True        # prefill: w_in on its gate-interleaved copy, SiLU * gate in the 256 x 256-tile GEMM's epilogue
RESIDUAL_PREFILL = True     # prefill: `hidden + o_proj(...)` / `hidden + w_out(...)` with the add in the GEMM's epilogue
# developer experiments (chatglm_q_amd/dev/experiments.py: enable_mlp_engine / enable_mlp_pair) plug a one-launch MLP in here:
# callable (model, cache, ffn, ffn_ln, h) -> new hidden state or None.  None in the product.
MLP_HOOK = None


@dataclass
class ChatGLM2Config:
    # field names and defaults as chatglm_q/model.py:9-22 (ChatGLM2-6B)
    hidden_size: int = 4096
    inner_hidden_size: int = 13696
    head_hidden_size: int = 128
    num_multi_query_groups: int = 2
    num_attention_heads: int = 32
    num_layers: int = 28
    vocab_size: int = 65024
    dropout_rate: float = 0.0
    layernorm_epsilon: float = 1e-05
    max_sequence_length: int = 8192


# Patch points, as in the reference (chatglm_q/model.py:76-87): a loader rebinds these two names to the
# quantized classes before constructing the model (chatglm_q/loader.py:41-66).
class Linear(nn.Linear):
    def forward(self, x: Tensor) -> Tensor:
        return F.linear(x, self.weight.type_as(x), None if self.bias is None else self.bias.type_as(x))

    def reset_parameters(self):
        pass


class Embedding(nn.Embedding):
    def reset_parameters(self):
        pass


def rotary_table(d_head: int, length: int, theta: float = 10000.0) -> Tensor:
    """(length, d_head/2, 2) table of (cos, sin): the first d_head/4 pairs rotate, the rest pass through
    as (1, 0) - "half of the head_dim bypassed" (chatglm_q/model.py:35-44)."""
    half = d_head // 2
    inv = 1.0 / (theta ** (torch.arange(0, half, 2).float() / half))
    ang = torch.outer(torch.arange(length).float(), inv)
    rot = torch.stack((torch.cos(ang), torch.sin(ang)), dim=-1)
    keep = torch.stack((torch.ones_like(ang), torch.zeros_like(ang)), dim=-1)
    return torch.cat((rot, keep), dim=-2)


def rotate_pairs(x: Tensor, cs: Tensor) -> Tensor:
    """Complex multiply of interleaved (re, im) pairs by (cos, sin) (chatglm_q/model.py:48-59).
    x: (..., d/2, 2); cs broadcastable (..., d/2, 2).  Evaluated in fp32, returned in x's dtype."""
    xr, xi = x[..., 0].float(), x[..., 1].float()
    c, s = cs[..., 0].float(), cs[..., 1].float()
    return torch.stack((xr * c - xi * s, xr * s + xi * c), dim=-1).flatten(-2).to(x.dtype)


class RMSNorm(nn.Module):
    """x * rsqrt(mean(x^2) + eps) in fp32, cast back, times weight (chatglm_q/model.py:62-73)."""

    def __init__(self, dim: int, eps: float = 1e-5, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim, dtype=dtype))
        self.eps = eps

    def forward(self, x: Tensor) -> Tensor:
        xf = x.float()
        return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)).type_as(x) * self.weight


class KVCache:
    """Preallocated per-layer key/value store: (batch, capacity, groups, d_head)."""

    def __init__(self, n_layers: int, batch: int, capacity: int, groups: int, d_head: int, device, dtype):
        self.k = [torch.zeros(batch, capacity, groups, d_head, device=device, dtype=dtype) for _ in range(n_layers)]
        self.v = [torch.zeros(batch, capacity, groups, d_head, device=device, dtype=dtype) for _ in range(n_layers)]
        self.capacity = capacity
        self.length = 0            # host-side count of valid positions (not used inside captured graphs)
        self.att_plans: dict = {}  # layer -> (layout epoch, operand addresses, pre-bound attention launch); model._step_one_row

    def as_tuples(self):
        """Reference-shaped view: tuple of (k, v) each (batch, length, groups, 1, d_head)."""
        n = self.length
        return tuple((k[:, :n].unsqueeze(3), v[:, :n].unsqueeze(3)) for k, v in zip(self.k, self.v))


class ChatGLM2Attention(nn.Module):
    """Multi-query attention (chatglm_q/model.py:90-177)."""

    def __init__(self, n_state: int, n_head: int, d_head: int, n_groups: int, layer_idx: int, dtype=None):
        super().__init__()
        self.n_head, self.d_head, self.n_groups, self.layer_idx = n_head, d_head, n_groups, layer_idx
        self.qkv_proj = Linear(n_state, d_head * (n_head + 2 * n_groups), bias=True, dtype=dtype)
        self.o_proj = Linear(d_head * n_head, n_state, bias=False, dtype=dtype)

    def project(self, x: Tensor, cs: Tensor):
        """qkv projection + rotary.  Returns q (B,S,G,H/G,D), k (B,S,G,D), v (B,S,G,D)."""
        B, S, _ = x.shape
        H, D, G = self.n_head, self.d_head, self.n_groups
        q, k, v = torch.split(self.qkv_proj(x), [D * H, D * G, D * G], dim=-1)
        q = rotate_pairs(q.view(B, S, G, H // G, D // 2, 2), cs[:, :, None, None])
        k = rotate_pairs(k.view(B, S, G, D // 2, 2), cs[:, :, None])
        return q, k, v.view(B, S, G, D)

    def attend(self, x_dtype, q: Tensor, k_all: Tensor, v_all: Tensor, mask: Optional[Tensor]) -> Tensor:
        return self.o_proj(self.core(x_dtype, q, k_all, v_all, mask))

    def core(self, x_dtype, q: Tensor, k_all: Tensor, v_all: Tensor, mask: Optional[Tensor]) -> Tensor:
        """softmax((q / sqrt(D)) k^T + mask) v with the softmax in fp32 (chatglm_q/model.py:157-175).
        q (B,S,G,Hg,D); k_all, v_all (B,T,G,D); mask (B,S,T) additive.  Returns (B,S,H*D) (before o_proj)."""
        B, S, G, Hg, D = q.shape
        T = k_all.shape[1]
        # per sequence: one batched GEMM over the G groups with a group's Hg heads stacked along the rows; K^T and
        # V are strided VIEWS of the cache (batch stride D, leading dimension G*D: what BLAS takes as-is), so no
        # transposed / broadcast copies are made, and the fp32 softmax takes the cast inside its kernel
        qh = (q / math.sqrt(D)).permute(0, 2, 3, 1, 4).reshape(B, G, Hg * S, D)
        # autograd (labels -> loss -> backward, chatglm_q/model.py:384-390): the fused softmax is a raw launch with no
        # graph and bmm(out=) cannot be differentiated - those are inference-only shortcuts
        grad = torch.is_grad_enabled() and (q.requires_grad or k_all.requires_grad or v_all.requires_grad)
        fused = (FUSED_DECODE_OPS and not grad and q.is_cuda and q.dtype in (torch.float16, torch.bfloat16)
                 and q.dtype == x_dtype)
        out = None if grad else torch.empty((B, G, Hg * S, D), device=q.device, dtype=x_dtype)
        rows = []
        for b in range(B):
            qk = torch.bmm(qh[b], k_all[b].permute(1, 2, 0)).view(G, Hg, S, T)
            if fused:
                from . import fused_ops
                p = fused_ops.masked_softmax(qk, None if mask is None else mask[b]).view(G, Hg * S, T)   # add + softmax + cast
            else:
                if mask is not None:
                    qk = qk + mask[b][None, None]
                p = F.softmax(qk, dim=-1, dtype=torch.float32).to(x_dtype).view(G, Hg * S, T)
            if grad:
                rows.append(torch.bmm(p, v_all[b].permute(1, 0, 2)))
            else:
                torch.bmm(p, v_all[b].permute(1, 0, 2), out=out[b])
        if grad:
            out = torch.stack(rows)
        return out.view(B, G, Hg, S, D).permute(0, 3, 1, 2, 4).reshape(B, S, G * Hg * D)


class GatedFeedForward(nn.Module):
    """w_out(silu(h) * gate), (h, gate) = split(w_in(x)) (chatglm_q/model.py:180-201)."""

    def __init__(self, dim: int, hidden_dim: int, dtype=None):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.w_in = Linear(dim, hidden_dim * 2, bias=False, dtype=dtype)
        self.w_out = Linear(hidden_dim, dim, bias=False, dtype=dtype)

    def forward(self, x: Tensor) -> Tensor:
        h, gate = torch.split(self.w_in(x), self.hidden_dim, dim=-1)
        return self.w_out(F.silu(h) * gate)


class ChatGLM2Block(nn.Module):
    """Pre-norm residual block (chatglm_q/model.py:204-246)."""

    def __init__(self, layer_idx: int, config: ChatGLM2Config, dtype=None):
        super().__init__()
        self.layer_idx = layer_idx
        self.attn_ln = RMSNorm(config.hidden_size, config.layernorm_epsilon, dtype)
        self.attn = ChatGLM2Attention(config.hidden_size, config.num_attention_heads, config.head_hidden_size,
                                      config.num_multi_query_groups, layer_idx, dtype)
        self.ffn_ln = RMSNorm(config.hidden_size, config.layernorm_epsilon, dtype)
        self.ffn = GatedFeedForward(config.hidden_size, config.inner_hidden_size, dtype)


class ChatGLM2Model(nn.Module):
    def __init__(self, config: ChatGLM2Config, dtype=None):
        super().__init__()
        self.config = config
        self.word_embedding = Embedding(config.vocab_size, config.hidden_size, dtype=dtype)
        self.layers = nn.ModuleList([ChatGLM2Block(i, config, dtype) for i in range(config.num_layers)])
        self.final_ln = RMSNorm(config.hidden_size, config.layernorm_epsilon, dtype)
        self.lm_head = Linear(config.hidden_size, config.vocab_size, bias=False, dtype=dtype)
        # the reference keeps the table in the model dtype (chatglm_q/model.py:268-270)
        # One row more than the reference's table: positions are 1-based (cumsum of the mask), so a cache of
        # max_sequence_length rows reaches position max_sequence_length - where the reference's F.embedding raises; the
        # HIP kernels clamp positions to [0, capacity] instead (no table length crosses the ABI) and must find that row.
        table = rotary_table(config.head_hidden_size, config.max_sequence_length + 1).to(dtype=dtype)
        self.register_buffer("freqs_cis_cache", table.view(config.max_sequence_length + 1, -1), persistent=False)

    # -- cache helpers ---------------------------------------------------------------------------
    def new_cache(self, batch: int, capacity: int, device=None, dtype=None) -> KVCache:
        c = self.config
        p = self.final_ln.weight
        return KVCache(c.num_layers, batch, capacity, c.num_multi_query_groups, c.head_hidden_size,
                       device or p.device, dtype or p.dtype)

    def _rotary(self, position_ids: Tensor) -> Tensor:
        d2 = self.config.head_hidden_size // 2
        return F.embedding(position_ids, self.freqs_cis_cache).view(*position_ids.shape, d2, 2)

    # -- one pass over `n_new` positions against a preallocated cache ------------------------------
    def step(self, input_ids: Tensor, cache: KVCache, write_index: Tensor, position_ids: Tensor, mask: Tensor,
             last_only: bool = False, kv_len: Optional[int] = None) -> Tensor:
        """Shape-static forward: embeds input_ids (B, S), writes the new keys/values into the cache rows
        `write_index` (S,), attends over the WHOLE cache capacity under the additive `mask` (B, S, capacity)
        and returns logits.  Every argument is a device tensor, so the call can be captured in a HIP graph
        and replayed with new contents.  `last_only` evaluates lm_head for the final position only (the
        decode loop uses nothing else, chatglm_q/decoder.py:85).  `kv_len` (a HOST integer, so not for captured
        steps) promises that every cache row >= kv_len is masked: attention then reads only the first kv_len
        rows - the same sums, since a masked column's probability is exactly 0 (prefill uses this)."""
        h = self.word_embedding(input_ids)
        fused = FUSED_DECODE_OPS and h.is_cuda
        if fused and h.shape[1] == 1:
            kv_len = None                              # the one-position kernels take the whole cache + the FULL-capacity mask
        if kv_len is not None:
            mask = mask[..., :kv_len]
        if fused:
            return self._step_fused(h, cache, write_index, position_ids, mask, last_only, kv_len)
        cs = self._rotary(position_ids)
        for i, layer in enumerate(self.layers):
            q, k, v = layer.attn.project(layer.attn_ln(h), cs)
            cache.k[i].index_copy_(1, write_index, k)
            cache.v[i].index_copy_(1, write_index, v)
            h = h + layer.attn.attend(h.dtype, q, cache.k[i][:, :kv_len], cache.v[i][:, :kv_len], mask)
            h = h + layer.ffn(layer.ffn_ln(h))
        h = self.final_ln(h[:, -1:] if last_only else h)
        return self.lm_head(h)

    def _step_fused(self, h: Tensor, cache: KVCache, write_index: Tensor, position_ids: Tensor, mask: Tensor,
                    last_only: bool, kv_len: Optional[int] = None) -> Tensor:
        """Same graph with the small ops around the QLinear calls fused into single HIP launches
        (csrc/decode_ops.hip): RMSNorm; split + rotary + cache write; single-position attention; SiLU * gate."""
        from . import fused_ops as F_
        c = self.config
        H, G, D = c.num_attention_heads, c.num_multi_query_groups, c.head_hidden_size
        B, S, _ = h.shape
        if S == 1 and (kv_len is not None or mask.shape[-1] != cache.capacity):
            raise ValueError("one-position steps take the full-capacity mask (B, 1, capacity) and no kv_len")
        mask = mask.contiguous()
        if B * S == 1 and h.dtype in (torch.float16, torch.bfloat16):
            kind = self._one_row_kind(h.dtype)
            if kind:
                return self._step_one_row(h, cache, write_index, position_ids, mask, kind)
        # batched decode at 2 rows: residual add + RMSNorm inside the projections' launches (the entry point serves 2..4 rows, but the
        # prologue handles two rows per pass: measured 1.45 -> 1.41 ms per step at 2 rows, 1.56 -> 1.59 / 1.61 -> 1.64 at 3 / 4)
        few = S == 1 and 2 <= B <= ROWS_FUSED_MAX and h.dtype in (torch.float16, torch.bfloat16)
        from .int8.qlinear import DynamicQuantizeLinear as Q8

        def prequant(mod):
            # int8-activation modules (act_quant, row-wise scales): the producer in front emits the int8 rows + scales itself
            # (qlinear_rmsnorm_quant_i8 / qlinear_silu_mul_quant_i8) - no quantiser launch, no 16-bit copy of the row
            return (PREQUANT and isinstance(mod, Q8) and mod.act_quant is True and mod.in_features % 16 == 0
                    and mod.in_features <= 16384      # the quantising producers hold a row in registers (decode_ops.hip)
                    and h.dtype in (torch.float16, torch.bfloat16) and mod.weight_scale.dtype == h.dtype)
        # many-position attention in one launch per layer for ChatGLM2's head geometry; the tile flags (which key tiles a block of
        # query rows can skip / needs no mask for) come from one pass over the mask, shared by all layers
        one_launch = (S > 1 and PREFILL_ATTENTION and F_.prefill_attention_supported(h.dtype, H, G, D) and mask.dtype == torch.float32
                      and mask.dim() == 3 and cache.k[0].is_contiguous())
        if one_launch:
            T_kv = mask.shape[-1]
            if mask.shape[0] != B:
                mask = mask.expand(B, S, T_kv).contiguous()
            tile_flags = F_.attention_tile_flags(mask)
        delta = None                                   # pending residual contribution of the previous sub-block
        for i, layer in enumerate(self.layers):
            qkv = self._rows_fused(layer.attn.qkv_proj, h, delta, layer.attn_ln) if few else None
            if qkv is not None:                        # h += delta, RMSNorm and qkv_proj: one launch
                qkv, hn = qkv
                h = hn if hn is not None else h
            elif prequant(layer.attn.qkv_proj):
                h, a_q, a_s, _ = F_.rmsnorm_quant(h, layer.attn_ln.weight, layer.attn_ln.eps, delta)
                qkv = layer.attn.qkv_proj.forward_quantized(a_q, a_s).view(B, S, -1)
            else:
                if delta is None:
                    x = F_.rmsnorm(h, layer.attn_ln.weight, layer.attn_ln.eps)
                else:                                  # h += delta and the next norm in one launch
                    h, x = F_.add_rmsnorm(h, delta, layer.attn_ln.weight, layer.attn_ln.eps)
                qkv = layer.attn.qkv_proj(x)
            if S == 1:                                 # rotary + cache write + attention: one launch
                att = F_.decode_attention_rope(qkv, self.freqs_cis_cache, position_ids, write_index,
                                               cache.k[i], cache.v[i], mask, H, G, D)
            else:
                q = F_.rope_kv_write(qkv, self.freqs_cis_cache, position_ids, write_index,
                                     cache.k[i], cache.v[i], H, G, D)
                if one_launch:                         # many positions: scores never leave the chip (csrc/prefill_attention.hip)
                    att = F_.prefill_attention(q.view(B, S, H * D), cache.k[i], cache.v[i], mask, tile_flags, T_kv, H, G, D)
                else:
                    att = layer.attn.core(h.dtype, q.view(B, S, G, H // G, D), cache.k[i][:, :kv_len], cache.v[i][:, :kv_len], mask)
            many = RESIDUAL_PREFILL and B * S >= 1024       # prefill row counts: residual adds inside the GEMMs' epilogues
            hn = self._residual_many(layer.attn.o_proj, att, h) if many else None
            if hn is not None:                         # h + o_proj(att) came out of the GEMM
                h, o = hn, None
            else:
                o = layer.attn.o_proj(att)
            y = self._rows_fused(layer.ffn.w_in, h, o, layer.ffn_ln, gate_hidden=layer.ffn.hidden_dim) if few else None
            if y is not None:                          # h += o, RMSNorm, w_in and SiLU * gate: one launch
                y, h = y
            else:
                if prequant(layer.ffn.w_in):
                    h, a_q, a_s, _ = F_.rmsnorm_quant(h, layer.ffn_ln.weight, layer.ffn_ln.eps, o)
                    y = layer.ffn.w_in.forward_quantized_gated(a_q, a_s, layer.ffn.hidden_dim) if GATED_PREFILL and B * S >= 1024 else None
                    if y is not None:                  # SiLU * gate came out of the int8 x int8 GEMM: (B, S, hidden)
                        y, u = y.view(B, S, -1), None
                    else:
                        u = layer.ffn.w_in.forward_quantized(a_q, a_s).view(B, S, -1)
                else:
                    if o is None:
                        x = F_.rmsnorm(h, layer.ffn_ln.weight, layer.ffn_ln.eps)
                    else:
                        h, x = F_.add_rmsnorm(h, o, layer.ffn_ln.weight, layer.ffn_ln.eps)
                    y = self._gated_w_in(layer.ffn, x) if (S == 1 and 2 <= B <= 32) or (GATED_PREFILL and B * S >= 1024) else None   # SiLU * gate in w_in's epilogue
                    u = layer.ffn.w_in(x) if y is None else None
                if y is None and prequant(layer.ffn.w_out):
                    a_q, a_s, _ = F_.silu_mul_quant(u, layer.ffn.hidden_dim)
                    delta = layer.ffn.w_out.forward_quantized(a_q, a_s).view(B, S, -1)
                    continue
                if y is None:
                    y = F_.silu_mul(u, layer.ffn.hidden_dim)
            hn = self._residual_many(layer.ffn.w_out, y, h) if many else None
            if hn is not None:                         # h + w_out(y) came out of the GEMM: nothing pending
                h, delta = hn, None
            else:
                delta = layer.ffn.w_out(y)
        if last_only:
            h, delta = h[:, -1:], None if delta is None else delta[:, -1:]
        out = self._rows_fused(self.lm_head, h, delta, self.final_ln, want_hout=False) if few else None
        if out is not None:
            return out[0]
        if delta is None:
            x = F_.rmsnorm(h, self.final_ln.weight, self.final_ln.eps)
        else:
            _, x = F_.add_rmsnorm(h, delta, self.final_ln.weight, self.final_ln.eps)
        return self.lm_head(x)

    @staticmethod
    def _residual_many(mod, x: Tensor, h: Tensor) -> Optional[Tensor]:
        """Prefill row counts: ``h + mod(x)`` with the add in the int4g32 256 x 256-tile GEMM's epilogue (bit-equal to the two ops);
        None when module / shape are not served that way."""
        from .int4 import hip_ops as H4
        from .int8 import hip_ops as H8
        from .int4.qlinear import DynamicQuantizeLinear as Q4
        from .int8.qlinear import DynamicQuantizeLinear as Q8
        if not (x.dtype in (torch.float16, torch.bfloat16) and isinstance(mod, (Q4, Q8)) and not mod.act_quant
                and mod.weight_scale.dtype == x.dtype and h.dtype == x.dtype and h.shape[-1] == mod.out_features):
            return None
        if isinstance(mod, Q4):
            if not mod._packed_supported():
                return None
            out = H4.w4_forward_tiled_residual(x, mod.tiled(), mod.out_features, mod.bias, h)
        else:
            tiled = mod.prepare()._tiled
            if tiled is None:
                return None
            out = H8.w8_forward_tiled_residual(x, tiled, mod.out_features, mod.weight_scale, mod.bias, h)
        return None if out is None else out.view(h.shape)

    @staticmethod
    def _rows_fused(mod, h: Tensor, delta: Optional[Tensor], ln, gate_hidden: Optional[int] = None, want_hout: bool = True):
        """Batched decode at 2..4 rows: ``mod(rmsnorm(h + delta))`` (optionally SiLU * gate on a first MLP projection) in ONE
        launch of the 4x4x4-MFMA kernel; (out, hnew) or None when module / shape are not served that way."""
        from . import _lib
        from .int4 import hip_ops as H4
        from .int4.qlinear import DynamicQuantizeLinear as Q4
        if not (isinstance(mod, Q4) and mod._packed_supported() and not mod.act_quant and mod.weight_scale.dtype == h.dtype
                and not _lib.strict_for(h.dtype)):         # the 4x4x4-MFMA kernel computes in the exact-dequant arithmetic only
            return None
        if gate_hidden:
            if mod.out_features != 2 * gate_hidden or gate_hidden % 2:
                return None
            packed, bias = mod.gated_packed(gate_hidden)
            kind = _lib.PRO_ADDNORM | _lib.EPI_SILU_GATE
        else:
            packed, bias, kind = mod.prepare()._packed, mod.bias, _lib.PRO_ADDNORM
        return H4.w4_forward_rows_fused(kind, h, packed, mod.out_features, bias, delta, ln.weight, ln.eps, want_hout)

    @staticmethod
    def _gated_w_in(ffn, x: Tensor) -> Optional[Tensor]:
        """Batched decode (2..32 rows): w_in on its gate-interleaved int4g32 layout with SiLU * gate in the epilogue of the
        4x4x4-MFMA kernel (few rows, part 1 of that copy) or of the few-row kernel (part 2) - one launch instead of two; None
        when layer / shape are not served that way."""
        from .int4 import hip_ops as H4
        from .int4.qlinear import DynamicQuantizeLinear as Q4
        from .int8.qlinear import DynamicQuantizeLinear as Q8
        from . import _lib
        w_in = ffn.w_in
        if (isinstance(w_in, Q8) and x.dtype in (torch.float16, torch.bfloat16) and not w_in.act_quant and w_in.weight_scale.dtype == x.dtype
                and w_in.out_features == 2 * ffn.hidden_dim and ffn.hidden_dim % 2 == 0 and w_in.in_features % 16 == 0
                and x.numel() // x.shape[-1] >= 1024):     # int8 weight-only at prefill row counts: the same epilogue on the 256-tile GEMM
            from .int8 import hip_ops as H8
            if not _lib.get_lib().qlinear_gated_serves(x.numel() // x.shape[-1], w_in.out_features, w_in.in_features,
                                                       _lib.dtype_code(x.dtype), 8):
                return None                                # asked BEFORE the gate-interleaved copy is built
            tiled, s_perm, b_perm = w_in.gated_tiled(ffn.hidden_dim)
            return H8.w8_forward_tiled_gated(x, tiled, w_in.out_features, s_perm, b_perm)
        if not (isinstance(w_in, Q4) and x.dtype in (torch.float16, torch.bfloat16) and w_in._packed_supported()
                and not w_in.act_quant and w_in.weight_scale.dtype == x.dtype
                and w_in.out_features == 2 * ffn.hidden_dim and ffn.hidden_dim % 2 == 0):
            return None
        rows = x.numel() // x.shape[-1]
        strict = _lib.strict_for(x.dtype)
        if not H4.rows_on_tiled(rows, w_in.out_features, w_in.in_features, x.dtype, strict):
            if strict:                                     # rows the strict policy keeps on the VALU GEMV: no gated variant of it
                return None
            packed, bias = w_in.gated_packed(ffn.hidden_dim)
            return H4.w4_forward_gated(x, packed, w_in.out_features, bias, part1=True)
        if not _lib.get_lib().qlinear_gated_serves(rows, w_in.out_features, w_in.in_features, _lib.dtype_code(x.dtype), 4):
            return None                                    # asked BEFORE the gate-interleaved copies are built
        tiled, bias = w_in.gated_tiled(ffn.hidden_dim)
        return H4.w4_forward_gated(x, tiled, w_in.out_features, bias)

    def _one_row_kind(self, dtype) -> Optional[str]:
        """"int4" / "int8" when every QLinear of the graph can take the fused one-row launches, else None."""
        from .int4.qlinear import DynamicQuantizeLinear as Q4
        from .int8.qlinear import DynamicQuantizeLinear as Q8
        from . import _lib
        cached = self.__dict__.get("_kind_cache")          # 113 isinstance / layout checks per token otherwise; every change
        if cached is not None and cached[0] == (_lib.layout_epoch(), dtype):   # they depend on bumps the layout epoch
            return cached[1]
        mods = [self.lm_head] + [m for l in self.layers for m in (l.attn.qkv_proj, l.attn.o_proj, l.ffn.w_in, l.ffn.w_out)]
        kind = None
        if all(isinstance(m, Q4) and m._packed_supported() and not m.act_quant and m.weight_scale.dtype == dtype for m in mods):
            kind = "int4"
        elif dtype == torch.float16 and self.config.hidden_size <= 16384 and all(
                isinstance(m, Q8) and not m.act_quant and m.in_features % 16 == 0 and m.weight.is_contiguous() for m in mods):
            kind = "int8"
        self.__dict__["_kind_cache"] = ((_lib.layout_epoch(), dtype), kind)
        return kind

    def _step_one_row(self, h: Tensor, cache: KVCache, write_index: Tensor, position_ids: Tensor, mask: Tensor,
                      kind: str) -> Tensor:
        """Decode step of ONE row: 5 launches per layer.  The residual add + RMSNorm in front of qkv_proj / w_in /
        lm_head run inside those QLinear kernels' activation staging (qlinear_w4g32_fwd_packed_fused /
        qlinear_w8_fwd_fused); rotary + cache write + attention are one launch; SiLU * gate is w_in's epilogue."""
        from . import _lib, fused_ops as F_
        from .int4 import hip_ops as H4
        from .int8 import hip_ops as H8
        c = self.config
        H, G, D = c.num_attention_heads, c.num_multi_query_groups, c.head_hidden_size

        # Every launch of the step goes through a pre-bound plan (``_lib.make_plan``) once the checked wrapper has served the
        # call site: mod._fast[site] for the QLinear launches, cache.att_plans[layer] for the attention launch.  A plan
        # re-validates buffers / operands itself and returns None to fall back to the checked wrapper (which rebuilds it).
        def norm_linear(mod, x, ln, gate_hidden=None):
            """mod(rmsnorm(x)) (the residual adds run in the epilogues of o_proj / w_out); optional SiLU * gate epilogue."""
            site = ("norm", gate_hidden)
            plan = mod._fast.get(site)
            if plan is not None:
                out = plan(x, None, None)
                if out is not None:
                    return out
            flags = _lib.PRO_ADDNORM | (_lib.EPI_SILU_GATE if gate_hidden else 0)
            plan_out = []
            if kind == "int4":
                packed, bias = mod.gated_packed(gate_hidden) if gate_hidden else (mod.prepare()._packed, mod.bias)
                out = H4.w4_forward_fused(flags, x, packed, mod.out_features, bias, None, ln.weight, None, ln.eps,
                                          plan_out=plan_out, guards=(mod.weight, mod.weight_scale, mod.bias))
            else:
                w, sc, bias = mod.gated(gate_hidden) if gate_hidden else (mod.weight, mod.weight_scale, mod.bias)
                out = H8.w8_forward_fused(flags, x, w, sc, bias, None, ln.weight, None, ln.eps, plan_out=plan_out,
                                          guards=(mod.weight, mod.weight_scale, mod.bias))
            if plan_out[0] is not None:
                mod._fast[site] = plan_out[0]
            return out

        # The residual adds run in the EPILOGUES of o_proj / w_out (round(y + h)), so the RMSNorm prologues of qkv_proj /
        # w_in / lm_head stage two operands instead of three in every workgroup.  SiLU * gate runs in w_in's EPILOGUE on a
        # gate-interleaved copy of its weights (each wave owns (h, h, gate, gate) column quads), so the (1, 2 * hidden)
        # intermediate is never written.  (As a PROLOGUE of w_out it measured +6 us: every one of w_out's ~1000 blocks
        # redid the 13696 exponentials.)  The attention launch is a chain of round trips on B * G workgroups: its spare
        # workgroups read o_proj's weights into the caches meanwhile.
        def residual_linear(mod, x, resid):
            plan = mod._fast.get("resid")
            if plan is not None:
                out = plan(x, resid)
                if out is not None:
                    return out
            plan_out = []
            if kind == "int4":
                out = H4.w4_forward_residual(x, mod.prepare()._packed, mod.out_features, mod.bias, resid, plan_out=plan_out,
                                             guards=(mod.weight, mod.weight_scale))
            else:
                out = H8.w8_forward_residual(x, mod.weight, mod.weight_scale, mod.bias, resid, plan_out=plan_out)
            if plan_out[0] is not None:
                mod._fast["resid"] = plan_out[0]
            return out

        epoch = _lib.layout_epoch()
        att_key = (position_ids.data_ptr(), write_index.data_ptr(), mask.data_ptr(), cache.capacity)
        for i, layer in enumerate(self._layer_sites()):
            at, ff, attn_ln, ffn_ln = layer
            qkv = norm_linear(at.qkv_proj, h, attn_ln)
            att = None
            ap = cache.att_plans.get(i)
            if ap is not None and ap[0] == epoch and ap[1] == att_key:
                att = ap[2](qkv)
            if att is None:
                if kind == "int4":
                    nxt = (at.o_proj.prepare()._packed, _lib.NEXT_W4G32_PACKED, at.o_proj.out_features, at.o_proj.in_features)
                else:
                    nxt = (at.o_proj.weight, _lib.NEXT_W8_ROWS, at.o_proj.out_features, at.o_proj.in_features)
                plan_out = []
                att = F_.decode_attention_rope(qkv, self.freqs_cis_cache, position_ids, write_index, cache.k[i], cache.v[i],
                                               mask, H, G, D, prefetch=nxt, plan_out=plan_out)
                if plan_out and plan_out[0] is not None:
                    cache.att_plans[i] = (_lib.layout_epoch(), att_key, plan_out[0])
                epoch = _lib.layout_epoch()                 # prepare() above may have built a layout
            h = residual_linear(at.o_proj, att, h)
            hn = MLP_HOOK(self, cache, ff, ffn_ln, h) if (MLP_HOOK is not None and kind == "int4") else None
            if hn is None:
                y = norm_linear(ff.w_in, h, ffn_ln, gate_hidden=ff.hidden_dim)
                hn = residual_linear(ff.w_out, y, h)
            h = hn
        return norm_linear(self.lm_head, h, self.final_ln)

    def _layer_sites(self):
        """(attn, ffn, attn_ln, ffn_ln) per layer as a plain list: the decode step walks it 28 times per token, and every
        ``layer.attn`` through nn.Module.__getattr__ costs as much as a pre-bound launch.  Dropped by ``_apply``."""
        sites = self.__dict__.get("_sites")
        if sites is None:
            sites = [(l.attn, l.ffn, l.attn_ln, l.ffn_ln) for l in self.layers]
            self.__dict__["_sites"] = sites
        return sites

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_sites", None)
        self.__dict__.pop("_kind_cache", None)
        return super()._apply(fn, *args, **kwargs)

    # -- reference-shaped call ----------------------------------------------------------------------
    def forward(self, input_ids: Optional[Tensor] = None, input_embeddings: Optional[Tensor] = None,
                attention_mask: Optional[Tensor] = None, position_ids: Optional[Tensor] = None,
                labels: Optional[Tensor] = None, past_key_values=None):
        """Same contract as the reference forward (chatglm_q/model.py:329-392): returns
        (loss, logits, current_key_values) with key/values shaped (B, T, groups, 1, d_head)."""
        if input_embeddings is None:
            if input_ids is None:
                raise AssertionError("No input")
            h = self.word_embedding(input_ids)
        else:
            if input_ids is not None:
                raise AssertionError("Specify either 'input_ids' or 'input_embeddings'")
            h = input_embeddings
        B, S, _ = h.shape
        device = h.device
        past = 0 if past_key_values is None else past_key_values[0][0].shape[1]
        T = past + S
        if attention_mask is None:
            attention_mask = torch.ones(B, T, dtype=torch.long, device=device)
        if position_ids is None:
            position_ids = torch.cumsum(attention_mask, dim=1)           # positions start at 1 (model.py:308)
        t = torch.arange(T, device=device)
        blocked = (t[:, None] < t[None, :])[None] | ~attention_mask[:, None, :].bool()
        mask = (blocked.float() * -1e10)[:, -S:]                          # (B, S, T), model.py:311-317
        cs = self._rotary(position_ids[:, -S:])

        current = []
        for i, layer in enumerate(self.layers):
            q, k, v = layer.attn.project(layer.attn_ln(h), cs)
            if past_key_values is not None:
                k = torch.cat((past_key_values[i][0].squeeze(3), k), dim=1)
                v = torch.cat((past_key_values[i][1].squeeze(3), v), dim=1)
            current.append((k.detach().unsqueeze(3), v.detach().unsqueeze(3)))
            h = h + layer.attn.attend(h.dtype, q, k, v, mask)
            h = h + layer.ffn(layer.ffn_ln(h))
        logits = self.lm_head(self.final_ln(h))

        loss = None
        if labels is not None:
            n_classes = self.config.vocab_size
            loss = F.cross_entropy(logits[..., :-1, :].contiguous().float().view(-1, n_classes),
                                   labels[..., 1:].contiguous().view(-1))
        return loss, logits, tuple(current)


# ---- construction with quantized layers (chatglm_q/loader.py:41-66) ------------------------------
def _build_with(linear_cls, embedding_cls, config: ChatGLM2Config, dtype):
    import sys
    this = sys.modules[__name__]
    prev = this.Linear, this.Embedding
    this.Linear, this.Embedding = linear_cls, embedding_cls
    try:
        return ChatGLM2Model(config, dtype)
    finally:
        this.Linear, this.Embedding = prev


def create_quant_int4_model(config: ChatGLM2Config = ChatGLM2Config(), group_size: int = 32, dtype=None) -> ChatGLM2Model:
    from .int4.qlinear import DynamicQuantizeLinear, QEmbedding
    if group_size != 32:
        raise AssertionError("the model graph constructs int4 layers with the default group of 32")
    return _build_with(DynamicQuantizeLinear, QEmbedding, config, dtype)


def create_quant_int8_model(config: ChatGLM2Config = ChatGLM2Config(), dtype=None) -> ChatGLM2Model:
    from .int8.qlinear import DynamicQuantizeLinear, QEmbedding
    return _build_with(DynamicQuantizeLinear, QEmbedding, config, dtype)


@torch.no_grad()
def fill_synthetic_(model: ChatGLM2Model, seed: int = 0) -> ChatGLM2Model:
    """Random weights of the right format and scale (no checkpoint is available offline): uniform nibbles /
    int8 codes, scales such that activations keep O(1) magnitude through 28 layers, unit norms."""
    dev = model.final_ln.weight.device
    g = torch.Generator(device=dev).manual_seed(seed)
    for name, buf in model.state_dict().items():
        if buf.dtype == torch.uint8:
            buf.copy_(torch.randint(0, 256, buf.shape, dtype=torch.uint8, device=dev, generator=g))
        elif buf.dtype == torch.int8:
            buf.copy_(torch.randint(-127, 128, buf.shape, dtype=torch.int8, device=dev, generator=g))
        elif name.endswith("weight_scale"):
            fan_in = buf.shape[0] * 32 if buf.dim() == 2 else model.config.hidden_size
            if name.startswith("word_embedding"):
                amp = 0.25                                             # embeddings ~ N(0, 1)
            else:
                amp = 1.0 / (4.6 * math.sqrt(fan_in)) if buf.dim() == 2 else 1.0 / (73.0 * math.sqrt(fan_in))
            buf.copy_((torch.rand(buf.shape, device=dev, generator=g) * 0.5 + 0.75).to(buf.dtype) * amp)
        elif name.endswith("bias"):
            buf.copy_((torch.randn(buf.shape, device=dev, generator=g) * 0.02).to(buf.dtype))
        elif name.endswith("ln.weight"):
            buf.fill_(1.0)
        else:
            buf.copy_((torch.randn(buf.shape, device=dev, generator=g) * 0.02).to(buf.dtype))
    return model
