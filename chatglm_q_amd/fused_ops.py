"""Host wrappers for the fused decode-step neighbours of the QLinear calls (csrc/decode_ops.hip,
SURVEY.md 8f row N1).  Each replaces a handful of tiny torch launches with one HIP launch and keeps the
model graph's rounding sequence (chatglm_q/model.py, lines cited in include/qlinear_hip.h)."""
from __future__ import annotations

import os

import torch
from torch import Tensor

from . import _lib


def rmsnorm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    lib = _lib.get_lib()
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    out = torch.empty((x2.shape[0], x2.shape[1]), device=x.device, dtype=x.dtype)
    with torch.cuda.device(x.device):
        st = lib.qlinear_rmsnorm(x2.data_ptr(), weight.data_ptr(), out.data_ptr(), x2.shape[0], x2.shape[1],
                                 x2.stride(0) if x2.shape[0] > 1 else x2.shape[1], x2.shape[1], float(eps),
                                 _lib.dtype_code(x.dtype), _lib.stream_ptr(x.device))
    _lib.check(st, "qlinear_rmsnorm")
    return out.reshape(x.shape)


def add_rmsnorm(x: Tensor, delta: Tensor, weight: Tensor, eps: float):
    """h = x + delta (rounded), out = rmsnorm(h) * weight in ONE launch.  Returns (h, out)."""
    lib = _lib.get_lib()
    shape = x.shape
    x2 = x.reshape(-1, shape[-1]).contiguous()
    d2 = delta.reshape(-1, shape[-1]).contiguous()
    h = torch.empty_like(x2)
    out = torch.empty_like(x2)
    with torch.cuda.device(x.device):
        st = lib.qlinear_add_rmsnorm(x2.data_ptr(), d2.data_ptr(), weight.data_ptr(), h.data_ptr(), out.data_ptr(),
                                     x2.shape[0], x2.shape[1], x2.shape[1], float(eps), _lib.dtype_code(x.dtype),
                                     _lib.stream_ptr(x.device))
    _lib.check(st, "qlinear_add_rmsnorm")
    return h.reshape(shape), out.reshape(shape)


def rmsnorm_quant(x: Tensor, weight: Tensor, eps: float, delta: Tensor | None = None, want_out: bool = False):
    """RMSNorm (of ``x + delta`` when ``delta`` is given) emitted as int8 rows + one fp32 scale per row in the SAME launch
    (``qlinear_rmsnorm_quant_i8``): bit for bit ``act_quant_rowwise(rmsnorm(...))``.  Returns ``(h, a_q, a_scale, out)``:
    ``h`` = the rounded ``x + delta`` (``x`` itself without delta), ``out`` = the 16-bit row only with ``want_out``."""
    lib = _lib.get_lib()
    shape = x.shape
    x2 = x.reshape(-1, shape[-1]).contiguous()
    rows, dim = x2.shape
    d2 = delta.reshape(-1, dim).contiguous() if delta is not None else None
    h = torch.empty_like(x2) if d2 is not None else x2
    out = torch.empty_like(x2) if want_out else None
    a_q = torch.empty((rows, dim), device=x.device, dtype=torch.int8)
    a_s = torch.empty((rows,), device=x.device, dtype=torch.float32)
    if rows:
        with torch.cuda.device(x.device):
            st = lib.qlinear_rmsnorm_quant_i8(x2.data_ptr(), _lib.ptr(d2), weight.data_ptr(), h.data_ptr() if d2 is not None else None,
                                              _lib.ptr(out), a_q.data_ptr(), a_s.data_ptr(), rows, dim, dim, float(eps),
                                              _lib.dtype_code(x.dtype), _lib.stream_ptr(x.device))
        _lib.check(st, "qlinear_rmsnorm_quant_i8")
    return h.reshape(shape), a_q, a_s, (out.reshape(shape) if out is not None else None)


def silu_mul_quant(x: Tensor, hidden: int, want_out: bool = False):
    """``silu(x[..., :hidden]) * x[..., hidden:]`` as int8 rows + scales in one launch (``qlinear_silu_mul_quant_i8``): bit for
    bit ``act_quant_rowwise(silu_mul(x, hidden))``.  Returns ``(a_q, a_scale, out_or_None)``."""
    lib = _lib.get_lib()
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    rows = x2.shape[0]
    out = torch.empty((rows, hidden), device=x.device, dtype=x.dtype) if want_out else None
    a_q = torch.empty((rows, hidden), device=x.device, dtype=torch.int8)
    a_s = torch.empty((rows,), device=x.device, dtype=torch.float32)
    if rows:
        with torch.cuda.device(x.device):
            st = lib.qlinear_silu_mul_quant_i8(x2.data_ptr(), _lib.ptr(out), a_q.data_ptr(), a_s.data_ptr(), rows, hidden,
                                               x2.stride(0) if rows > 1 else 2 * hidden, hidden, _lib.dtype_code(x.dtype),
                                               _lib.stream_ptr(x.device))
        _lib.check(st, "qlinear_silu_mul_quant_i8")
    return a_q, a_s, (out.reshape(*x.shape[:-1], hidden) if out is not None else None)


def _check_rope_table(table: Tensor, k_cache: Tensor, d_head: int):
    """The kernels clamp positions to [0, capacity] (1-based positions: row r holds position <= r + 1) and skip cache
    rows outside the cache - the ABI carries no table length - so the rotary table must hold capacity + 1 positions."""
    if table.numel() < (k_cache.shape[1] + 1) * d_head:
        raise ValueError(f"rotary table of {table.numel() // d_head} positions does not cover a cache of {k_cache.shape[1]} rows "
                         f"(positions 0..{k_cache.shape[1]})")


def rope_kv_write(qkv: Tensor, table: Tensor, pos: Tensor, write_index: Tensor, k_cache: Tensor, v_cache: Tensor,
                  n_head: int, n_groups: int, d_head: int) -> Tensor:
    """qkv (B, S, (H+2G) D) -> rotated q (B, S, H*D); rotated k and v are written into the caches in place."""
    lib = _lib.get_lib()
    B, S, W = qkv.shape
    _check_rope_table(table, k_cache, d_head)
    qkv = qkv.contiguous()
    q = torch.empty((B, S, n_head * d_head), device=qkv.device, dtype=qkv.dtype)
    with torch.cuda.device(qkv.device):
        st = lib.qlinear_rope_kv_write(qkv.data_ptr(), table.data_ptr(), pos.contiguous().data_ptr(),
                                       write_index.data_ptr(), q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                                       B, S, n_head, n_groups, d_head, k_cache.shape[1], W,
                                       _lib.dtype_code(qkv.dtype), _lib.stream_ptr(qkv.device))
    _lib.check(st, "qlinear_rope_kv_write")
    return q


def decode_attention(q: Tensor, k_cache: Tensor, v_cache: Tensor, mask: Tensor, n_head: int, n_groups: int,
                     d_head: int) -> Tensor:
    """q (B, 1, H*D), caches (B, capacity, G, D), mask (B, 1, capacity) additive fp32 -> (B, 1, H*D)."""
    lib = _lib.get_lib()
    B = q.shape[0]
    out = torch.empty_like(q)
    with torch.cuda.device(q.device):
        st = lib.qlinear_decode_attention(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), mask.data_ptr(),
                                          out.data_ptr(), B, n_head, n_groups, d_head, k_cache.shape[1],
                                          _lib.dtype_code(q.dtype), _lib.stream_ptr(q.device))
    _lib.check(st, "qlinear_decode_attention")
    return out


def group_attention() -> bool:
    """False when QLINEAR_DISPATCH (include/qlinear_hip.h) turns the grouped MFMA decode attention off - asked of the library, which
    parses the variable once (``qlinear_dispatch_reload`` parses again): the split / workspace choice made here and the kernel the
    library launches cannot disagree after an environment change."""
    return not (_lib.get_lib().qlinear_dispatch_flags() & 16)


PREFETCH_NEXT = os.environ.get("QLINEAR_ATTENTION_PREFETCH", "1") != "0"   # spare workgroups warm the next linear's weights
SPLIT_ATTENTION_FROM = int(os.environ.get("QLINEAR_SPLIT_ATTENTION_FROM", "448"))   # cache capacity from which windows are split


def decode_attention_rope(qkv: Tensor, table: Tensor, pos: Tensor, write_index: Tensor, k_cache: Tensor, v_cache: Tensor,
                          mask: Tensor, n_head: int, n_groups: int, d_head: int, split: bool | None = None,
                          prefetch: tuple | None = None, plan_out: list | None = None) -> Tensor:
    """``decode_attention(rope_kv_write(qkv, ...), ...)`` for one position per sequence in a single launch:
    qkv (B, 1, (H+2G) D) -> (B, 1, H*D); the rotated key and the value are written into the caches at
    ``write_index[0]``.  ``prefetch = (weights, kind, N, K)``: the weights of the next one-row linear on the stream
    (kind ``_lib.NEXT_W4G32_PACKED`` / ``_lib.NEXT_W8_ROWS``) are pulled into the caches by spare workgroups of this
    launch (include/qlinear_hip.h).  ``plan_out``: a list receiving a pre-bound launch ``run(qkv)`` (``_lib.make_plan``) that
    is valid while table / pos / write_index / caches / mask / prefetched weights stay at their addresses - the caller's key.
    Cache rows behind ``write_index[0]`` hold nothing of the sequence (the reference appends the step's key / value at the end of its
    cache, chatglm_q/model.py:148-151): ``mask`` must hide them, and the 16-heads-per-group kernel skips them whatever it says."""
    lib = _lib.get_lib()
    B, S, W = qkv.shape
    if S != 1:
        raise ValueError("decode_attention_rope serves one position per sequence")
    _check_rope_table(table, k_cache, d_head)
    qkv = qkv.contiguous()
    out = torch.empty((B, 1, n_head * d_head), device=qkv.device, dtype=qkv.dtype)
    capacity = k_cache.shape[1]
    # the kernel reads mask row b at b * capacity: a mask sliced to the filled prefix and compacted would make rows b >= 1
    # read other sequences' columns
    if mask.dtype != torch.float32 or mask.numel() != B * capacity or mask.shape[-1] != capacity or not mask.is_contiguous():
        raise ValueError(f"decode_attention_rope: mask must be a contiguous fp32 (B, 1, capacity) = ({B}, 1, {capacity}) tensor, "
                         f"got {tuple(mask.shape)} {mask.dtype}")
    if split is None:
        # 16 heads per key/value group (16-bit, D = 128): the group kernel takes one 256-position window per block
        group_kernel = qkv.dtype != torch.float32 and d_head == 128 and n_head == 16 * n_groups and group_attention()
        split = capacity > 256 if group_kernel else capacity >= SPLIT_ATTENTION_FROM
    pos_c = pos.contiguous()
    with torch.cuda.device(qkv.device):
        ws_bytes = int(lib.qlinear_decode_attention_split_bytes(B, n_head, d_head, capacity)) if split else 0
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=qkv.device) if ws_bytes else None
        const = (table.data_ptr(), pos_c.data_ptr(), write_index.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), mask.data_ptr())
        shape = (B, n_head, n_groups, d_head, capacity, W, _lib.dtype_code(qkv.dtype))
        if prefetch is not None and PREFETCH_NEXT:
            weights, kind, n_next, k_next = prefetch
            name, tail = "qlinear_decode_attention_rope_prefetch", (weights.data_ptr(), kind, n_next, k_next)
        else:
            weights, name, tail = None, "qlinear_decode_attention_rope", ()
        st = getattr(lib, name)(qkv.data_ptr(), *const, out.data_ptr(), *shape, _lib.ptr(ws), ws_bytes, *tail, _lib.stream_ptr(qkv.device))
        _lib.check(st, name)
    if plan_out is not None and pos_c is pos:
        plan_out.append(_lib.make_plan(name, (None, *const, None, *shape, None, ws_bytes, *tail, None), 0, 7, 17 + len(tail), B, W,
                                       n_head * d_head, qkv.dtype, qkv.device, (), ws_slot=15 if ws_bytes else None, ws_bytes=ws_bytes,
                                       keep=(table, pos, write_index, k_cache, v_cache, mask, weights)))
    return out


def masked_softmax(scores: Tensor, mask: Tensor | None) -> Tensor:
    """``softmax(scores.float() + mask, -1).to(scores.dtype)`` in one launch.  scores (..., S, T) contiguous, mask
    (S, T) fp32 shared by all leading dims (or None)."""
    lib = _lib.get_lib()
    S, T = scores.shape[-2:]
    scores = scores.contiguous()
    out = torch.empty_like(scores)
    rows = scores.numel() // T
    if mask is not None:
        if mask.shape != (S, T) or mask.dtype != torch.float32:
            raise ValueError("mask must be an fp32 (S, T) tensor")
        if mask.stride(1) != 1:
            mask = mask.contiguous()
    with torch.cuda.device(scores.device):
        st = lib.qlinear_masked_softmax(scores.data_ptr(), _lib.ptr(mask), out.data_ptr(), rows, T, S, T,
                                        mask.stride(0) if mask is not None else T, T, _lib.dtype_code(scores.dtype),
                                        _lib.stream_ptr(scores.device))
    _lib.check(st, "qlinear_masked_softmax")
    return out


_PF_TILES = None


def prefill_attention_tiles():
    """(query positions per workgroup, keys per tile) of qlinear_prefill_attention: the granularity of its tile flags."""
    global _PF_TILES
    if _PF_TILES is None:
        import ctypes
        qb, kb = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(_lib.get_lib().qlinear_prefill_attention_tiles(ctypes.byref(qb), ctypes.byref(kb)), "qlinear_prefill_attention_tiles")
        _PF_TILES = (int(qb.value), int(kb.value))
    return _PF_TILES


def attention_tile_flags(mask: Tensor) -> Tensor:
    """Per (sequence, query block, key tile) of an additive fp32 mask (B, S, T): 0 = the tile can be skipped (every entry
    <= -1e9 and every row of the block has an entry >= -1e6: the probabilities are exactly 0 in fp32), 2 = every entry is 0
    (no mask loads), 1 = otherwise.  One pass over the mask per prefill chunk, shared by all layers."""
    QB, KB = prefill_attention_tiles()
    B, S, T = mask.shape
    nq, nk = (S + QB - 1) // QB, (T + KB - 1) // KB
    blocked, zero, row_ok = mask <= -1e9, mask == 0, (mask >= -1e6).any(dim=-1)
    if nq * QB != S or nk * KB != T:                       # rows / keys past the end never force a load or forbid a skip
        pad = (0, nk * KB - T, 0, nq * QB - S)
        blocked = torch.nn.functional.pad(blocked, pad, value=True)
        zero = torch.nn.functional.pad(zero, pad, value=True)
        row_ok = torch.nn.functional.pad(row_ok, (0, nq * QB - S), value=True)
    skip = blocked.view(B, nq, QB, nk, KB).all(dim=4).all(dim=2) & row_ok.view(B, nq, QB).all(dim=2)[:, :, None]
    clear = zero.view(B, nq, QB, nk, KB).all(dim=4).all(dim=2)
    # (torch.where, not masked assignment: no host synchronisation, capturable)
    return torch.where(skip, 0, torch.where(clear, 2, 1)).to(torch.uint8).contiguous()


def prefill_attention_supported(q_dtype, H: int, G: int, D: int) -> bool:
    return D == 128 and H == 16 * G and q_dtype in (torch.float16, torch.bfloat16)


def prefill_attention(q: Tensor, k_cache: Tensor, v_cache: Tensor, mask: Tensor | None, flags: Tensor | None, T: int,
                      H: int, G: int, D: int) -> Tensor:
    """softmax(round(q / sqrt(D)) k^T + mask) v for S new positions against cache rows [0, T) in one launch
    (chatglm_q/model.py:157-175).  q (B, S, H * D) rotated queries, caches (B, capacity, G, D), mask (B, S, T) additive fp32
    (or None), flags from attention_tile_flags(mask) (or None: every tile is processed with its mask).  Returns (B, S, H * D)."""
    lib = _lib.get_lib()
    B, S = q.shape[0], q.shape[1]
    if not (q.is_contiguous() and k_cache.is_contiguous() and v_cache.is_contiguous()):
        raise ValueError("prefill_attention: q and the caches must be contiguous")
    if k_cache.shape != v_cache.shape or k_cache.shape[0] != B or tuple(k_cache.shape[2:]) != (G, D) or q.numel() != B * S * H * D:
        raise ValueError("prefill_attention: shape mismatch")
    if k_cache.dtype != q.dtype or v_cache.dtype != q.dtype:
        raise ValueError("prefill_attention: dtype mismatch")
    ldm = 0
    if mask is not None:
        if mask.dtype != torch.float32 or tuple(mask.shape) != (B, S, T):
            raise ValueError("mask must be an fp32 (B, S, T) tensor")
        mask = mask.contiguous()
        ldm = T
    if flags is not None:
        QB, KB = prefill_attention_tiles()
        if mask is None or flags.dtype != torch.uint8 or tuple(flags.shape) != (B, (S + QB - 1) // QB, (T + KB - 1) // KB):
            raise ValueError("tile flags do not match the mask's shape")
        flags = flags.contiguous()
    out = torch.empty((B, S, H * D), device=q.device, dtype=q.dtype)
    with torch.cuda.device(q.device):
        st = lib.qlinear_prefill_attention(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), _lib.ptr(mask), _lib.ptr(flags),
                                           out.data_ptr(), B, S, T, H, G, D, k_cache.shape[1], ldm, _lib.dtype_code(q.dtype),
                                           _lib.stream_ptr(q.device))
    _lib.check(st, "qlinear_prefill_attention")
    return out


def silu_mul(x: Tensor, hidden: int) -> Tensor:
    lib = _lib.get_lib()
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    out = torch.empty((x2.shape[0], hidden), device=x.device, dtype=x.dtype)
    with torch.cuda.device(x.device):
        st = lib.qlinear_silu_mul(x2.data_ptr(), out.data_ptr(), x2.shape[0], hidden,
                                  x2.stride(0) if x2.shape[0] > 1 else 2 * hidden, hidden,
                                  _lib.dtype_code(x.dtype), _lib.stream_ptr(x.device))
    _lib.check(st, "qlinear_silu_mul")
    return out.reshape(*x.shape[:-1], hidden)


def greedy_advance(logits: Tensor, tok: Tensor, write_index: Tensor, pos: Tensor, mask: Tensor):
    """In place: tok = argmax(logits), pos += 1, write_index += 1, mask[..., new write_index] = 0 - one launch."""
    lib = _lib.get_lib()
    B, N = logits.shape
    with torch.cuda.device(logits.device):
        st = lib.qlinear_greedy_advance(logits.data_ptr(), B, N, logits.stride(0), tok.data_ptr(), write_index.data_ptr(),
                                        pos.data_ptr(), mask.data_ptr(), mask.shape[-1], _lib.dtype_code(logits.dtype),
                                        _lib.stream_ptr(logits.device))
    _lib.check(st, "qlinear_greedy_advance")


SAMPLER_MAX_TOP_K = 1024     # csrc/sampler.hip S_KMAX


def new_rng_state(batch: int, seed: int, device) -> Tensor:
    """Device state of qlinear_top_p_sample's generator: int64[1 + batch] = (seed, one draw counter per row)."""
    st = torch.zeros(1 + batch, dtype=torch.int64)
    st[0] = int(seed) & 0x7FFFFFFFFFFFFFFF
    return st.to(device)


def top_p_sample(logits: Tensor, tok: Tensor, rng_state: Tensor | None = None, top_k: int = 100, top_p: float = 0.8,
                 temperature: float = 1.0, dev_params: Tensor | None = None, write_index: Tensor | None = None,
                 pos: Tensor | None = None, mask: Tensor | None = None, return_distribution: bool = False):
    """chatglm_q/decoder.py:12-27 in one launch: ``tok[b] = top_p_sampling(logits[b], top_k, top_p, temperature)`` for every row of
    `logits` (B, N), drawing from the counter-based generator in `rng_state` (new_rng_state; None: seed 0, counter 0).  With
    write_index / pos / mask the launch also does a decode step's bookkeeping (greedy_advance).  dev_params: device float32[3] =
    (top_k, top_p, temperature) read by the kernel instead of the arguments (a captured graph then serves any setting).
    return_distribution: also return (probs (B, k), indices (B, k), u (B,)) - the filtered distribution in sorted order (ties: lowest
    index first), its token ids and the uniform number each row drew."""
    lib = _lib.get_lib()
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise ValueError("top_p_sample: logits must be (B, N) with unit column stride")
    B, N = logits.shape
    if tok.dtype != torch.int64 or tok.numel() != B or not tok.is_contiguous():
        raise ValueError("top_p_sample: tok must be a contiguous int64 tensor of B elements")
    if dev_params is None and not (1 <= top_k and temperature > 0):
        raise ValueError("top_p_sample: top_k >= 1 and temperature > 0")
    if rng_state is not None and (rng_state.dtype != torch.int64 or rng_state.numel() < 1 + B or not rng_state.is_contiguous()):
        raise ValueError("top_p_sample: rng_state must be int64[1 + B] (new_rng_state)")
    if dev_params is not None and (dev_params.dtype != torch.float32 or dev_params.numel() < 3 or not dev_params.is_contiguous()):
        raise ValueError("top_p_sample: dev_params must be float32[3]")
    k_out = min(SAMPLER_MAX_TOP_K if dev_params is not None else top_k, N)
    probs = idx = u = None
    if return_distribution:
        probs = torch.zeros((B, k_out), device=logits.device, dtype=torch.float32)
        idx = torch.zeros((B, k_out), device=logits.device, dtype=torch.int64)
        u = torch.zeros(B, device=logits.device, dtype=torch.float32)
    with torch.cuda.device(logits.device):
        st = lib.qlinear_top_p_sample(logits.data_ptr(), B, N, logits.stride(0), int(top_k), float(top_p), float(temperature),
                                      _lib.ptr(dev_params), _lib.ptr(rng_state), tok.data_ptr(), _lib.ptr(write_index), _lib.ptr(pos),
                                      _lib.ptr(mask), mask.shape[-1] if mask is not None else 0, _lib.ptr(probs), _lib.ptr(idx),
                                      _lib.ptr(u), k_out, _lib.dtype_code(logits.dtype), _lib.stream_ptr(logits.device))
    _lib.check(st, "qlinear_top_p_sample")
    if return_distribution:
        return probs, idx, u
