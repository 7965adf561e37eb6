"""Round-to-nearest int4 group quantiser: the WRITER of the buffer format the kernels consume
(chatglm_q/int4/quantizer.py:8-75).  Used to synthesise weights for benchmarks and to convert
``nn.Linear`` / ``nn.Embedding`` layers; the GPTQ calibration of the reference is offline tooling and
out of scope."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from . import qlinear
from .qlinear import DynamicQuantizeLinear, QEmbedding

max_q_int4 = 2 ** (4 - 1) - 1  # 7


@torch.no_grad()
def quantize_int4(x: Tensor, GROUP_K: int = qlinear.DEFAULT_GROUP_SIZE):
    """x: (in_dim, out_dim) weight.  Returns (packed (in/2, out) uint8, scale (in/GROUP_K, out)).

    Per group: scale = clamp(max|x| / 7, min=1e-10); q = clamp(round(x / scale), -7, 7) (round half to
    even); stored nibble = q + 8; two K rows per byte, even row in the low nibble.
    """
    if x.dim() != 2:
        raise AssertionError("expected a 2-D weight")
    K, N = x.shape
    if K % GROUP_K != 0:
        raise AssertionError(f"K={K} not divisible by group {GROUP_K}")
    G = K // GROUP_K
    xg = x.reshape(G, GROUP_K, N)
    scale = torch.clamp(xg.abs().amax(dim=1, keepdim=True) / max_q_int4, min=1e-10)
    q = torch.clamp(torch.round(xg / scale), -max_q_int4, max_q_int4)
    q = (q + 8).to(torch.uint8).reshape(K, N)
    packed = (q[0::2] & 0xF) | ((q[1::2] & 0xF) << 4)
    return packed.contiguous(), scale.reshape(G, N)


@torch.no_grad()
def get_quant_int4_linear(layer: nn.Linear, group_size: int = qlinear.DEFAULT_GROUP_SIZE):
    if not isinstance(layer, nn.Linear):
        raise AssertionError("expected nn.Linear")
    q_weight, scale = quantize_int4(layer.weight.t(), group_size)
    out = DynamicQuantizeLinear(layer.in_features, layer.out_features, layer.bias is not None, group_size,
                                device=layer.weight.device, dtype=layer.weight.dtype)
    out.apply_weights_(q_weight, scale, layer.bias)
    return out


@torch.no_grad()
def get_quant_embedding(layer: nn.Embedding, group_size: int = qlinear.DEFAULT_GROUP_SIZE):
    if not isinstance(layer, nn.Embedding):
        raise AssertionError("expected nn.Embedding")
    q_weight, scale = quantize_int4(layer.weight, group_size)
    out = QEmbedding(layer.num_embeddings, layer.embedding_dim, group_size,
                     device=layer.weight.device, dtype=layer.weight.dtype)
    out.apply_weights_(q_weight, scale)
    return out
