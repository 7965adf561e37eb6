"""Round-to-nearest int8 per-row quantiser: the writer of the int8 buffer format
(chatglm_q/int8/quantizer.py:7-52).  GPTQ calibration is offline tooling, out of scope."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .qlinear import DynamicQuantizeLinear, QEmbedding

max_q_int8 = 2 ** (8 - 1) - 1  # 127


@torch.no_grad()
def quantize_int8(inputs: Tensor):
    """Row-wise symmetric: weights (out, in) or activations (channels, features).
    scale = clamp(max|row| / 127, min=1e-10); q = clamp(round(x / scale), -127, 127)."""
    scale = torch.clamp(inputs.abs().amax(dim=1, keepdim=True) / max_q_int8, min=1e-10)
    q = torch.clamp(torch.round(inputs / scale), -max_q_int8, max_q_int8)
    return q.to(torch.int8), scale.squeeze(dim=-1)


@torch.no_grad()
def get_quant_int8_linear(layer: nn.Linear):
    if not isinstance(layer, nn.Linear):
        raise AssertionError("expected nn.Linear")
    q_weight, scale = quantize_int8(layer.weight)
    out = DynamicQuantizeLinear(layer.in_features, layer.out_features, layer.bias is not None,
                                device=layer.weight.device, dtype=layer.weight.dtype)
    out.apply_weights_(q_weight, scale, layer.bias)
    return out


@torch.no_grad()
def get_quant_embedding(layer: nn.Embedding):
    if not isinstance(layer, nn.Embedding):
        raise AssertionError("expected nn.Embedding")
    q_weight, scale = quantize_int8(layer.weight.t())
    out = QEmbedding(layer.num_embeddings, layer.embedding_dim, device=layer.weight.device, dtype=layer.weight.dtype)
    out.apply_weights_(q_weight.t(), scale)
    return out
