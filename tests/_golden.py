"""Loader for the fixtures in tests/golden/ (data generated from the reference by
tests/golden/make_golden.py).  bf16 arrays are stored as uint16 bit patterns."""
import os

import numpy as np

from oracle import qlinear_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Case(dict):
    """dict of arrays for one fixture case; bf16 entries decoded to float32."""
    dtype: str = "f32"


def load(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def case(npz, prefix: str, dtype: str | None = None) -> Case:
    c = Case()
    pre = prefix + "/"
    for k in npz.files:
        if not k.startswith(pre):
            continue
        key = k[len(pre):]
        if key.endswith("_bf16bits"):
            c[key[: -len("_bf16bits")]] = O.bf16_bits_to_f32(npz[k])
        else:
            c[key] = npz[k]
    c.dtype = dtype or "f32"
    return c


def cases(npz):
    """Yield (name, dtype, has_bias, extra...) tuples from the ``__cases__`` index."""
    for entry in npz["__cases__"]:
        yield str(entry).split(":")


def to_torch(arr, dtype: str):
    import torch
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    if arr.dtype in (np.uint8, np.int8, np.int32, np.int64):
        return torch.from_numpy(np.ascontiguousarray(arr))
    return torch.from_numpy(np.ascontiguousarray(arr).astype(np.float32)).to(tdt)


# ---- real-dimension ChatGLM2 fixture (tests/golden/real_model.npz) --------------------------------------------
# ChatGLM2-6B layer geometry (hidden 4096, FFN 13696, 32 heads x 128, 2 key/value groups) with 2 layers and a
# 1024-token vocabulary.  The int4g32 buffers are ~410 MB, so the fixture stores NO weights: both the generator
# (tests/golden/make_golden.py, which fills the REFERENCE model) and the tests (which fill the build's model) call
# fill_seeded_() - every buffer is a pure function of (seed, its state_dict key, shape, dtype) drawn from a CPU
# torch.Generator, independent of iteration order.
REAL_DIM_CONFIG = dict(hidden_size=4096, inner_hidden_size=13696, head_hidden_size=128, num_multi_query_groups=2,
                       num_attention_heads=32, num_layers=2, vocab_size=1024, max_sequence_length=2304)
REAL_DIM_SEED = 8100


def fill_seeded_(state_dict, seed: int = REAL_DIM_SEED):
    """In-place synthetic int4g32 weights: nibbles uniform over 1..15, i.e. q = n - 8 uniform over -7..7 and ZERO MEAN
    (uniform 0..15 gives every weight matrix a rank-one mean component that swamps the input-dependent part of the
    logits and would make a logit comparison insensitive), scales sized so activations stay O(1)."""
    import math
    import zlib

    import torch
    with torch.no_grad():
        for key in sorted(state_dict.keys()):
            buf = state_dict[key]
            gen = torch.Generator().manual_seed(seed + zlib.crc32(key.encode()))
            if buf.dtype == torch.uint8:
                lo = torch.randint(1, 16, buf.shape, dtype=torch.uint8, generator=gen)
                val = lo | (torch.randint(1, 16, buf.shape, dtype=torch.uint8, generator=gen) << 4)
            elif key.endswith("weight_scale"):
                if key.startswith("word_embedding"):
                    amp = 0.25                                    # embedding rows ~ unit variance
                else:
                    amp = 1.0 / (4.32 * math.sqrt(buf.shape[0] * 32))  # q uniform over -7..7 has std 4.32
                val = (torch.rand(buf.shape, generator=gen) * 0.5 + 0.75) * amp
            elif key.endswith("bias"):
                val = torch.randn(buf.shape, generator=gen) * 0.05
            elif "ln" in key:
                val = 1.0 + 0.1 * torch.randn(buf.shape, generator=gen)
            else:
                raise KeyError(f"unexpected state_dict entry {key}")
            buf.copy_(val.to(buf.dtype))
    return state_dict


def fill_seeded_int8_(state_dict, seed: int = REAL_DIM_SEED + 17):
    """In-place synthetic int8 per-channel weights for the int8 model fixtures (tests/golden/model_r3.npz): codes uniform
    over -127..127 (std 73.3), per-channel scales sized so activations stay O(1); a pure function of (seed, key, shape,
    dtype) like fill_seeded_."""
    import math
    import zlib

    import torch
    with torch.no_grad():
        for key in sorted(state_dict.keys()):
            buf = state_dict[key]
            gen = torch.Generator().manual_seed(seed + zlib.crc32(key.encode()))
            if buf.dtype == torch.int8:
                val = torch.randint(-127, 128, buf.shape, dtype=torch.int8, generator=gen)
            elif key.endswith("weight_scale"):
                if key.startswith("word_embedding"):
                    amp = 1.0 / 73.3                              # embedding rows ~ unit variance
                else:
                    fan_in = state_dict[key[: -len("_scale")]].shape[1]
                    amp = 1.0 / (73.3 * math.sqrt(fan_in))
                val = (torch.rand(buf.shape, generator=gen) * 0.5 + 0.75) * amp
            elif key.endswith("bias"):
                val = torch.randn(buf.shape, generator=gen) * 0.05
            elif "ln" in key:
                val = 1.0 + 0.1 * torch.randn(buf.shape, generator=gen)
            else:
                raise KeyError(f"unexpected state_dict entry {key}")
            buf.copy_(val.to(buf.dtype))
    return state_dict


TINY_INT8_CONFIG = dict(hidden_size=256, inner_hidden_size=384, head_hidden_size=32, num_multi_query_groups=2,
                        num_attention_heads=8, num_layers=2, vocab_size=320, max_sequence_length=64)
