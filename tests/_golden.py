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
