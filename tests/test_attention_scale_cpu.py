"""The decode attention kernel (csrc/decode_ops.hip, decode_attention_group_kernel) scales the rotated query by 1 / sqrt(d) without a
division instruction sequence: q0 = x r, q = fma(fma(-q0, d, x), r, q0) with r = RN(1 / d) made on the host (Markstein's correction).
Every input it can see is a value of the activation dtype, so the claim "the correctly rounded quotient" is checked here exhaustively:
all 2^16 fp16 and bf16 bit patterns at d = sqrt(128) (the only head size the kernel serves), compared with the division the per-head
kernels and the reference formula use (chatglm_q/model.py:160, q / sqrt(d)) after the rounding to the activation dtype that follows."""
import numpy as np
import pytest
import torch


def _fma(a, b, c):
    # float32 fma through float64: a * b is exact in 53 bits (24 x 24); the sum's operands here are within a few binades of each other
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_two_step_division_by_sqrt_d_is_the_division(dt):
    d = np.float32(np.sqrt(np.float32(128.0)))
    r = np.float32(1.0 / np.float64(d))
    x = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(dt).float()
    x = x[torch.isfinite(x) & (x != 0)].numpy()                     # (+-0: the quotient's sign of zero may differ, its value does not)
    q0 = x * r
    q = _fma(_fma(-q0, np.full_like(x, d), x), np.full_like(x, r), q0)
    want = torch.from_numpy(x / d).to(dt)
    got = torch.from_numpy(q).to(dt)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    # ... and the plain multiplication by the reciprocal is NOT enough for fp16 (why the correction steps are there)
    if dt == torch.float16:
        assert not torch.equal(torch.from_numpy(x * r).to(dt).view(torch.int16), want.view(torch.int16))
