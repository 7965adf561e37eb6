"""The C restatement of the oracle (oracle/qlinear_oracle.c) against the numpy oracle and the fixtures."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import _golden as G
from oracle import qlinear_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = {"f32": 0, "f16": 1, "bf16": 2}


@pytest.fixture(scope="module")
def clib():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    lib.oracle_w4_fwd.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64] * 4 + [ctypes.c_int]
    lib.oracle_w8_fwd.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64] * 3 + [ctypes.c_int]
    lib.oracle_act_quant_rowwise.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 2 + [ctypes.c_int]
    lib.oracle_w8a8_fwd.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int64] * 3 + [ctypes.c_int]
    return lib


def enc(x, dt):
    """numpy array -> storage the C side expects (bf16 as uint16 bit patterns)."""
    if dt == "f32":
        return np.ascontiguousarray(x, dtype=np.float32)
    if dt == "f16":
        return np.ascontiguousarray(x, dtype=np.float16)
    return O.f32_to_bf16_bits(np.asarray(x, dtype=np.float32))


def dec(buf, dt):
    return O.bf16_bits_to_f32(buf) if dt == "bf16" else buf


def P(x):
    return None if x is None else x.ctypes.data_as(ctypes.c_void_p)


INT4 = G.load("int4_matmul.npz")
INT8 = G.load("int8_matmul.npz")


@pytest.mark.parametrize("entry", [":".join(c) for c in G.cases(INT4)])
def test_c_w4_equals_numpy_oracle(clib, entry):
    name, dt, has_bias = entry.split(":")
    c = G.case(INT4, name, dt)
    a = enc(c["a"].reshape(-1, c["a"].shape[-1]), dt)
    M, K = a.shape
    N = c["qweight"].shape[1]
    sc = enc(c["scale"], dt)
    bias = enc(c["bias"], dt) if has_bias == "1" else None
    out = np.zeros((M, N), dtype=a.dtype)
    assert clib.oracle_w4_fwd(P(a), P(c["qweight"]), P(sc), P(bias), P(out), M, N, K, 32, CODE[dt]) == 0
    ref = O.w4_matmul(c["a"], c["qweight"], c["scale"], c.get("bias") if has_bias == "1" else None, dtype=dt)
    got = dec(out, dt).reshape(ref.shape)
    # same algorithm, same double accumulation order per output: identical up to fp64 summation order
    assert O.rel_l2(got, ref) < 1e-6 if dt == "f32" else np.mean(np.asarray(got, np.float32) != np.asarray(ref, np.float32)) < 0.01
    assert O.rel_l2(got, c["out_fallback"]) <= {"f32": 1e-5, "f16": 1e-3, "bf16": 4e-3}[dt]


@pytest.mark.parametrize("entry", [":".join(c) for c in G.cases(INT8) if c[3] == "nk"])
def test_c_w8_equals_numpy_oracle(clib, entry):
    name, dt, has_bias, _ = entry.split(":")
    c = G.case(INT8, name, dt)
    a = enc(c["a"].reshape(-1, c["a"].shape[-1]), dt)
    M, K = a.shape
    N = c["weight_nk"].shape[0]
    sc = enc(c["scale"], dt)
    bias = enc(c["bias"], dt) if has_bias == "1" else None
    out = np.zeros((M, N), dtype=a.dtype)
    w = np.ascontiguousarray(c["weight_nk"])
    assert clib.oracle_w8_fwd(P(a), P(w), P(sc), P(bias), P(out), M, N, K, CODE[dt]) == 0
    got = dec(out, dt).reshape(c["out_fallback"].shape)
    assert O.rel_l2(got, c["out_fallback"]) <= {"f32": 1e-5, "f16": 1e-3, "bf16": 4e-3}[dt]


def test_c_w8a8_integer_stage_exact(clib):
    z = G.load("w8a8.npz")
    a = np.ascontiguousarray(z["a"])
    M, K = a.shape
    aq = np.zeros((M, K), np.int8)
    a_s = np.zeros(M, np.float32)
    assert clib.oracle_act_quant_rowwise(P(a), P(aq), P(a_s), M, K, 0) == 0
    assert np.array_equal(aq, z["a_q"]) and np.array_equal(a_s, z["a_scale"])
    w = np.ascontiguousarray(z["weight_nk"])
    N = w.shape[0]
    out = np.zeros((M, N), np.float32)
    ws = np.ascontiguousarray(z["w_scale"])
    assert clib.oracle_w8a8_fwd(P(aq), P(a_s), P(w), P(ws), None, P(out), M, N, K, 0) == 0
    assert np.allclose(out, z["out_w8a8"], rtol=1e-6, atol=1e-6)


def test_c_half_and_bf16_rounding(clib):
    # exercised through w8 with K = 1, weight 1, scale 1: out = round(a)
    vals = np.array([[65504.0], [65520.0], [1e-8], [6.1e-5], [5.96e-8], [2.98e-8], [2.9802322e-8], [1.0009765625]], dtype=np.float32)
    for dt in ("f16", "bf16"):
        a = enc(vals, dt)
        out = np.zeros((len(vals), 1), dtype=a.dtype)
        w = np.ones((1, 1), np.int8)
        sc = enc(np.ones(1), dt)
        assert clib.oracle_w8_fwd(P(a), P(w), P(sc), None, P(out), len(vals), 1, 1, CODE[dt]) == 0
        assert np.array_equal(np.asarray(dec(out, dt), np.float32), np.asarray(dec(a, dt), np.float32))
