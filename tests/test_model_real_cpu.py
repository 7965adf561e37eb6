"""ChatGLM2-6B LAYER DIMENSIONS (hidden 4096, FFN 13696, 32 heads x 128, 2 groups; 2 layers, vocab 1024) against logits
and cache rows the REFERENCE model produced for the same seeded weights (tests/golden/real_model.npz, generator:
tests/golden/make_golden.py::gen_model_real).  CPU branch of the modules; the GPU twin is test_model_real_gpu.py."""
import numpy as np
import pytest
import torch

import _golden as G
from oracle import qlinear_oracle as O
from chatglm_q_amd import model as M
from chatglm_q_amd.decoder import DecodeSession

R = G.load("real_model.npz")


def build_real(device="cpu", dtype=torch.float16):
    assert [int(v) for v in R["config"]] == [G.REAL_DIM_CONFIG[k] for k in (
        "hidden_size", "inner_hidden_size", "head_hidden_size", "num_multi_query_groups", "num_attention_heads",
        "num_layers", "vocab_size", "max_sequence_length")]
    cfg = M.ChatGLM2Config(**G.REAL_DIM_CONFIG)
    model = M.create_quant_int4_model(cfg, dtype=dtype)
    G.fill_seeded_(model.state_dict(), int(R["seed"][0]))
    for m in model.modules():
        if hasattr(m, "invalidate"):
            m.invalidate()
    return model.to(device).eval(), cfg


def t2n(t):
    return t.detach().float().cpu().numpy()


def f32(name):
    return R[name].astype(np.float32)


# fp16 end to end, two layers of real width: the reference itself moves by ~1e-3 between matmul summation orders
# (one fp16 rounding of a 4096-term sum per layer output); measured values are printed by the GPU twin
TOL_CPU = 2e-3


@pytest.fixture(scope="module")
def real_cpu():
    """The modules' CPU branch with its dense product accumulated by the fp32 GEMM and rounded once (torch's fp16 CPU GEMM
    does the same arithmetic in another summation order at 0.1 GFLOP/s: this file took 14 minutes with it)."""
    from chatglm_q_amd.int4 import qlinear as Q4
    torch.set_num_threads(8)
    saved = Q4._dense_matmul
    Q4._dense_matmul = lambda A, W: (A.float() @ W.float()).to(A.dtype) if A.dtype == torch.float16 else saved(A, W)
    try:
        yield build_real()
    finally:
        Q4._dense_matmul = saved


def test_b1_prefill_and_cached_decode_match_reference(real_cpu):
    model, cfg = real_cpu
    ids = torch.from_numpy(R["b1/ids"])
    with torch.no_grad():
        _, logits, kv = model(input_ids=ids)
        assert O.rel_l2(t2n(logits), f32("b1/prefill_logits")) < TOL_CPU
        for t in range(3):
            _, lg, kv = model(input_ids=torch.from_numpy(R["b1/next_ids"][t:t + 1]).view(1, 1), past_key_values=kv)
            assert O.rel_l2(t2n(lg[:, -1]), f32(f"b1/decode_logits_{t}")) < TOL_CPU
    assert kv[1][0].shape == R["b1/kv1_k"].shape
    assert O.rel_l2(t2n(kv[1][0]), f32("b1/kv1_k")) < TOL_CPU
    assert O.rel_l2(t2n(kv[1][1]), f32("b1/kv1_v")) < TOL_CPU


def test_b4_left_padded_chunked_prefill_matches_reference(real_cpu):
    """chatglm_q/model.py:297-318: pads masked as columns, positions = cumsum(attention_mask); 3 chunks of 8."""
    model, cfg = real_cpu
    ids = torch.from_numpy(R["b4/ids"])
    attn = torch.from_numpy(R["b4/attention_mask"])
    S = ids.shape[1]
    with torch.no_grad():
        kv = None
        for c0 in (0, 8, 16):
            _, lg, kv = model(input_ids=ids[:, c0:c0 + 8], attention_mask=attn[:, :c0 + 8], past_key_values=kv)
        assert O.rel_l2(t2n(lg), f32("b4/last_chunk_logits")) < TOL_CPU
        # the same through the static-cache session (what the GPU path runs)
        sess = DecodeSession(model, 4, 32, use_graph=False)
        last = sess.prefill(ids, chunk=8, attention_mask=attn[:, :S])
        assert O.rel_l2(t2n(last), f32("b4/last_chunk_logits")[:, -1]) < TOL_CPU
        for t in range(2):
            lg = sess.decode_step(torch.from_numpy(R["b4/next_ids"][t]).view(4, 1), greedy=False)
            assert O.rel_l2(t2n(lg), f32(f"b4/decode_logits_{t}")) < TOL_CPU
    valid = torch.from_numpy(R["b4/attention_mask"]).bool()[:, : S + 2]
    k_ref, v_ref = f32("b4/kv1_k")[:, :, :, 0], f32("b4/kv1_v")[:, :, :, 0]
    k_got, v_got = t2n(sess.cache.k[1][:, : S + 2]), t2n(sess.cache.v[1][:, : S + 2])
    sel = valid.numpy()
    assert O.rel_l2(k_got[sel], k_ref[sel]) < TOL_CPU and O.rel_l2(v_got[sel], v_ref[sel]) < TOL_CPU
