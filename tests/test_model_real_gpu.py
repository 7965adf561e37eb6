"""Configs 4 and 5 at ChatGLM2-6B LAYER DIMENSIONS on the GPU, against the reference model's own outputs
(tests/golden/real_model.npz): the fused 5-launch one-row step, the HIP-graph step, the grouped MFMA attention,
the few-row kernels (batch 4) and the large-M GEMMs (chunked prefill 4 x 2048, chunks of 512 and of 2048 = 8192
rows per QLinear call), with left padding (chatglm_q/model.py:297-318)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import qlinear_oracle as O  # noqa: E402
from chatglm_q_amd import _lib  # noqa: E402
from chatglm_q_amd.decoder import DecodeSession  # noqa: E402
from test_model_real_cpu import R, build_real, f32, t2n  # noqa: E402

DEV = "cuda:0"
# fp16, two layers of real width.  Bar from VERDICT r1: rel-L2 <= 2e-3 on logits; the measured values are printed
# (pytest -s) and recorded in DESIGN.md section 5.
TOL = 2e-3


@pytest.fixture(scope="module")
def real_gpu():
    model, cfg = build_real(DEV)
    return model, cfg


def _err(name, got, want):
    e = O.rel_l2(got, want)
    print(f"[real-dim parity] {name}: rel-L2 {e:.3e}")
    return e


@pytest.mark.parametrize("use_graph", [False, True])
def test_b1_generate_pattern_fused_one_row_step(real_gpu, use_graph):
    """Config 4's call pattern: prefill, then one-row decode steps through the 5-launch fused layer
    (qkv GEMV + RMSNorm prologue, rotary + cache write + MFMA group attention, o_proj + residual, w_in + SiLU * gate,
    w_out + residual), eager and replayed from a HIP graph."""
    model, cfg = real_gpu
    ids = torch.from_numpy(R["b1/ids"])
    sess = DecodeSession(model, 1, 64, use_graph=use_graph)
    before = _lib.launch_count()
    last = sess.prefill(ids)
    assert _err("b1 prefill last", t2n(last), f32("b1/prefill_logits")[:, -1]) < TOL
    if use_graph:
        sess.tok.fill_(int(R["b1/next_ids"][0]))
        sess.capture(greedy=False)
        assert sess.graph is not None
    for t in range(3):
        lg = sess.decode_step(torch.from_numpy(R["b1/next_ids"][t:t + 1]).view(1, 1), greedy=False)
        assert _err(f"b1 decode {t} graph={use_graph}", t2n(lg), f32(f"b1/decode_logits_{t}")) < TOL
    assert _lib.launch_count() > before
    n = ids.shape[1] + 3
    assert _err("b1 kv1 k", t2n(sess.cache.k[1][:, :n]), f32("b1/kv1_k")[:, :, :, 0]) < TOL
    assert _err("b1 kv1 v", t2n(sess.cache.v[1][:, :n]), f32("b1/kv1_v")[:, :, :, 0]) < TOL


def test_b1_full_prefill_logits_every_position(real_gpu):
    model, cfg = real_gpu
    ids = torch.from_numpy(R["b1/ids"]).to(DEV)
    with torch.no_grad():
        _, logits, _ = model(input_ids=ids)
    assert _err("b1 prefill all positions", t2n(logits), f32("b1/prefill_logits")) < TOL


@pytest.mark.parametrize("use_graph", [False, True])
def test_b4_left_padded_chunked_prefill_then_batched_decode(real_gpu, use_graph):
    model, cfg = real_gpu
    ids = torch.from_numpy(R["b4/ids"])
    attn = torch.from_numpy(R["b4/attention_mask"])
    S = ids.shape[1]
    sess = DecodeSession(model, 4, 64, use_graph=use_graph)
    last = sess.prefill(ids, chunk=8, attention_mask=attn[:, :S])
    assert _err("b4 chunked prefill last", t2n(last), f32("b4/last_chunk_logits")[:, -1]) < TOL
    if use_graph:
        sess.tok.copy_(torch.from_numpy(R["b4/next_ids"][0]).view(4, 1))
        sess.capture(greedy=False)
    for t in range(2):
        lg = sess.decode_step(torch.from_numpy(R["b4/next_ids"][t]).view(4, 1), greedy=False)
        assert _err(f"b4 decode {t} graph={use_graph}", t2n(lg), f32(f"b4/decode_logits_{t}")) < TOL
    sel = attn.bool()[:, : S + 2].numpy()
    k_got, v_got = t2n(sess.cache.k[1][:, : S + 2]), t2n(sess.cache.v[1][:, : S + 2])
    assert _err("b4 kv1 k", k_got[sel], f32("b4/kv1_k")[:, :, :, 0][sel]) < TOL
    assert _err("b4 kv1 v", v_got[sel], f32("b4/kv1_v")[:, :, :, 0][sel]) < TOL


@pytest.mark.parametrize("chunk", [512, 2048])
def test_config5_chunked_prefill_2048x4(real_gpu, chunk):
    """BASELINE config 5's workload at two layers: batch 4 x 2048 positions, left padded.  chunk 512 = the fixture's
    own chunking (2048 rows per QLinear call); chunk 2048 = one pass, M = 8192 rows per QLinear call (the reference's
    chunked and unchunked results coincide: make_golden prints the difference)."""
    model, cfg = real_gpu
    ids = torch.from_numpy(R["big/ids"].astype(np.int64))
    lens = [int(v) for v in R["big/lens"]]
    S = ids.shape[1]
    attn = torch.zeros(4, S, dtype=torch.long)
    for b, n in enumerate(lens):
        attn[b, S - n:] = 1
    sess = DecodeSession(model, 4, S, use_graph=False)
    last = sess.prefill(ids, chunk=chunk, attention_mask=attn)
    torch.cuda.synchronize()
    assert _err(f"config5 chunk={chunk} last logits", t2n(last), f32("big/last_logits")) < TOL
    pos = torch.from_numpy(R["big/kv_positions"])
    sel = attn[:, pos].bool().numpy()
    k_got, v_got = t2n(sess.cache.k[1][:, pos.to(DEV)]), t2n(sess.cache.v[1][:, pos.to(DEV)])
    assert _err(f"config5 chunk={chunk} kv1 k", k_got[sel], f32("big/kv1_k")[:, :, :, 0][sel]) < TOL
    assert _err(f"config5 chunk={chunk} kv1 v", v_got[sel], f32("big/kv1_v")[:, :, :, 0][sel]) < TOL


def test_config5_sampled_positions_inside_chunks(real_gpu):
    """Logits at sampled positions INSIDE chunks 1 and 3 (not only the last position): step() without last_only."""
    model, cfg = real_gpu
    ids = torch.from_numpy(R["big/ids"].astype(np.int64)).to(DEV)
    lens = [int(v) for v in R["big/lens"]]
    S = ids.shape[1]
    valid = torch.zeros(4, S, dtype=torch.bool, device=DEV)
    for b, n in enumerate(lens):
        valid[b, S - n:] = True
    positions = torch.where(valid, torch.cumsum(valid.long(), 1), torch.zeros(4, S, dtype=torch.long, device=DEV))
    cache = model.new_cache(4, S)
    t = torch.arange(S, device=DEV)
    rows_s = torch.from_numpy(R["big/sample_rows"]).to(DEV)
    with torch.no_grad():
        for ci, c0 in enumerate(range(0, S, 512)):
            rows = torch.arange(c0, c0 + 512, device=DEV)
            mask = ((t[None, None, :] > rows[None, :, None]) | ~valid[:, None, :]).float() * -1e10
            logits = model.step(ids[:, c0:c0 + 512], cache, rows, positions[:, c0:c0 + 512], mask, kv_len=c0 + 512)
            if ci in (1, 3):
                got = t2n(logits[:, rows_s])
                want = f32(f"big/chunk{ci}_logits")
                sel = valid[:, c0 + rows_s].cpu().numpy()                      # pad query rows are undefined by design
                assert _err(f"config5 chunk {ci} sampled rows", got[sel], want[sel]) < TOL


def test_default_prefill_chunking_bounds_a_pass_at_8192_rows(real_gpu):
    """DecodeSession.prefill without a chunk length: the whole prompt while batch x positions <= 8192, chunks of 8192 // batch
    positions beyond that - here 4 x 2300 = a pass of 2048 and one of 252 positions (the second one attends over a 2048-row
    prefix through the one-launch attention) - against explicit chunks of 512."""
    model, cfg = real_gpu
    g = torch.Generator().manual_seed(23)
    ids = torch.randint(0, cfg.vocab_size, (4, 2300), generator=g)
    a = DecodeSession(model, 4, 2304, use_graph=False)
    before = _lib.launch_count()
    la = a.prefill(ids)
    assert a.length == 2300 and _lib.launch_count() > before
    b = DecodeSession(model, 4, 2304, use_graph=False)
    lb = b.prefill(ids, chunk=512)
    assert _err("default chunking vs chunks of 512, last logits", t2n(la), t2n(lb)) < TOL
    assert _err("default chunking vs chunks of 512, cache rows", t2n(a.cache.k[1][:, :2300]), t2n(b.cache.k[1][:, :2300])) < TOL
