"""BASELINE config 4 at the WHOLE ChatGLM2-6B geometry - 28 layers, hidden 4096, FFN 13696, 32 heads x 128, 2 groups, vocabulary
65024 - on the GPU against logits and cache rows the REFERENCE model produced on the CPU for the same seeded weights
(tests/golden/model_full.npz; generator tests/golden/make_golden.py::gen_model_full, which imports the reference and follows
chatglm_q/decoder.py:65-108's call pattern: one prefill of 32 ids, then four cached one-token steps).  Nothing of the 3.4 GB of
weights is stored: tests/_golden.py::fill_seeded_ rebuilds them from the seed.

Tolerance: fp16 end to end through 28 layers.  Each layer output is ONE fp16 rounding of 4096 / 13696-term sums that the two
implementations add in different orders (torch's CPU GEMM there, MFMA / v_dot2c here); the two-layer fixtures measure 4e-4 .. 1e-3
against a 2e-3 bar, and the distance grows like the square root of the depth: the bar here is 6e-3 on the logits (relative L2), the
measured values are printed (pytest -s).  The greedy choice must agree wherever the reference's own top-2 margin exceeds that noise."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import _golden as G  # noqa: E402
from oracle import qlinear_oracle as O  # noqa: E402
from chatglm_q_amd import _lib  # noqa: E402
from chatglm_q_amd import model as M  # noqa: E402
from chatglm_q_amd.decoder import DecodeSession  # noqa: E402

DEV = "cuda:0"
F = G.load("model_full.npz")
TOL = 6e-3


def t2n(t):
    return t.detach().float().cpu().numpy()


def f32(name):
    return F[name].astype(np.float32)


@pytest.fixture(scope="module")
def full_gpu():
    keys = ("hidden_size", "inner_hidden_size", "head_hidden_size", "num_multi_query_groups", "num_attention_heads", "num_layers",
            "vocab_size", "max_sequence_length")
    assert [int(v) for v in F["config"]] == [G.FULL_CONFIG[k] for k in keys]
    assert G.FULL_CONFIG["num_layers"] == 28 and G.FULL_CONFIG["vocab_size"] == 65024
    cfg = M.ChatGLM2Config(**G.FULL_CONFIG)
    model = M.create_quant_int4_model(cfg, dtype=torch.float16)
    G.fill_seeded_(model.state_dict(), int(F["seed"][0]))
    for m in model.modules():
        if hasattr(m, "invalidate"):
            m.invalidate()
    return model.to(DEV).eval(), cfg


def _check_logits(name, got, want):
    err = O.rel_l2(got, want)
    top2 = np.sort(want[0])[-2:]
    margin = float(top2[1] - top2[0])
    noise = float(np.abs(got - want).max())
    print(f"[28-layer parity] {name}: rel-L2 {err:.3e}, max abs diff {noise:.3e}, reference top-2 margin {margin:.3e}")
    assert err < TOL, (name, err)
    if margin > 4 * noise:
        assert int(got[0].argmax()) == int(want[0].argmax()), name
    return err


@pytest.mark.parametrize("use_graph", [False, True])
def test_full_depth_generate_pattern_against_reference_logits(full_gpu, use_graph):
    """Prefill of 32 ids, then four teacher-forced one-row steps through the fused 5-launch layer, eager and replayed from ONE
    captured HIP graph per token (the product's decode loop), against the reference's logits; last layer's cache rows too."""
    model, cfg = full_gpu
    ids = torch.from_numpy(F["ids"].astype(np.int64))
    nxt = F["next_ids"].astype(np.int64)
    sess = DecodeSession(model, 1, 64, use_graph=use_graph)
    before = _lib.launch_count()
    last = sess.prefill(ids)
    _check_logits("prefill last position", t2n(last), f32("prefill_last_logits"))
    if use_graph:
        sess.tok.fill_(int(nxt[0]))
        sess.capture(greedy=False)
        assert sess.graph is not None
    for t in range(4):
        lg = sess.decode_step(torch.tensor([[int(nxt[t])]]), greedy=False)
        _check_logits(f"decode step {t} graph={use_graph}", t2n(lg), f32(f"decode_logits_{t}"))
    assert _lib.launch_count() > before
    pos = torch.from_numpy(F["kv_positions"].astype(np.int64)).to(DEV)
    L = cfg.num_layers - 1
    k_err = O.rel_l2(t2n(sess.cache.k[L][:, pos]), f32("kv_last_k")[:, :, :, 0])
    v_err = O.rel_l2(t2n(sess.cache.v[L][:, pos]), f32("kv_last_v")[:, :, :, 0])
    print(f"[28-layer parity] layer {L} cache rows: k {k_err:.3e}, v {v_err:.3e}")
    assert k_err < TOL and v_err < TOL


def test_full_depth_graph_decode_equals_eager_bit_for_bit(full_gpu):
    """The captured step replays the same launches on the same buffers: its logits equal the eager fused step's bit for bit."""
    model, cfg = full_gpu
    ids = torch.from_numpy(F["ids"].astype(np.int64))
    nxt = F["next_ids"].astype(np.int64)
    outs = []
    for use_graph in (False, True):
        sess = DecodeSession(model, 1, 64, use_graph=use_graph)
        sess.prefill(ids)
        if use_graph:
            sess.tok.fill_(int(nxt[0]))
            sess.capture(greedy=False)
        outs.append([sess.decode_step(torch.tensor([[int(nxt[t])]]), greedy=False).clone() for t in range(4)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
