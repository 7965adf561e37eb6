"""Round-3 host-path tests on the GPU: the pre-bound launch plans behind DynamicQuantizeLinear.forward and the fused decode
step (same bits as the checked wrappers; every change of buffers / operands sends the call back through the checks), the
bf16 default arithmetic of the module (<= 1e-3 of the oracle), graph re-capture after the weights moved
(ADVICE r2), and the one-position chunk of a left-padded batched prefill."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import qlinear_oracle as O  # noqa: E402
from chatglm_q_amd import _lib  # noqa: E402
from chatglm_q_amd import model as M  # noqa: E402
from chatglm_q_amd.decoder import ChatGLMDecoder, DecodeSession  # noqa: E402
from chatglm_q_amd.int4 import hip_ops as h4  # noqa: E402
from chatglm_q_amd.int4 import qlinear as q4  # noqa: E402
from chatglm_q_amd.int8 import qlinear as q8  # noqa: E402
from test_parity_gpu import _rand_w4, t2n  # noqa: E402

DEV = "cuda:0"
TDT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}


def _layer4(K, N, dt, seed, bias=True):
    layer = q4.DynamicQuantizeLinear(K, N, bias=bias, dtype=TDT[dt])
    qw, sc = _rand_w4(K, N, dt, seed)
    b = (torch.randn(N, generator=torch.Generator().manual_seed(seed + 1)) * 0.1).to(TDT[dt]) if bias else None
    layer.apply_weights_(qw, sc, b)
    return layer.to(DEV), qw, sc, b


@pytest.mark.parametrize("rows", [1, 2, 3, 4])
def test_bf16_module_default_arithmetic_is_within_1e3(rows):
    """The path DynamicQuantizeLinear.forward takes for bf16 at 1..4 rows (strict per-weight rounding unless QLINEAR_STRICT=0)
    against the oracle's bf16 restatement of the reference (chatglm_q/int4/triton_ops.py:72-73) at the headline shape."""
    if _lib.STRICT_MODE == "off":
        pytest.skip("QLINEAR_STRICT=0: the exact-dequant arithmetic was asked for")
    layer, qw, sc, b = _layer4(4096, 4096, "bf16", 300 + rows)
    x = torch.randn(rows, 4096, generator=torch.Generator().manual_seed(rows)).bfloat16()
    with torch.no_grad():
        y1 = layer(x.to(DEV))                  # checked path (builds the plan)
        y2 = layer(x.to(DEV))                  # pre-bound plan
    ref = O.w4_matmul(t2n(x), qw.numpy(), t2n(sc), t2n(b), dtype="bf16")
    err = O.rel_l2(t2n(y1), ref)
    print(f"[bf16 default] rows={rows}: rel-L2 {err:.3e}")
    assert err <= 1e-3
    assert torch.equal(y1, y2)


@pytest.mark.parametrize("dt,rows", [("f16", 1), ("f16", 2), ("f16", 7), ("f16", 300), ("bf16", 1), ("f32", 1), ("f32", 3)])
def test_forward_plan_equals_checked_path_and_revalidates(dt, rows):
    K, N = 1024, 768
    layer, qw, sc, b = _layer4(K, N, dt, 17)
    x = torch.randn(rows, K, device=DEV).to(TDT[dt])
    with torch.no_grad():
        y0 = layer(x)
        assert layer._plans, "the checked path did not leave a pre-bound launch behind"
        before = _lib.launch_count()
        y1 = layer(x)
        assert _lib.launch_count() > before and torch.equal(y0, y1)
        ref = O.w4_matmul(t2n(x), qw.numpy(), t2n(sc), t2n(b), dtype=dt)
        assert O.rel_l2(t2n(y1), ref) <= (2e-6 if dt == "f32" else 1e-3)
        # (1) another shape with the same element count, a non-contiguous view, a misaligned view: still right
        if rows % 2 == 0:
            assert torch.equal(layer(x.view(2, rows // 2, K)).reshape(rows, N), y0)
        xt = torch.randn(K, rows, device=DEV).to(TDT[dt]).t()
        assert O.rel_l2(t2n(layer(xt)), O.w4_matmul(t2n(xt), qw.numpy(), t2n(sc), t2n(b), dtype=dt)) <= (2e-6 if dt == "f32" else 1e-3)
        # (2) the reference loader's in-place refill: the plan must notice (version counters) and the result must follow
        qw2, sc2 = _rand_w4(K, N, dt, 18)
        layer.state_dict()["weight"].copy_(qw2.to(DEV))
        layer.state_dict()["weight_scale"].copy_(sc2.to(DEV))
        y2 = layer(x)
        assert O.rel_l2(t2n(y2), O.w4_matmul(t2n(x), qw2.numpy(), t2n(sc2), t2n(b), dtype=dt)) <= (2e-6 if dt == "f32" else 1e-3)
        # (3) invalidate() drops the plans; the checks fire again on the next call
        layer.invalidate()
        assert not layer._plans
        with pytest.raises(AssertionError):
            layer(x.to(torch.float32 if dt != "f32" else torch.float16))          # dtype check of the reference wrapper
        with pytest.raises(AssertionError):
            layer(torch.randn(rows, K // 2, device=DEV).to(TDT[dt]))              # K mismatch
        assert torch.equal(layer(x), y2)
        # (4) a live plan does not shadow the checks either: wrong dtype / wrong K reach the checked path
        assert layer._plans
        with pytest.raises(AssertionError):
            layer(x.to(torch.float32 if dt != "f32" else torch.float16))
        # (5) bias replaced by assignment: plans dropped, new bias used
        if b is not None:
            layer.bias = torch.zeros_like(layer.bias)
            assert not layer._plans
            assert O.rel_l2(t2n(layer(x)), O.w4_matmul(t2n(x), qw2.numpy(), t2n(sc2), None, dtype=dt)) <= (2e-6 if dt == "f32" else 1e-3)


def test_int8_forward_plan_equals_checked_path():
    K, N = 1024, 512
    g = torch.Generator().manual_seed(4)
    w = torch.randint(-127, 128, (N, K), dtype=torch.int8, generator=g)
    s = (torch.rand(N, generator=g) * 0.01 + 0.001).half()
    layer = q8.DynamicQuantizeLinear(K, N, bias=False, dtype=torch.float16)
    layer.apply_weights_(w, s)
    layer = layer.to(DEV)
    for rows in (1, 2, 5, 64):
        x = torch.randn(rows, K, device=DEV).half()
        with torch.no_grad():
            y0 = layer(x)
            assert x.numel() in layer._plans
            y1 = layer(x)
        assert torch.equal(y0, y1)
        ref = O.w8_matmul(t2n(x), np.ascontiguousarray(w.numpy().T), s.numpy(), None, dtype="f16")
        assert O.rel_l2(t2n(y1), ref) <= 1e-3
    w2 = torch.randint(-127, 128, (N, K), dtype=torch.int8, generator=g)
    layer.state_dict()["weight"].copy_(w2.to(DEV))
    with torch.no_grad():
        y2 = layer(x)
    assert O.rel_l2(t2n(y2), O.w8_matmul(t2n(x), np.ascontiguousarray(w2.numpy().T), s.numpy(), None, dtype="f16")) <= 1e-3


def _tiny(dtype=torch.float16, kind="int4", seed=5):
    cfg = M.ChatGLM2Config(hidden_size=256, inner_hidden_size=384, head_hidden_size=32, num_multi_query_groups=2,
                           num_attention_heads=8, num_layers=2, vocab_size=320, max_sequence_length=96)
    with torch.device(DEV):
        model = (M.create_quant_int4_model if kind == "int4" else M.create_quant_int8_model)(cfg, dtype=dtype)
    M.fill_synthetic_(model, seed)
    return model.eval(), cfg


@pytest.mark.parametrize("kind", ["int4", "int8"])
def test_one_row_step_plans_equal_checked_wrappers(kind):
    """Decode steps with the pre-bound launches (second step onwards) against a session whose plans are dropped before every
    step: same logits bit for bit."""
    model, cfg = _tiny(kind=kind)
    ids = torch.randint(0, 320, (1, 9), device=DEV)

    def run(drop):
        sess = DecodeSession(model, 1, 32, use_graph=False)
        logits = [sess.prefill(ids)]
        sess.tok.copy_(logits[0].argmax(-1, keepdim=True))
        for _ in range(4):
            if drop:
                for m in model.modules():
                    if hasattr(m, "_fast"):
                        m._fast.clear()
                sess.cache.att_plans.clear()
            logits.append(sess.decode_step(greedy=True).clone())
        return torch.stack(logits)

    a = run(drop=True)
    b = run(drop=False)
    assert any(m._fast for m in model.modules() if hasattr(m, "_fast"))
    assert torch.equal(a, b)


def test_graph_is_recaptured_when_the_weights_move():
    """a decoder reuses its session and graph; apply_weights_ / load_state_dict drop the derived layouts the graph
    baked in.  The next generation must re-capture - and produce what an eager run on the new weights produces."""
    model, cfg = _tiny()
    dec = ChatGLMDecoder(None, model)
    prefix = [3, 17, 200, 5, 77]
    kw = dict(max_generated_tokens=10, greedy=True, ignore_eos=True)
    first = list(dec.generate_ids(prefix, use_graph=True, **kw))
    sess = dec._session
    g0 = sess.graph
    assert g0 is not None
    again = list(dec.generate_ids(prefix, use_graph=True, **kw))
    assert again == first and dec._session.graph is g0                         # nothing moved: the graph is kept
    M.fill_synthetic_(model, 99)                                               # in-place refill of every buffer (version bump)
    want = list(ChatGLMDecoder(None, model).generate_ids(prefix, use_graph=False, **kw))
    got = list(dec.generate_ids(prefix, use_graph=True, **kw))
    assert got == want
    assert dec._session.graph is not g0
    for mod in model.modules():                                                # explicit invalidate(): same story
        if hasattr(mod, "invalidate"):
            mod.invalidate()
    g1 = dec._session.graph
    assert list(dec.generate_ids(prefix, use_graph=True, **kw)) == want
    assert dec._session.graph is not g1


def test_launch_ahead_greedy_decoding_yields_the_same_tokens_and_stops_at_the_end_token(monkeypatch):
    """Round 5: on a captured greedy step generate_ids launches step k + 1 before it reads token k (the graph feeds its own argmax back).  Same
    tokens as the launch-read-launch loop and as the eager loop; with an end token the generation stops AT it (the speculative step behind it is
    thrown away), a following generation on the same session is unaffected, and the budget / the cache's end are respected."""
    from chatglm_q_amd import decoder as Dm
    model, cfg = _tiny()
    prefix = [3, 17, 200, 5, 77]
    kw = dict(max_generated_tokens=12, greedy=True, ignore_eos=True)
    eager = list(ChatGLMDecoder(None, model).generate_ids(prefix, use_graph=False, **kw))
    monkeypatch.setattr(Dm, "AHEAD_LAUNCH", False)
    plain = list(ChatGLMDecoder(None, model).generate_ids(prefix, use_graph=True, **kw))
    monkeypatch.setattr(Dm, "AHEAD_LAUNCH", True)
    dec = ChatGLMDecoder(None, model)
    ahead = list(dec.generate_ids(prefix, use_graph=True, **kw))
    assert ahead == plain == eager and len(ahead) == 12
    stop = 5
    if eager[stop] not in eager[:stop]:                       # an end token in the middle: stop at it, in both loops
        dec_eos = ChatGLMDecoder(None, model, eos_token_id=eager[stop])
        assert list(dec_eos.generate_ids(prefix, max_generated_tokens=12, greedy=True, use_graph=True)) == eager[:stop + 1]
        assert list(dec_eos.generate_ids(prefix, max_generated_tokens=12, greedy=True, use_graph=True)) == eager[:stop + 1]   # session reused
    assert list(dec.generate_ids(prefix, use_graph=True, **kw)) == eager                         # ... and after a full run
    assert list(dec.generate_ids(prefix, max_generated_tokens=1, greedy=True, ignore_eos=True, use_graph=True)) == eager[:1]
    assert list(dec.generate_ids(prefix, max_generated_tokens=2, greedy=True, ignore_eos=True, use_graph=True)) == eager[:2]
    full = cfg.max_sequence_length - len(prefix)               # up to the cache's last row
    out = list(dec.generate_ids(prefix, max_generated_tokens=10 ** 6, greedy=True, ignore_eos=True, use_graph=True))
    assert len(out) == full and out[:12] == eager


def test_two_live_generators_do_not_share_a_session():
    model, cfg = _tiny()
    dec = ChatGLMDecoder(None, model)
    kw = dict(max_generated_tokens=6, greedy=True, ignore_eos=True, use_graph=False)
    want_a = list(dec.generate_ids([1, 2, 3], **kw))
    want_b = list(dec.generate_ids([9, 8, 7, 6], **kw))
    ga, gb = dec.generate_ids([1, 2, 3], **kw), dec.generate_ids([9, 8, 7, 6], **kw)
    got_a, got_b = [], []
    for _ in range(6):
        got_a.append(next(ga))
        got_b.append(next(gb))
    assert got_a == want_a and got_b == want_b


def test_budget_respects_the_models_sequence_limit():
    model, cfg = _tiny()
    dec = ChatGLMDecoder(None, model, max_sequence_length=4096)               # larger than the model's rotary table
    out = list(dec.generate_ids(list(range(90)), max_generated_tokens=50, greedy=True, ignore_eos=True, use_graph=False))
    assert len(out) == cfg.max_sequence_length - 90


@pytest.mark.parametrize("S,chunk", [(9, 8), (1, 8), (17, 4)])
def test_left_padded_batched_prefill_with_a_one_position_chunk(S, chunk, monkeypatch):
    """S % chunk == 1 leaves a one-position chunk, which takes the one-position attention kernel with the
    FULL-capacity mask (a sliced + compacted mask made rows b >= 1 read other sequences' columns).  Fused ops against the
    plain-torch graph of the same model."""
    model, cfg = _tiny()
    B = 4
    lens = [S, max(1, S - 3), max(1, S // 2), 1]
    g = torch.Generator().manual_seed(S)
    ids = torch.randint(1, 320, (B, S), generator=g)
    attn = torch.zeros(B, S, dtype=torch.long)
    for b, n in enumerate(lens):
        attn[b, S - n:] = 1
        ids[b, : S - n] = 0
    outs = {}
    for fused in (False, True):
        monkeypatch.setattr(M, "FUSED_DECODE_OPS", fused)
        sess = DecodeSession(model, B, 32, use_graph=False)
        # dirty cache rows beyond the prefix: a stale-row read would show
        for k, v in zip(sess.cache.k, sess.cache.v):
            k.normal_(0, 3)
            v.normal_(0, 3)
        logits = [sess.prefill(ids, chunk=chunk, attention_mask=attn)]
        sess.tok.copy_(logits[0].argmax(-1, keepdim=True))
        for _ in range(2):
            logits.append(sess.decode_step(greedy=True).clone())
        outs[fused] = torch.stack([l.float() for l in logits])
    assert O.rel_l2(t2n(outs[True]), t2n(outs[False])) < 5e-3


def test_decode_attention_rope_rejects_a_compacted_mask():
    from chatglm_q_amd import fused_ops as F_
    B, H, Gq, D, cap = 2, 8, 2, 32, 64
    qkv = torch.randn(B, 1, (H + 2 * Gq) * D, device=DEV).half()
    table = M.rotary_table(D, cap + 1).half().view(cap + 1, -1).to(DEV)
    pos = torch.ones(B, 1, dtype=torch.long, device=DEV)
    widx = torch.zeros(1, dtype=torch.long, device=DEV)
    k = torch.zeros(B, cap, Gq, D, device=DEV).half()
    v = torch.zeros_like(k)
    mask = torch.zeros(B, 1, cap, device=DEV)
    F_.decode_attention_rope(qkv, table, pos, widx, k, v, mask, H, Gq, D)
    with pytest.raises(ValueError):
        F_.decode_attention_rope(qkv, table, pos, widx, k, v, mask[..., :40].contiguous(), H, Gq, D)
    with pytest.raises(ValueError):
        F_.decode_attention_rope(qkv, table, pos, widx, k, v, mask[..., :40], H, Gq, D)


def test_release_frees_and_rebuilds_derived_layouts():
    layer, qw, sc, b = _layer4(1024, 768, "f16", 33)
    x1, x8 = torch.randn(1, 1024, device=DEV).half(), torch.randn(24, 1024, device=DEV).half()   # (24 rows: past what part 1 serves)
    with torch.no_grad():
        y1, y8 = layer(x1), layer(x8)
    nb = layer.derived_nbytes()
    assert nb["packed"] > 0 and nb["tiled"] > 0 and nb["canonical"] > 0
    layer.release("tiled")
    assert layer.derived_nbytes()["tiled"] == 0 and layer.derived_nbytes()["packed"] == nb["packed"]
    with torch.no_grad():
        assert torch.equal(layer(x1), y1)
        assert layer.derived_nbytes()["tiled"] == 0           # one row never needs part 2
        assert torch.equal(layer(x8), y8)                      # rebuilt on demand
    assert layer.derived_nbytes()["tiled"] == nb["tiled"]
    with pytest.raises(ValueError):
        layer.release("nonsense")


def test_decode_only_session_frees_prefill_layouts_and_decodes_the_same():
    model, cfg = _tiny()
    ids = torch.randint(0, 320, (1, 40), device=DEV)

    def run(decode_only):
        for m in model.modules():
            if hasattr(m, "invalidate"):
                m.invalidate()
        sess = DecodeSession(model, 1, 64, use_graph=True, decode_only=decode_only)
        lg = sess.prefill(ids)                                   # 40 rows: builds part 2 of every layer
        sess.tok.copy_(lg.argmax(-1, keepdim=True))
        sess.capture(greedy=True)
        out = [int(sess.tok.item())]
        for _ in range(6):
            sess.decode_step(greedy=True)
            out.append(int(sess.tok.item()))
        resident = sum(sum(v for k, v in m.derived_nbytes().items() if k != "canonical") for m in model.modules()
                       if hasattr(m, "derived_nbytes"))
        return out, resident

    want, full = run(False)
    got, lean = run(True)
    assert got == want
    assert lean < full
    w_in = model.layers[0].ffn.w_in
    assert w_in.derived_nbytes()["tiled"] == 0 and w_in.derived_nbytes()["packed"] == 0 and w_in.derived_nbytes()["gated"] > 0


def test_decode_only_session_serves_a_batch_of_8_from_part_1_alone():
    """Round 5: 3..16 rows run on part 1 of the derived layout (w4_rows16.hip), so a decode-only session that dropped the tile-major copies
    after its prefill decodes a batch of 8 without rebuilding them - and chooses the same tokens as a session that kept everything."""
    model, cfg = _tiny()
    ids = torch.randint(0, 320, (8, 12), device=DEV)

    def run(decode_only):
        for m in model.modules():
            if hasattr(m, "invalidate"):
                m.invalidate()
        sess = DecodeSession(model, 8, 64, use_graph=True, decode_only=decode_only)
        lg = sess.prefill(ids)                                   # 96 rows: part 2 of every layer
        sess.tok.copy_(lg.argmax(-1, keepdim=True))
        sess.capture(greedy=True)
        out = [sess.tok.flatten().tolist()]
        for _ in range(5):
            sess.decode_step(greedy=True)
            out.append(sess.tok.flatten().tolist())
        tiled = sum(m.derived_nbytes()["tiled"] + m.derived_nbytes()["gated_tiled"] for m in model.modules() if hasattr(m, "derived_nbytes"))
        return out, tiled

    want, tiled_full = run(False)
    got, tiled_lean = run(True)
    assert got == want
    assert tiled_full > 0 and tiled_lean == 0


# ---- quantising producers (round 3): RMSNorm / SiLU * gate emit the int8 rows + scales of the int8-activation GEMM behind them ----
@pytest.mark.parametrize("rows,dim", [(5, 4096), (1, 256), (37, 1024), (3, 13696 // 2), (2, 16384), (9, 264)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("with_delta", [False, True])
def test_rmsnorm_quant_equals_norm_then_quantiser_and_the_oracle(rows, dim, dtype, with_delta):
    """qlinear_rmsnorm_quant_i8: int8 rows and scales bit for bit what the quantiser launch (and the oracle's quantize_int8
    restatement, chatglm_q/int8/quantizer.py:11-19) make of the rounded 16-bit output row; h and the optional 16-bit row equal
    the plain launches'.  Includes an all-zero row (scale floor) and a row of ties."""
    import numpy as np
    from oracle import qlinear_oracle as O
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd.int8 import hip_ops as h8
    g = torch.Generator(device=DEV).manual_seed(rows * 131 + dim)
    x = torch.randn(rows, dim, device=DEV, generator=g).to(dtype)
    x[0] = 0
    delta = (torch.randn(rows, dim, device=DEV, generator=g) * 0.3).to(dtype) if with_delta else None
    if with_delta:
        delta[0] = 0
    w = (1 + 0.1 * torch.randn(dim, device=DEV, generator=g)).to(dtype)
    if with_delta:
        want_h, want_out = F_.add_rmsnorm(x, delta, w, 1e-5)
    else:
        want_h, want_out = x, F_.rmsnorm(x, w, 1e-5)
    want_q, want_s = h8.act_quant_rowwise(want_out)
    before = _lib.launch_count()
    h, a_q, a_s, out = F_.rmsnorm_quant(x, w, 1e-5, delta, want_out=True)
    assert _lib.launch_count() - before == 1
    assert torch.equal(h, want_h) and torch.equal(out, want_out)
    assert torch.equal(a_q, want_q) and torch.equal(a_s, want_s)
    q_ref, s_ref = O.act_quant_rowwise(want_out.float().cpu().numpy() if dtype == torch.bfloat16 else want_out.cpu().numpy())
    assert np.array_equal(a_q.cpu().numpy(), q_ref) and np.array_equal(a_s.cpu().numpy(), s_ref)
    _, a_q2, a_s2, none = F_.rmsnorm_quant(x, w, 1e-5, delta)                # without the 16-bit row
    assert none is None and torch.equal(a_q2, want_q) and torch.equal(a_s2, want_s)


@pytest.mark.parametrize("rows,hidden", [(4, 13696), (1, 384), (33, 1024), (2, 16384), (7, 8)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_silu_mul_quant_equals_silu_mul_then_quantiser(rows, hidden, dtype):
    import numpy as np
    from oracle import qlinear_oracle as O
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd.int8 import hip_ops as h8
    g = torch.Generator(device=DEV).manual_seed(rows * 7 + hidden)
    x = (torch.randn(rows, 2 * hidden + 8, device=DEV, generator=g) * 2).to(dtype)[:, :2 * hidden]     # row stride != 2 * hidden
    want = F_.silu_mul(x, hidden)
    want_q, want_s = h8.act_quant_rowwise(want)
    before = _lib.launch_count()
    a_q, a_s, out = F_.silu_mul_quant(x, hidden, want_out=True)
    assert _lib.launch_count() - before == 1
    assert torch.equal(out, want) and torch.equal(a_q, want_q) and torch.equal(a_s, want_s)
    q_ref, s_ref = O.act_quant_rowwise(want.float().cpu().numpy() if dtype == torch.bfloat16 else want.cpu().numpy())
    assert np.array_equal(a_q.cpu().numpy(), q_ref) and np.array_equal(a_s.cpu().numpy(), s_ref)
    a_q2, a_s2, none = F_.silu_mul_quant(x, hidden)
    assert none is None and torch.equal(a_q2, want_q) and torch.equal(a_s2, want_s)


def test_int8_activation_model_prefill_with_quantising_producers_is_bit_equal(monkeypatch):
    """An int8 model whose QLinear modules take int8 activations (``act_quant``): the prefill with the quantising producers
    (3 fewer launches per layer, no 16-bit copies of the rows) produces the logits of the path with separate quantiser
    launches bit for bit, and the module's ``forward_quantized`` equals its ``forward``."""
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd import model as M
    cfg = M.ChatGLM2Config(hidden_size=256, inner_hidden_size=384, head_hidden_size=32, num_multi_query_groups=2,
                           num_attention_heads=8, num_layers=2, vocab_size=320, max_sequence_length=64)
    with torch.device(DEV):
        model = M.create_quant_int8_model(cfg, dtype=torch.float16)
    M.fill_synthetic_(model, 5)
    model.eval()
    for m in model.modules():
        if hasattr(m, "act_quant"):
            m.act_quant = True
    ids = torch.randint(0, 320, (2, 24), device=DEV)
    outs, launches = {}, {}
    DecodeSession(model, 2, 32, use_graph=False).prefill(ids)                # builds the tile-major weight copies
    for pre in (False, True):
        monkeypatch.setattr(M, "PREQUANT", pre)
        sess = DecodeSession(model, 2, 32, use_graph=False)
        before = _lib.launch_count()
        outs[pre] = sess.prefill(ids).clone()
        launches[pre] = _lib.launch_count() - before
    assert torch.equal(outs[True], outs[False])
    assert launches[False] - launches[True] == 3 * cfg.num_layers, launches
    mod = model.layers[0].attn.qkv_proj
    x = torch.randn(5, cfg.hidden_size, device=DEV).half()
    w = model.layers[0].attn_ln.weight
    _, a_q, a_s, out = F_.rmsnorm_quant(x, w, 1e-5, want_out=True)
    with torch.no_grad():
        assert torch.equal(mod.forward_quantized(a_q, a_s), mod(out))


def test_default_generate_samples_on_the_device_inside_the_captured_step(monkeypatch):
    """Round 6: the reference's only decoding mode (chatglm_q/decoder.py:85: top_p_sampling per token) is the default of generate_ids and
    runs as one launch inside the captured step.  (a) graph + launch-ahead, graph without it, the device loop and the eager loop yield
    the same tokens for one seed; (b) every token is what the oracle's filter of THAT step's logits and the documented Philox draw give;
    (c) another seed changes the tokens, torch.manual_seed makes an unseeded run repeatable; (d) greedy and sampled graphs live side by
    side on one session; (e) the host-side torch sampler (device_sampler=False) still works and draws from the same support."""
    import _philox
    from chatglm_q_amd import decoder as Dm
    from oracle import qlinear_oracle as O
    model, cfg = _tiny()
    prefix = [3, 17, 200, 5, 77]
    kw = dict(max_generated_tokens=12, ignore_eos=True, top_k=20, top_p=0.9, temperature=1.5)
    dec = ChatGLMDecoder(None, model)
    ahead = list(dec.generate_ids(prefix, use_graph=True, seed=41, **kw))
    sess = dec._session
    assert sess._captured_mode == "sample" and sess.graph is not None
    eager = list(ChatGLMDecoder(None, model).generate_ids(prefix, use_graph=False, seed=41, **kw))
    loop = list(ChatGLMDecoder(None, model).generate_ids(prefix, use_graph=True, seed=41, sync_every_token=False, **kw))
    monkeypatch.setattr(Dm, "AHEAD_LAUNCH", False)
    plain = list(ChatGLMDecoder(None, model).generate_ids(prefix, use_graph=True, seed=41, **kw))
    monkeypatch.setattr(Dm, "AHEAD_LAUNCH", True)
    assert ahead == plain == eager == loop and len(ahead) == 12
    assert int(sess.rng_state[1].item()) >= 12                                 # one draw per token (+ at most the speculative step)
    # (b) step by step against the oracle: feed the same tokens through an eager session and redo the sampler on the host
    chk = DecodeSession(model, 1, 64, use_graph=False)
    logits = chk.prefill(torch.tensor([prefix], device=DEV))
    for step, tok in enumerate(ahead):
        p, idx = O.top_p_filter(logits[0].float().cpu().numpy(), 20, 0.9, 1.5)
        u = float(_philox.uniform(41, step, 0))
        cdf = np.cumsum(p.astype(np.float64))
        j = min(int(np.searchsorted(cdf, u * cdf[-1], side="right")), int((p > 0).sum()) - 1)
        edge = min(abs(u * cdf[-1] - c) for c in cdf[max(0, j - 1): j + 1])
        assert tok == idx[j] or (edge < 1e-5 and tok in idx[max(0, j - 1): j + 2]), (step, tok, idx[j])
        logits = chk.decode_step(torch.tensor([[tok]], device=DEV), greedy=False)
    # (c)
    other = list(dec.generate_ids(prefix, use_graph=True, seed=42, **kw))
    assert other != ahead
    torch.manual_seed(7)
    r1 = list(dec.generate_ids(prefix, use_graph=True, **kw))
    torch.manual_seed(7)
    r2 = list(dec.generate_ids(prefix, use_graph=True, **kw))
    assert r1 == r2
    # (d)
    g_sample = sess.graph
    greedy = list(dec.generate_ids(prefix, use_graph=True, greedy=True, max_generated_tokens=12, ignore_eos=True))
    assert dec._session is sess and sess._captured_mode == "greedy" and sess.graph is not g_sample
    assert greedy == list(ChatGLMDecoder(None, model).generate_ids(prefix, use_graph=False, greedy=True, max_generated_tokens=12, ignore_eos=True))
    assert list(dec.generate_ids(prefix, use_graph=True, seed=41, **kw)) == ahead and sess.graph is g_sample
    top1 = list(dec.generate_ids(prefix, use_graph=True, seed=5, max_generated_tokens=12, ignore_eos=True, top_k=1))
    assert top1 == greedy                                                       # top_k = 1 through the sampler graph == argmax
    # (e)
    host = list(dec.generate_ids(prefix, use_graph=True, device_sampler=False, **kw))
    assert len(host) == 12 and all(0 <= t < cfg.vocab_size for t in host)
    # an end token stops a sampled generation AT it
    stop = 5
    if ahead[stop] not in ahead[:stop]:
        dec_eos = ChatGLMDecoder(None, model, eos_token_id=ahead[stop])
        kw2 = dict(kw, ignore_eos=False)
        assert list(dec_eos.generate_ids(prefix, use_graph=True, seed=41, **kw2)) == ahead[:stop + 1]
        assert list(dec_eos.generate_ids(prefix, use_graph=True, seed=41, **kw2)) == ahead[:stop + 1]


def test_routing_answers_follow_a_dispatch_reload(monkeypatch):
    """rows_on_tiled / the workspace size were lru_cached on the shape alone; after qlinear_dispatch_reload() with
    `norows16` Python still said "part 1 serves 5..16 rows" and handed a part-1-only buffer to a kernel that reads part 2.  The cached
    answers are keyed on the library's dispatch flags now: the module serves 8 rows correctly under either setting."""
    lib = _lib.get_lib()
    K, N = 4096, 4096
    qw, sc = _rand_w4(K, N, "f16", seed=5)
    layer = q4.DynamicQuantizeLinear(K, N, bias=False, dtype=torch.float16, device=DEV)
    layer.apply_weights_(qw.to(DEV), sc.to(DEV))
    x = torch.randn(8, K, device=DEV, dtype=torch.float16)
    ref = O.w4_matmul(t2n(x.cpu()), qw.numpy(), t2n(sc), None, dtype="f16")
    assert h4.rows_on_tiled(8, N, K, torch.float16, False) is False           # round 5's one-launch kernel on part 1
    y0 = layer(x)
    try:
        monkeypatch.setenv("QLINEAR_DISPATCH", "norows16")
        lib.qlinear_dispatch_reload()
        assert h4.rows_on_tiled(8, N, K, torch.float16, False) is True        # ... the few-row kernel on part 2 without it
        fresh = q4.DynamicQuantizeLinear(K, N, bias=False, dtype=torch.float16, device=DEV)
        fresh.apply_weights_(qw.to(DEV), sc.to(DEV))
        y1 = fresh(x)                                                          # builds what THIS routing reads
        y2 = layer(x)                                                          # the module that had only part 1
    finally:
        monkeypatch.delenv("QLINEAR_DISPATCH")
        lib.qlinear_dispatch_reload()
    assert h4.rows_on_tiled(8, N, K, torch.float16, False) is False
    for y in (y0, y1, y2):
        assert O.rel_l2(t2n(y), ref) <= 2e-4


def test_drop_canonical_never_keeps_a_stale_gated_copy_as_the_only_copy():
    """w_in after a decode-only release holds only its gate-interleaved part 1; an in-place weight update with no forward in
    between left that copy stale, and drop_canonical() would have made it the only copy - state_dict() then returned the OLD weights."""
    K, hidden = 256, 96
    g = torch.Generator(device=DEV).manual_seed(3)
    layer = q4.DynamicQuantizeLinear(K, 2 * hidden, bias=False, dtype=torch.float16, device=DEV)
    layer.weight.copy_(torch.randint(0, 256, layer.weight.shape, dtype=torch.uint8, device=DEV, generator=g))
    layer.weight_scale.copy_((torch.rand(layer.weight_scale.shape, device=DEV, generator=g) * 0.01 + 0.002).half())
    layer.gated_packed(hidden)
    if layer._packed is not None:
        layer.release("packed")
    assert layer._packed is None and layer._gated is not None
    new_w = torch.randint(0, 256, layer.weight.shape, dtype=torch.uint8, device=DEV, generator=g)
    layer.weight.copy_(new_w)                                                  # in place: version bump, no forward
    assert layer.drop_canonical() > 0
    sd = layer.state_dict()
    assert torch.equal(sd["weight"], new_w)
