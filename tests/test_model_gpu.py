"""End-to-end harness on the GPU: the tiny reference-pinned model through the HIP kernels, eager vs
HIP-graph decode, device-resident greedy loop."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import qlinear_oracle as O  # noqa: E402
from chatglm_q_amd import _lib  # noqa: E402
from chatglm_q_amd.decoder import ChatGLMDecoder  # noqa: E402
from test_model_cpu import Z, build, t2n  # noqa: E402

DEV = "cuda:0"
# hidden-128 model against the reference's own logits (tests/golden/tiny_model.npz).  Measured on MI355X (round 2,
# profiles/r02_real_dim_parity.log): fp32 4.8e-7 (prefill) / 2.6e-7 (decode) / 2.8e-7 (fused step); fp16 5.4e-4 / 6.9e-4 /
# 6.9e-4.  The bars are ~2x the measurement (round 1 had 2e-2 here); real layer dimensions: tests/test_model_real_gpu.py.
TINY_TOL = {"f32": 2e-6, "f16": 1.5e-3}


@pytest.mark.parametrize("dt", ["f32", "f16"])
def test_tiny_model_logits_through_hip_kernels(dt):
    model, cfg = build(dt, DEV)
    ids = torch.from_numpy(Z[f"{dt}/ids"]).to(DEV)
    nxt = torch.from_numpy(Z[f"{dt}/next_id"]).to(DEV)
    before = _lib.launch_count()
    with torch.no_grad():
        _, logits, kv = model(input_ids=ids)
        _, logits2, _ = model(input_ids=nxt, past_key_values=kv)
    assert _lib.launch_count() - before >= 2 * (4 * cfg.num_layers + 2)     # every QLinear + QEmbedding call
    tol = TINY_TOL[dt]
    e1 = O.rel_l2(t2n(logits), Z[f"{dt}/prefill_logits"].astype(np.float32))
    e2 = O.rel_l2(t2n(logits2), Z[f"{dt}/decode_logits"].astype(np.float32))
    print(f"[tiny parity] {dt}: prefill {e1:.3e} decode {e2:.3e}")
    assert e1 < tol and e2 < tol


def test_graph_decode_equals_eager_and_cpu():
    model_cpu, _ = build("f32")
    prefix = Z["f32/ids"][0].tolist()
    want = list(ChatGLMDecoder(None, model_cpu).generate_ids(prefix, max_generated_tokens=12, greedy=True, ignore_eos=True))
    model, _ = build("f32", DEV)
    dec = ChatGLMDecoder(None, model)
    eager = list(dec.generate_ids(prefix, max_generated_tokens=12, greedy=True, ignore_eos=True, use_graph=False))
    graph = list(dec.generate_ids(prefix, max_generated_tokens=12, greedy=True, ignore_eos=True, use_graph=True))
    loop = list(dec.generate_ids(prefix, max_generated_tokens=12, greedy=True, ignore_eos=True, use_graph=True,
                                 sync_every_token=False))
    assert eager == want and graph == want and loop == want
    sampled = list(dec.generate_ids(prefix, max_generated_tokens=6, greedy=False, ignore_eos=True))
    assert len(sampled) == 6 and all(0 <= t < 256 for t in sampled)


@pytest.mark.parametrize("dt", ["f32", "f16"])
def test_fused_decode_ops_match_the_torch_graph(dt, monkeypatch):
    """csrc/decode_ops.hip against the plain-torch formulation of the same ops (model.py), step by step."""
    from chatglm_q_amd import model as M
    from chatglm_q_amd.decoder import DecodeSession
    model, cfg = build(dt, DEV)
    ids = torch.from_numpy(Z[f"{dt}/ids"]).to(DEV)
    outs = {}
    for fused in (False, True):
        monkeypatch.setattr(M, "FUSED_DECODE_OPS", fused)
        sess = DecodeSession(model, 1, 64, use_graph=False)
        logits = [sess.prefill(ids, chunk=5)]
        sess.tok.fill_(int(Z[f"{dt}/next_id"][0, 0]))
        for _ in range(4):
            logits.append(sess.decode_step(greedy=True).clone())
        outs[fused] = torch.stack([l.float() for l in logits])
    tol = {"f32": 1e-5, "f16": 5e-3}[dt]
    assert O.rel_l2(t2n(outs[True]), t2n(outs[False])) < tol
    e = O.rel_l2(t2n(outs[True][1]), Z[f"{dt}/decode_logits"].astype(np.float32)[0])
    print(f"[tiny parity] {dt}: fused step decode {e:.3e}")
    assert e < TINY_TOL[dt]


def test_fused_ops_unit_shapes():
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd import model as M
    torch.manual_seed(3)
    for dtype in (torch.float16, torch.bfloat16, torch.float32):
        x = torch.randn(3, 4096, device=DEV).to(dtype)
        w = (1 + 0.1 * torch.randn(4096, device=DEV)).to(dtype)
        norm = M.RMSNorm(4096, 1e-5, dtype).to(DEV)
        norm.weight.data.copy_(w)
        assert torch.equal(F_.rmsnorm(x, w, 1e-5), norm(x)) or O.rel_l2(t2n(F_.rmsnorm(x, w, 1e-5)), t2n(norm(x))) < 2e-3
        y = torch.randn(2, 27392, device=DEV).to(dtype)
        h, g = torch.split(y, 13696, dim=-1)
        want = torch.nn.functional.silu(h) * g
        assert O.rel_l2(t2n(F_.silu_mul(y, 13696)), t2n(want)) < 2e-3
    # attention at full ChatGLM2 head geometry: 32 heads, 2 groups, d 128, capacity 192, 70 valid positions
    B, H, Gq, D, cap, n = 2, 32, 2, 128, 192, 70
    q = torch.randn(B, 1, H * D, device=DEV).half()
    kc = torch.randn(B, cap, Gq, D, device=DEV).half()
    vc = torch.randn(B, cap, Gq, D, device=DEV).half()
    mask = torch.full((B, 1, cap), -1e10, device=DEV)
    mask[:, :, :n] = 0
    attn = M.ChatGLM2Attention(H * D, H, D, Gq, 0)
    want = attn.core(torch.float16, q.view(B, 1, Gq, H // Gq, D), kc, vc, mask)
    got = F_.decode_attention(q, kc, vc, mask, H, Gq, D)
    assert O.rel_l2(t2n(got), t2n(want)) < 2e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fused_prologues_equal_separate_ops(dtype):
    """qlinear_w4g32_fwd_packed_fused (add + RMSNorm / SiLU * gate inside the GEMV's activation staging) against
    the separate launches, at ChatGLM2-6B layer shapes."""
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd.int4 import hip_ops as H4
    from chatglm_q_amd.int4.qlinear import DynamicQuantizeLinear
    g = torch.Generator(device=DEV).manual_seed(21)
    for K, N, bias in ((4096, 4608, True), (4096, 27392, False)):
        layer = DynamicQuantizeLinear(K, N, bias=bias, dtype=dtype, device=DEV)
        layer.weight.copy_(torch.randint(0, 256, layer.weight.shape, dtype=torch.uint8, device=DEV, generator=g))
        layer.weight_scale.copy_((torch.rand(layer.weight_scale.shape, device=DEV, generator=g) * 0.01 + 0.002).to(dtype))
        if bias:
            layer.bias.copy_((torch.randn(N, device=DEV, generator=g) * 0.1).to(dtype))
        layer.prepare()
        h = torch.randn(1, 1, K, device=DEV, generator=g).to(dtype)
        d = torch.randn(1, 1, K, device=DEV, generator=g).to(dtype)
        w = (1 + 0.1 * torch.randn(K, device=DEV, generator=g)).to(dtype)
        for delta in (d, None):
            hout = torch.empty_like(h)
            got = H4.w4_forward_fused(_lib.PRO_ADDNORM, h, layer._packed, N, layer.bias, delta, w, hout, 1e-5)
            if delta is None:
                want_h, x = h, F_.rmsnorm(h, w, 1e-5)
            else:
                want_h, x = F_.add_rmsnorm(h, delta, w, 1e-5)
            with torch.no_grad():
                want = layer(x)
            assert torch.equal(hout, want_h)
            assert O.rel_l2(t2n(got), t2n(want)) < 2e-3
    layer = DynamicQuantizeLinear(13696, 4096, bias=False, dtype=dtype, device=DEV)
    layer.weight.copy_(torch.randint(0, 256, layer.weight.shape, dtype=torch.uint8, device=DEV, generator=g))
    layer.weight_scale.copy_((torch.rand(layer.weight_scale.shape, device=DEV, generator=g) * 0.01 + 0.002).to(dtype))
    layer.prepare()
    y = torch.randn(1, 1, 2 * 13696, device=DEV, generator=g).to(dtype)
    got = H4.w4_forward_fused(_lib.PRO_SILU, y, layer._packed, 4096)
    with torch.no_grad():
        want = layer(F_.silu_mul(y, 13696))
    assert O.rel_l2(t2n(got), t2n(want)) < 2e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bias", [False, True])
def test_gate_epilogue_equals_separate_ops(dtype, bias):
    """QL_EPI_SILU_GATE: add + RMSNorm prologue, w_in on the gate-interleaved layout, SiLU * gate epilogue - against
    the separate launches (w_in on the plain layout, then silu_mul).  The kernels add the same products in the
    same order, so the two paths agree exactly."""
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd.int4 import hip_ops as H4
    from chatglm_q_amd.int4.qlinear import DynamicQuantizeLinear
    g = torch.Generator(device=DEV).manual_seed(33)
    for K, hidden in ((4096, 13696), (256, 96)):
        layer = DynamicQuantizeLinear(K, 2 * hidden, bias=bias, dtype=dtype, device=DEV)
        layer.weight.copy_(torch.randint(0, 256, layer.weight.shape, dtype=torch.uint8, device=DEV, generator=g))
        layer.weight_scale.copy_((torch.rand(layer.weight_scale.shape, device=DEV, generator=g) * 0.01 + 0.002).to(dtype))
        if bias:
            layer.bias.copy_((torch.randn(2 * hidden, device=DEV, generator=g) * 0.1).to(dtype))
        gp, gb = layer.gated_packed(hidden)
        assert (gb is None) == (not bias)
        h = torch.randn(1, 1, K, device=DEV, generator=g).to(dtype)
        d = torch.randn(1, 1, K, device=DEV, generator=g).to(dtype)
        w = (1 + 0.1 * torch.randn(K, device=DEV, generator=g)).to(dtype)
        hout = torch.empty_like(h)
        got = H4.w4_forward_fused(_lib.PRO_ADDNORM | _lib.EPI_SILU_GATE, h, gp, 2 * hidden, gb, d, w, hout, 1e-5)
        want_h, x = F_.add_rmsnorm(h, d, w, 1e-5)
        with torch.no_grad():
            want = F_.silu_mul(layer(x), hidden)
        assert got.shape == (1, 1, hidden)
        assert torch.equal(hout, want_h)
        assert O.rel_l2(t2n(got), t2n(want)) < 2e-3
        # cached, and refreshed when the canonical buffers change
        assert layer.gated_packed(hidden)[0] is gp
        layer.weight.add_(1)
        assert layer.gated_packed(hidden)[0] is not gp


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("geom", [(2, 32, 2, 128, 192, 70), (3, 8, 4, 64, 64, 0), (1, 4, 1, 32, 40, 39),
                                  (1, 32, 2, 128, 640, 300), (2, 8, 2, 128, 320, 319), (1, 32, 2, 128, 64, 0),
                                  (1, 32, 2, 128, 2048, 1500), (2, 4, 2, 64, 1000, 999)])
def test_rope_attention_single_launch_equals_two(dtype, geom):
    """decode_attention_rope against rope_kv_write followed by decode_attention: same cache rows bit for bit,
    same attention output up to the position of the new value in the PV sum."""
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd import model as M
    B, H, Gq, D, cap, n = geom                               # n = positions already cached; the step writes row n
    g = torch.Generator(device=DEV).manual_seed(5)
    qkv = torch.randn(B, 1, (H + 2 * Gq) * D, device=DEV, generator=g).to(dtype)
    table = M.rotary_table(D, cap + 8).to(DEV).to(dtype).reshape(cap + 8, -1).contiguous()
    pos = torch.full((B, 1), n + 1, dtype=torch.long, device=DEV)
    widx = torch.tensor([n], dtype=torch.long, device=DEV)
    mask = torch.full((B, 1, cap), -1e10, device=DEV)
    mask[:, :, : n + 1] = 0
    caches = []
    for _ in range(2):
        g2 = torch.Generator(device=DEV).manual_seed(6)
        caches.append((torch.randn(B, cap, Gq, D, device=DEV, generator=g2).to(dtype),
                       torch.randn(B, cap, Gq, D, device=DEV, generator=g2).to(dtype)))
    (k1, v1), (k2, v2) = caches
    q = F_.rope_kv_write(qkv, table, pos, widx, k1, v1, H, Gq, D)
    want = F_.decode_attention(q, k1, v1, mask, H, Gq, D)
    got = F_.decode_attention_rope(qkv, table, pos, widx, k2, v2, mask, H, Gq, D, split=False)
    assert torch.equal(k1, k2) and torch.equal(v1, v2)
    # (bf16 with 16 heads per group: the group kernel rounds P at another scale than decode_attention)
    assert O.rel_l2(t2n(got), t2n(want)) < {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 8e-3}[dtype]
    # long-context form: 256-position windows + combine launch (probabilities not rounded before P.V)
    g3 = torch.Generator(device=DEV).manual_seed(6)
    k3 = torch.randn(B, cap, Gq, D, device=DEV, generator=g3).to(dtype)
    v3 = torch.randn(B, cap, Gq, D, device=DEV, generator=g3).to(dtype)
    got_split = F_.decode_attention_rope(qkv, table, pos, widx, k3, v3, mask, H, Gq, D, split=True)
    assert torch.equal(k1, k3) and torch.equal(v1, v3)
    # (bf16: the unsplit path rounds every probability to 8 mantissa bits, the split one does not)
    assert O.rel_l2(t2n(got_split), t2n(want)) < {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 8e-3}[dtype]


@pytest.mark.usefixtures("exact_dequant_policy")
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("M", [2, 3, 4])
def test_rows4_gate_epilogue_equals_separate_ops(M, bias, dtype):
    """SiLU * gate in the epilogue of the 4x4x4-MFMA kernel (2..4 rows, part 1 of the gate-interleaved copy,
    qlinear_w4g32_fwd_packed_gated) against the same kernel without the epilogue followed by silu_mul: same sums, same
    rounding sequence - bit for bit."""
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd.int4 import hip_ops as H4
    from chatglm_q_amd.int4.qlinear import DynamicQuantizeLinear
    g = torch.Generator(device=DEV).manual_seed(70 + M)
    for K, hidden in ((4096, 13696), (256, 96)):
        layer = DynamicQuantizeLinear(K, 2 * hidden, bias=bias, dtype=dtype, device=DEV)
        layer.weight.copy_(torch.randint(0, 256, layer.weight.shape, dtype=torch.uint8, device=DEV, generator=g))
        layer.weight_scale.copy_((torch.rand(layer.weight_scale.shape, device=DEV, generator=g) * 0.01 + 0.002).to(dtype))
        if bias:
            layer.bias.copy_((torch.randn(2 * hidden, device=DEV, generator=g) * 0.1).to(dtype))
        x = torch.randn(M, 1, K, device=DEV, generator=g).to(dtype)
        assert not H4.rows_on_tiled(M, 2 * hidden, K, dtype)
        gp, gb = layer.gated_packed(hidden)
        got = H4.w4_forward_gated(x, gp, 2 * hidden, gb, part1=True)
        with torch.no_grad():
            want = F_.silu_mul(layer(x), hidden)               # the module serves these row counts with the same kernel
        assert got is not None and got.shape == (M, 1, hidden)
        assert torch.equal(got, want)
        assert layer._gated_tiled is None and layer._tiled is None   # part 2 of neither copy was needed


@pytest.mark.parametrize("geom", [(2, 32, 2, 128, 64), (1, 4, 1, 32, 40)])
def test_rotary_entry_points_stay_inside_table_and_cache(geom):
    """No table length crosses the ABI: positions past the cache are clamped to `capacity` (the last row the table must
    hold), a write index outside the cache writes nothing, and a table that does not cover the cache is refused."""
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd import model as M
    B, H, Gq, D, cap = geom
    g = torch.Generator(device=DEV).manual_seed(9)
    qkv = torch.randn(B, 1, (H + 2 * Gq) * D, device=DEV, generator=g).half()
    table = M.rotary_table(D, cap + 1).to(DEV).half().reshape(cap + 1, -1).contiguous()
    k0 = torch.randn(B, cap, Gq, D, device=DEV, generator=g).half()
    v0 = torch.randn(B, cap, Gq, D, device=DEV, generator=g).half()
    mask = torch.zeros(B, 1, cap, device=DEV)
    far = torch.full((B, 1), 10 ** 9, dtype=torch.long, device=DEV)
    last = torch.full((B, 1), cap, dtype=torch.long, device=DEV)
    for widx in (cap, cap + 1000, -3):
        w = torch.tensor([widx], dtype=torch.long, device=DEV)
        k, v = k0.clone(), v0.clone()
        q_far = F_.rope_kv_write(qkv, table, far, w, k, v, H, Gq, D)
        assert torch.equal(k, k0) and torch.equal(v, v0)                  # nothing written outside the cache
        k2, v2 = k0.clone(), v0.clone()
        q_last = F_.rope_kv_write(qkv, table, last, torch.tensor([cap - 1], dtype=torch.long, device=DEV), k2, v2, H, Gq, D)
        assert torch.equal(q_far, q_last)                                # position clamped to `capacity`
        k3, v3 = k0.clone(), v0.clone()
        out = F_.decode_attention_rope(qkv, table, far, w, k3, v3, mask, H, Gq, D)
        assert torch.equal(k3, k0) and torch.equal(v3, v0) and bool(torch.isfinite(out).all())
    with pytest.raises(ValueError):
        F_.rope_kv_write(qkv, table[:cap], last, torch.tensor([0], dtype=torch.long, device=DEV), k0.clone(), v0.clone(), H, Gq, D)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("geom", [(1, 32, 2, 128, 256, 130), (2, 32, 2, 128, 192, 70), (1, 16, 1, 128, 40, 39),
                                  (1, 32, 2, 128, 2048, 1500), (2, 16, 1, 128, 320, 319), (1, 32, 2, 128, 8064, 7000),
                                  (3, 16, 1, 128, 1000, 256), (1, 32, 2, 128, 520, 511), (1, 32, 2, 128, 64, 0),
                                  (2, 48, 3, 128, 516, 300), (1, 32, 2, 128, 37, 36), (1, 16, 1, 128, 257, 256),
                                  (2, 32, 2, 128, 300, 150), (1, 32, 2, 128, 255, 0), (1, 16, 1, 128, 1, 0)])
def test_group_attention_on_matrix_cores_equals_per_head(dtype, geom, monkeypatch):
    """16 heads per key/value group: the group kernel (one block per group and 256-position window, Q.K and P.V as
    16x16 MFMA tiles, values transposed by ds_read_b64_tr_b16) against the per-head kernels: same
    cache rows bit for bit, outputs within the P rounding."""
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd import model as M
    B, H, Gq, D, cap, n = geom
    g = torch.Generator(device=DEV).manual_seed(11)
    qkv = torch.randn(B, 1, (H + 2 * Gq) * D, device=DEV, generator=g).to(dtype)
    table = M.rotary_table(D, cap + 8).to(DEV).to(dtype).reshape(cap + 8, -1).contiguous()
    pos = torch.full((B, 1), n + 1, dtype=torch.long, device=DEV)
    pos[-1] = n // 2 + 1                                     # sequences need not share a position
    widx = torch.tensor([n], dtype=torch.long, device=DEV)
    mask = torch.full((B, 1, cap), -1e10, device=DEV)
    mask[:, :, : n + 1] = 0
    if n > 4:
        mask[:, :, 3] = -1e10                                # a hole, as left padding makes
    outs = []
    from chatglm_q_amd import _lib
    for flag in ("nogroupattn", ""):
        monkeypatch.setenv("QLINEAR_DISPATCH", flag)
        _lib.get_lib().qlinear_dispatch_reload()
        g2 = torch.Generator(device=DEV).manual_seed(12)
        k = torch.randn(B, cap, Gq, D, device=DEV, generator=g2).to(dtype)
        v = torch.randn(B, cap, Gq, D, device=DEV, generator=g2).to(dtype)
        outs.append((F_.decode_attention_rope(qkv, table, pos, widx, k, v, mask, H, Gq, D, split=cap > 256), k, v))
        torch.cuda.synchronize()
    monkeypatch.delenv("QLINEAR_DISPATCH")
    _lib.get_lib().qlinear_dispatch_reload()
    (o0, k0, v0), (o1, k1, v1) = outs
    assert torch.isfinite(o1.float()).all()
    assert torch.equal(k0, k1) and torch.equal(v0, v1)
    tol = 2e-3 if dtype == torch.float16 else 1.2e-2        # P rounded to T at a different scale (see decode_ops.hip)
    assert O.rel_l2(t2n(o1), t2n(o0)) < tol
    # transposes / head mix-ups would pass a norm test on symmetric data: compare head by head
    a, r = o1.float().reshape(B, H, D), o0.float().reshape(B, H, D)
    assert ((a - r).norm(dim=-1) <= 4 * tol * r.norm(dim=-1) + 1e-3).all()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("M", [4096, 2048 + 200])
def test_gemm256_gate_epilogue_equals_separate_ops(dtype, bias, M):
    """Prefill row counts: qlinear_w4g32_fwd_tiled_gated on the 256 x 256-tile GEMM (SiLU * gate in its epilogue, output
    (M, hidden)) against the same GEMM followed by silu_mul - same sums, same rounding sequence, bit for bit
    (chatglm_q/model.py:199-201)."""
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd.int4 import hip_ops as H4
    from chatglm_q_amd.int4.qlinear import DynamicQuantizeLinear
    g = torch.Generator(device=DEV).manual_seed(11)
    K, hidden = 4096, 13696
    layer = DynamicQuantizeLinear(K, 2 * hidden, bias=bias, dtype=dtype, device=DEV)
    layer.weight.copy_(torch.randint(0, 256, layer.weight.shape, dtype=torch.uint8, device=DEV, generator=g))
    layer.weight_scale.copy_((torch.rand(layer.weight_scale.shape, device=DEV, generator=g) * 0.01 + 0.002).to(dtype))
    if bias:
        layer.bias.copy_((torch.randn(2 * hidden, device=DEV, generator=g) * 0.1).to(dtype))
    x = torch.randn(1, M, K, device=DEV, generator=g).to(dtype)
    gp, gb = layer.gated_tiled(hidden)
    got = H4.w4_forward_gated(x, gp, 2 * hidden, gb)
    with torch.no_grad():
        want = F_.silu_mul(layer(x), hidden)                 # the module takes the same 256 x 256-tile kernel at these row counts
    assert got is not None and got.shape == (1, M, hidden)
    assert torch.equal(got, want)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("shape", [(4096, 4096, 4096), (8192 - 56, 13696, 4096)])
def test_gemm256_residual_epilogue_equals_separate_ops(dtype, bias, shape):
    """Prefill row counts: qlinear_w4g32_fwd_tiled_residual (hidden + sublayer(...) with the add in the 256 x 256-tile GEMM's
    epilogue, chatglm_q/model.py:243,245) against the same GEMM followed by the elementwise add - bit for bit; row counts the
    kernel does not serve report None."""
    from chatglm_q_amd.int4 import hip_ops as H4
    from chatglm_q_amd.int4.qlinear import DynamicQuantizeLinear
    M, K, N = shape
    g = torch.Generator(device=DEV).manual_seed(13)
    layer = DynamicQuantizeLinear(K, N, bias=bias, dtype=dtype, device=DEV)
    layer.weight.copy_(torch.randint(0, 256, layer.weight.shape, dtype=torch.uint8, device=DEV, generator=g))
    layer.weight_scale.copy_((torch.rand(layer.weight_scale.shape, device=DEV, generator=g) * 0.01 + 0.002).to(dtype))
    if bias:
        layer.bias.copy_((torch.randn(N, device=DEV, generator=g) * 0.1).to(dtype))
    x = torch.randn(1, M, K, device=DEV, generator=g).to(dtype)
    h = torch.randn(1, M, N, device=DEV, generator=g).to(dtype)
    got = H4.w4_forward_tiled_residual(x, layer.tiled(), N, layer.bias, h)
    with torch.no_grad():
        want = h + layer(x)
    assert got is not None and got.shape == (1, M, N)
    assert torch.equal(got, want)
    assert H4.w4_forward_tiled_residual(x[:, :40], layer.tiled(), N, layer.bias, h[:, :40]) is None


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bias", [False, True])
def test_int8_gemm256_gate_and_residual_epilogues_equal_separate_ops(dtype, bias):
    """int8 weight-only at prefill row counts: qlinear_w8_fwd_tiled_gated (SiLU * gate on the gate-interleaved copy) and
    qlinear_w8_fwd_tiled_residual against the module's own GEMM followed by silu_mul / the elementwise add - bit for bit
    (chatglm_q/model.py:199-201,243-245 on chatglm_q/int8/qlinear.py:90-93)."""
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd.int8 import hip_ops as H8
    from chatglm_q_amd.int8.qlinear import DynamicQuantizeLinear as Q8
    g = torch.Generator(device=DEV).manual_seed(17)
    K, hidden, M = 4096, 13696, 4096
    layer = Q8(K, 2 * hidden, bias=bias, dtype=dtype, device=DEV)
    layer.weight.copy_(torch.randint(-128, 128, layer.weight.shape, dtype=torch.int8, device=DEV, generator=g))
    layer.weight_scale.copy_((torch.rand(2 * hidden, device=DEV, generator=g) * 0.001 + 0.0002).to(dtype))
    if bias:
        layer.bias.copy_((torch.randn(2 * hidden, device=DEV, generator=g) * 0.1).to(dtype))
    x = torch.randn(1, M, K, device=DEV, generator=g).to(dtype)
    tiled, s_perm, b_perm = layer.gated_tiled(hidden)
    got = H8.w8_forward_tiled_gated(x, tiled, 2 * hidden, s_perm, b_perm)
    with torch.no_grad():
        want = F_.silu_mul(layer(x), hidden)
    assert got is not None and got.shape == (1, M, hidden)
    assert torch.equal(got, want)
    out = Q8(hidden, K, bias=bias, dtype=dtype, device=DEV)
    out.weight.copy_(torch.randint(-128, 128, out.weight.shape, dtype=torch.int8, device=DEV, generator=g))
    out.weight_scale.copy_((torch.rand(K, device=DEV, generator=g) * 0.001 + 0.0002).to(dtype))
    if bias:
        out.bias.copy_((torch.randn(K, device=DEV, generator=g) * 0.1).to(dtype))
    M2 = 8192 - 24
    y = torch.randn(1, M2, hidden, device=DEV, generator=g).to(dtype)
    h = torch.randn(1, M2, K, device=DEV, generator=g).to(dtype)
    got = H8.w8_forward_tiled_residual(y, out.prepare()._tiled, K, out.weight_scale, out.bias, h)
    with torch.no_grad():
        want = h + out(y)
    assert got is not None and torch.equal(got, want)
    assert H8.w8_forward_tiled_residual(y[:, :40], out.prepare()._tiled, K, out.weight_scale, out.bias, h[:, :40]) is None


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("M", [3, 8, 32])
def test_fewrow_gate_epilogue_equals_separate_ops(dtype, bias, M):
    """qlinear_w4g32_fwd_packed_gated (few rows, SiLU * gate in the MFMA kernel's epilogue) against the projection
    followed by silu_mul: same sums, same rounding sequence - bit for bit; unsupported shapes report None."""
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd.int4 import hip_ops as H4
    from chatglm_q_amd.int4.qlinear import DynamicQuantizeLinear
    g = torch.Generator(device=DEV).manual_seed(7 + M)
    K, hidden = 4096, 13696
    layer = DynamicQuantizeLinear(K, 2 * hidden, bias=bias, dtype=dtype, device=DEV)
    layer.weight.copy_(torch.randint(0, 256, layer.weight.shape, dtype=torch.uint8, device=DEV, generator=g))
    layer.weight_scale.copy_((torch.rand(layer.weight_scale.shape, device=DEV, generator=g) * 0.01 + 0.002).to(dtype))
    if bias:
        layer.bias.copy_((torch.randn(2 * hidden, device=DEV, generator=g) * 0.1).to(dtype))
    x = torch.randn(M, 1, K, device=DEV, generator=g).to(dtype)
    assert layer._gated_tiled is None                        # part 2 of the gated copy: built on first use only
    gp, gb = layer.gated_tiled(hidden)
    assert layer.gated_tiled(hidden)[0] is gp
    got = H4.w4_forward_gated(x, gp, 2 * hidden, gb)
    # the same few-row kernel without the epilogue (the module itself may serve 3 / 4 rows through the 4x4x4-MFMA kernel,
    # whose default arithmetic rounds differently: within tolerance of this, not bit-equal)
    y = H4.w4_forward(x, layer.weight, layer.weight_scale, layer.bias, None, tiled=layer.tiled())
    want = F_.silu_mul(y, hidden)
    assert got is not None and got.shape == (M, 1, hidden)
    assert torch.equal(got, want)
    with torch.no_grad():
        assert O.rel_l2(t2n(F_.silu_mul(layer(x), hidden)), t2n(want)) < (2e-3 if dtype == torch.float16 else 8e-3)
    # narrow matrices take K slabs + a reduce launch: not served
    small = DynamicQuantizeLinear(4096, 512, bias=False, dtype=dtype, device=DEV)
    small.weight.copy_(torch.randint(0, 256, small.weight.shape, dtype=torch.uint8, device=DEV, generator=g))
    small.weight_scale.fill_(0.01)
    sp, _ = small.gated_tiled(256)
    assert H4.w4_forward_gated(x, sp, 512, None) is None


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("nk", [(4096, 4096), (4096, 13696), (250, 96), (8, 64)])
def test_residual_epilogue_and_norm_prologue_equal_separate_ops(dtype, bias, nk):
    """qlinear_w4g32_fwd_packed_residual = layer(x) + h (three rounded operations, chatglm_q/model.py:243,245), and the
    RMSNorm prologue without delta = layer(rmsnorm(h)): bit for bit, also for ragged column quads."""
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd.int4 import hip_ops as H4
    from chatglm_q_amd.int4.qlinear import DynamicQuantizeLinear
    N, K = nk
    g = torch.Generator(device=DEV).manual_seed(N + K)
    layer = DynamicQuantizeLinear(K, N, bias=bias, dtype=dtype, device=DEV)
    layer.weight.copy_(torch.randint(0, 256, layer.weight.shape, dtype=torch.uint8, device=DEV, generator=g))
    layer.weight_scale.copy_((torch.rand(layer.weight_scale.shape, device=DEV, generator=g) * 0.01 + 0.002).to(dtype))
    if bias:
        layer.bias.copy_((torch.randn(N, device=DEV, generator=g) * 0.1).to(dtype))
    x = torch.randn(1, 1, K, device=DEV, generator=g).to(dtype)
    h = torch.randn(1, 1, N, device=DEV, generator=g).to(dtype)
    packed = layer.prepare()._packed
    with torch.no_grad():
        want = layer(x) + h
    got = H4.w4_forward_residual(x, packed, N, layer.bias, h)
    assert torch.equal(got, want)
    got_alias = h.clone()
    got_alias = H4.w4_forward_residual(x, packed, N, layer.bias, got_alias)
    assert torch.equal(got_alias, want)
    w = (1 + 0.1 * torch.randn(K, device=DEV, generator=g)).to(dtype)
    got_n = H4.w4_forward_fused(_lib.PRO_ADDNORM, x, packed, N, layer.bias, None, w, None, 1e-5)
    with torch.no_grad():
        want_n = layer(F_.rmsnorm(x, w, 1e-5))
    assert torch.equal(got_n, want_n)


@pytest.mark.parametrize("kind", ["int4", "int8"])
@pytest.mark.parametrize("nk", [(4096, 4096), (1000, 512), (36, 64)])
def test_attention_prefetch_workgroups_change_nothing(kind, nk):
    """The spare workgroups that warm the next linear's weights only read: outputs and caches are bit-identical with
    and without them, also for weight buffers smaller than one prefetch unit."""
    from chatglm_q_amd import _lib, fused_ops as F_
    from chatglm_q_amd import model as M
    N, K = nk
    B, H, Gq, D, cap, n = 2, 32, 2, 128, 192, 70
    g = torch.Generator(device=DEV).manual_seed(3)
    if kind == "int4":
        from chatglm_q_amd.int4.qlinear import DynamicQuantizeLinear
        layer = DynamicQuantizeLinear(K, N, bias=False, dtype=torch.float16, device=DEV)
        layer.weight.copy_(torch.randint(0, 256, layer.weight.shape, dtype=torch.uint8, device=DEV, generator=g))
        layer.weight_scale.copy_((torch.rand(layer.weight_scale.shape, device=DEV, generator=g) * 0.01 + 0.002).half())
        nxt = (layer.prepare()._packed, _lib.NEXT_W4G32_PACKED, N, K)
    else:
        w = torch.randint(-127, 128, (N, K), device=DEV, generator=g, dtype=torch.int8)
        nxt = (w, _lib.NEXT_W8_ROWS, N, K)
    qkv = torch.randn(B, 1, (H + 2 * Gq) * D, device=DEV, generator=g).half()
    table = M.rotary_table(D, cap + 8).to(DEV).half().reshape(cap + 8, -1).contiguous()
    pos = torch.full((B, 1), n + 1, dtype=torch.long, device=DEV)
    widx = torch.tensor([n], dtype=torch.long, device=DEV)
    mask = torch.full((B, 1, cap), -1e10, device=DEV)
    mask[:, :, : n + 1] = 0
    outs = []
    for pf in (None, nxt):
        g2 = torch.Generator(device=DEV).manual_seed(4)
        k = torch.randn(B, cap, Gq, D, device=DEV, generator=g2).half()
        v = torch.randn(B, cap, Gq, D, device=DEV, generator=g2).half()
        outs.append((F_.decode_attention_rope(qkv, table, pos, widx, k, v, mask, H, Gq, D, prefetch=pf), k, v))
    torch.cuda.synchronize()
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("N", [65024, 1003, 8])
def test_greedy_advance_argmax_and_bookkeeping(dtype, N):
    """tok = argmax (lowest index on ties, as torch.argmax), pos += 1, write_index += 1, mask[new index] = 0."""
    from chatglm_q_amd import fused_ops as F_
    g = torch.Generator(device=DEV).manual_seed(N)
    B, cap = 3, 16
    logits = torch.randn(B, N + 5, device=DEV, generator=g).to(dtype)[:, :N]      # row stride N + 5: rows 1, 2 unaligned
    top = logits.max(dim=-1).values
    for b in range(B):                                                              # plant ties with the maximum
        idx = torch.randint(0, N, (3,), device=DEV, generator=g)
        logits[b, idx] = top[b]
    want = logits.float().argmax(dim=-1)
    tok = torch.zeros(B, 1, dtype=torch.long, device=DEV)
    widx = torch.tensor([4], dtype=torch.long, device=DEV)
    pos = torch.full((B, 1), 5, dtype=torch.long, device=DEV)
    mask = torch.full((B, 1, cap), -1e10, device=DEV)
    F_.greedy_advance(logits, tok, widx, pos, mask)
    assert torch.equal(tok[:, 0], want)
    assert int(widx[0]) == 5 and torch.equal(pos, torch.full_like(pos, 6))
    assert torch.all(mask[:, 0, 5] == 0) and torch.all(mask[:, 0, :5] == -1e10) and torch.all(mask[:, 0, 6:] == -1e10)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 16, 40, 70), (1, 3, 5, 2100), (2, 2, 64, 2048), (1, 1, 7, 3000)])
def test_masked_softmax_equals_torch_sequence(dtype, shape):
    """qlinear_masked_softmax against the reference op sequence: fp32 (scores + mask) -> softmax -> cast."""
    from chatglm_q_amd import fused_ops as F_
    G_, Hg, S, T = shape
    g = torch.Generator(device=DEV).manual_seed(S * T)
    scores = (torch.randn(G_, Hg, S, T, device=DEV, generator=g) * 3).to(dtype)
    full = torch.zeros(S, T + 9, device=DEV)
    full[:, T // 2:] = torch.where(torch.rand(S, T + 9 - T // 2, device=DEV, generator=g) < 0.5, -1e10, 0.0)
    full[:, 0] = 0.0                                            # every row keeps at least one visible key
    mask = full[:, :T]                                          # row stride T + 9, like a sliced cache mask
    want = torch.softmax(scores.float() + mask[None, None], dim=-1).to(dtype)
    got = F_.masked_softmax(scores, mask)
    assert got.shape == scores.shape
    assert torch.allclose(got.float(), want.float(), atol=2e-3 if dtype == torch.float16 else 1e-2, rtol=0)
    assert O.rel_l2(t2n(got), t2n(want)) < (1e-3 if dtype == torch.float16 else 4e-3)
    none = F_.masked_softmax(scores, None)
    assert O.rel_l2(t2n(none), t2n(torch.softmax(scores.float(), -1).to(dtype))) < (1e-3 if dtype == torch.float16 else 4e-3)


@pytest.mark.parametrize("bias", [False, True])
def test_int8_fused_prologue_and_gate_epilogue(bias):
    """qlinear_w8_fwd_fused against the separate launches (add + RMSNorm, int8 GEMV, SiLU * gate)."""
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd.int8 import hip_ops as H8
    from chatglm_q_amd.int8.qlinear import DynamicQuantizeLinear as L8
    g = torch.Generator(device=DEV).manual_seed(44)
    for K, N in ((4096, 4608), (4096, 27392), (256, 192)):
        layer = L8(K, N, bias=bias, dtype=torch.float16, device=DEV)
        layer.weight.copy_(torch.randint(-127, 128, (N, K), dtype=torch.int8, device=DEV, generator=g))
        layer.weight_scale.copy_((torch.rand(N, device=DEV, generator=g) * 0.004 + 0.001).half())
        if bias:
            layer.bias.copy_((torch.randn(N, device=DEV, generator=g) * 0.1).half())
        h = torch.randn(1, 1, K, device=DEV, generator=g).half()
        d = torch.randn(1, 1, K, device=DEV, generator=g).half()
        w = (1 + 0.1 * torch.randn(K, device=DEV, generator=g)).half()
        for delta in (d, None):
            hout = torch.empty_like(h)
            got = H8.w8_forward_fused(_lib.PRO_ADDNORM, h, layer.weight, layer.weight_scale, layer.bias, delta, w, hout, 1e-5)
            want_h, x = (h, F_.rmsnorm(h, w, 1e-5)) if delta is None else F_.add_rmsnorm(h, delta, w, 1e-5)
            with torch.no_grad():
                want = layer(x)
            assert torch.equal(hout, want_h)
            assert O.rel_l2(t2n(got), t2n(want)) < 2e-3
        # no delta, nothing to write back (the PRO_NORM kernels): bit-equal to the delta=None call above
        got_n = H8.w8_forward_fused(_lib.PRO_ADDNORM, h, layer.weight, layer.weight_scale, layer.bias, None, w, None, 1e-5)
        assert torch.equal(got_n, got)
        # residual epilogue: layer(x) + r as three rounded operations, bit for bit
        r = torch.randn(1, 1, N, device=DEV, generator=g).half()
        with torch.no_grad():
            want_r = layer(h) + r
        assert torch.equal(H8.w8_forward_residual(h, layer.weight, layer.weight_scale, layer.bias, r), want_r)
        hidden = N // 2
        gw, gs, gb = layer.gated(hidden)
        hout = torch.empty_like(h)
        got = H8.w8_forward_fused(_lib.PRO_ADDNORM | _lib.EPI_SILU_GATE, h, gw, gs, gb, d, w, hout, 1e-5)
        want_h, x = F_.add_rmsnorm(h, d, w, 1e-5)
        with torch.no_grad():
            want = F_.silu_mul(layer(x), hidden)
        assert got.shape == (1, 1, hidden) and torch.equal(hout, want_h)
        assert O.rel_l2(t2n(got), t2n(want)) < 2e-3
        assert layer.gated(hidden)[0] is gw


def test_int8_model_one_row_step_matches_unfused_graph():
    """The 5-launch decode step on an int8 model against the same model with the fused ops switched off."""
    from chatglm_q_amd import model as M
    from chatglm_q_amd.decoder import DecodeSession
    cfg = M.ChatGLM2Config(hidden_size=256, inner_hidden_size=384, head_hidden_size=32, num_multi_query_groups=2,
                           num_attention_heads=8, num_layers=2, vocab_size=320, max_sequence_length=64)
    with torch.device(DEV):
        model = M.create_quant_int8_model(cfg, dtype=torch.float16)
    M.fill_synthetic_(model, 3)
    model.eval()
    assert model._one_row_kind(torch.float16) == "int8"
    ids = torch.randint(0, 320, (1, 9), device=DEV)
    outs = {}
    for fused in (True, False):
        M.FUSED_DECODE_OPS = fused
        try:
            sess = DecodeSession(model, 1, 32, use_graph=False)
            logits = [sess.prefill(ids)]
            sess.tok.copy_(logits[0].argmax(-1, keepdim=True))
            for _ in range(3):
                logits.append(sess.decode_step(greedy=True).clone())
            outs[fused] = torch.stack([l.float() for l in logits])
        finally:
            M.FUSED_DECODE_OPS = True
    assert O.rel_l2(t2n(outs[True]), t2n(outs[False])) < 5e-3


@pytest.mark.usefixtures("exact_dequant_policy")
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [2, 3, 4])
@pytest.mark.parametrize("with_delta", [False, True])
def test_rows_fused_prologue_equals_separate_ops(M, with_delta, dtype):
    """qlinear_w4g32_fwd_rows_fused (2..4 rows: residual add + RMSNorm in the staging of the 4x4x4-MFMA kernel, optional
    SiLU * gate epilogue) against add_rmsnorm / rmsnorm followed by the module (and silu_mul): bit for bit, hnew included."""
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd.int4 import hip_ops as H4
    from chatglm_q_amd.int4.qlinear import DynamicQuantizeLinear
    g = torch.Generator(device=DEV).manual_seed(300 + M)
    for K, N, hidden in ((4096, 4608, None), (4096, 2 * 13696, 13696), (256, 200, None), (1024, 512, 256)):
        layer = DynamicQuantizeLinear(K, N, bias=True, dtype=dtype, device=DEV)
        layer.weight.copy_(torch.randint(0, 256, layer.weight.shape, dtype=torch.uint8, device=DEV, generator=g))
        layer.weight_scale.copy_((torch.rand(layer.weight_scale.shape, device=DEV, generator=g) * 0.01 + 0.002).to(dtype))
        layer.bias.copy_((torch.randn(N, device=DEV, generator=g) * 0.1).to(dtype))
        h = torch.randn(M, 1, K, device=DEV, generator=g).to(dtype)
        d = torch.randn(M, 1, K, device=DEV, generator=g).to(dtype) if with_delta else None
        w = (1 + 0.1 * torch.randn(K, device=DEV, generator=g)).to(dtype)
        if with_delta:
            want_h, x = F_.add_rmsnorm(h, d, w, 1e-5)
        else:
            want_h, x = None, F_.rmsnorm(h, w, 1e-5)
        with torch.no_grad():
            want = layer(x)
        if hidden:
            want = F_.silu_mul(want, hidden)
            packed, bias = layer.gated_packed(hidden)
            kind = _lib.PRO_ADDNORM | _lib.EPI_SILU_GATE
        else:
            packed, bias, kind = layer.prepare()._packed, layer.bias, _lib.PRO_ADDNORM
        res = H4.w4_forward_rows_fused(kind, h, packed, N, bias, d, w, 1e-5)
        assert res is not None
        got, got_h = res
        assert torch.equal(got, want)
        assert (got_h is None) == (not with_delta)
        if with_delta:
            assert torch.equal(got_h, want_h)
    # 5 rows are not served this way
    h5 = torch.randn(5, 1, 256, device=DEV, generator=g).to(dtype)
    small = DynamicQuantizeLinear(256, 64, bias=False, dtype=dtype, device=DEV)
    small.weight.fill_(0x88)
    small.weight_scale.fill_(0.01)
    assert H4.w4_forward_rows_fused(_lib.PRO_ADDNORM, h5, small.prepare()._packed, 64, None, None, w[:256].contiguous(), 1e-5) is None


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("M", [4096, 2048 + 200])
def test_int8_activation_gate_epilogue_equals_separate_ops(dtype, bias, M):
    """int8 ACTIVATIONS at prefill row counts: ``forward_quantized_gated`` (qlinear_w8a8_fwd_tiled_gated: SiLU * gate in the int8 x int8
    ring GEMM's epilogue on the gate-interleaved copy, output (M, hidden)) against ``forward_quantized`` followed by silu_mul - same
    integer sums, same rounding sequence, bit for bit (chatglm_q/model.py:199-201 on chatglm_q/int8/qlinear.py:56-62); row counts the
    many-row kernel does not serve report None before any copy is built."""
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd.int8 import hip_ops as H8
    from chatglm_q_amd.int8.qlinear import DynamicQuantizeLinear as Q8
    g = torch.Generator(device=DEV).manual_seed(23)
    K, hidden = 4096, 13696
    layer = Q8(K, 2 * hidden, bias=bias, dtype=dtype, device=DEV)
    layer.weight.copy_(torch.randint(-128, 128, layer.weight.shape, dtype=torch.int8, device=DEV, generator=g))
    layer.weight_scale.copy_((torch.rand(2 * hidden, device=DEV, generator=g) * 0.001 + 0.0002).to(dtype))
    if bias:
        layer.bias.copy_((torch.randn(2 * hidden, device=DEV, generator=g) * 0.1).to(dtype))
    layer.act_quant = True
    x = torch.randn(M, K, device=DEV, generator=g).to(dtype)
    a_q, a_s = H8.act_quant_rowwise(x)
    assert layer.forward_quantized_gated(a_q, a_s, hidden) is not None          # builds the gate-interleaved copy
    before = _lib.launch_count()
    got = layer.forward_quantized_gated(a_q, a_s, hidden)
    assert got is not None and got.shape == (M, hidden) and _lib.launch_count() - before == 1
    with torch.no_grad():
        want = F_.silu_mul(layer.forward_quantized(a_q, a_s), hidden)
    assert torch.equal(got, want)
    small = Q8(K, 2 * hidden, bias=bias, dtype=dtype, device=DEV)
    small.act_quant = True
    assert small.forward_quantized_gated(a_q[:40], a_s[:40], hidden) is None and small._gated_tiled is None


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("geom", [(2, 32, 2, 128, 256, 100), (1, 32, 2, 128, 516, 300), (2, 4, 2, 32, 64, 20)])
def test_rows_behind_the_write_index_are_hidden_by_every_attention_kernel(dtype, geom, monkeypatch):
    """the 16-heads-per-group kernel hides every cache row behind widx[0] whatever the mask says; the per-head kernels (fp32,
    D != 128, QLINEAR_DISPATCH=nogroupattn) honoured the mask alone - two functions for one (mask, widx).  Now all of them hide those rows:
    a mask that (wrongly) opens rows behind the write index gives the result of the correct mask, on every path."""
    from chatglm_q_amd import _lib
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd import model as M
    B, H, Gq, D, cap, n = geom
    g = torch.Generator(device=DEV).manual_seed(21)
    qkv = torch.randn(B, 1, (H + 2 * Gq) * D, device=DEV, generator=g).to(dtype)
    table = M.rotary_table(D, cap + 8).to(DEV).to(dtype).reshape(cap + 8, -1).contiguous()
    pos = torch.full((B, 1), n + 1, dtype=torch.long, device=DEV)
    widx = torch.tensor([n], dtype=torch.long, device=DEV)
    good = torch.full((B, 1, cap), -1e10, device=DEV)
    good[:, :, : n + 1] = 0
    open_behind = torch.zeros((B, 1, cap), device=DEV)                       # nothing hidden: rows n + 1 .. cap - 1 hold garbage
    for flag in ("nogroupattn", ""):
        monkeypatch.setenv("QLINEAR_DISPATCH", flag)
        _lib.get_lib().qlinear_dispatch_reload()
        outs = []
        for mask in (good, open_behind):
            g2 = torch.Generator(device=DEV).manual_seed(22)
            k = (torch.randn(B, cap, Gq, D, device=DEV, generator=g2) * 4).to(dtype)
            v = (torch.randn(B, cap, Gq, D, device=DEV, generator=g2) * 4).to(dtype)
            outs.append(F_.decode_attention_rope(qkv, table, pos, widx, k, v, mask, H, Gq, D, split=cap > 256))
        torch.cuda.synchronize()
        assert torch.isfinite(outs[1].float()).all()
        assert torch.equal(outs[0], outs[1]), flag
    monkeypatch.delenv("QLINEAR_DISPATCH")
    _lib.get_lib().qlinear_dispatch_reload()
