"""Recorded experiments of libqlinear_hip_dev.so (include/qlinear_hip_dev.h, chatglm_q_amd/dev/experiments.py): the chained-grid MLP
pair (round 2), the persistent MLP engine (round 3) and W4A8 (rounds 2 - 3, SURVEY row A10 as added by the judge).  They lost
to the product path and left the product library in round 4; their parity tests stay - marker ``dev`` - and still run under
``-m gpu`` (build() compiles the developer library too)."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.dev]

from oracle import qlinear_oracle as O  # noqa: E402
from chatglm_q_amd import _lib  # noqa: E402
from chatglm_q_amd import model as M  # noqa: E402
from chatglm_q_amd.decoder import ChatGLMDecoder  # noqa: E402
from chatglm_q_amd.dev import experiments as X  # noqa: E402
from chatglm_q_amd.int4 import hip_ops as h4  # noqa: E402
from chatglm_q_amd.int4 import hip_ops as H4  # noqa: E402
from chatglm_q_amd.int4 import qlinear as q4  # noqa: E402
from test_host_fast_gpu import _tiny  # noqa: E402
from test_parity_gpu import TDT, _rand_w4, t2n  # noqa: E402

DEV = "cuda:0"
W4A8_SHAPES = [(512, 4096, 4096, "f16"), (70, 1024, 200, "f16"), (33, 512, 96, "bf16"), (1, 4096, 256, "f16"),
               (130, 13696, 136, "f16"), (2048, 1024, 512, "bf16"), (513, 4160, 264, "f16"), (40, 64, 40, "f16"),
               (96, 32, 32, "f16"), (1000, 1088, 1000, "bf16"), (300, 96, 130, "f16")]


def dev_launches():
    return int(_lib.get_dev_lib().qlinear_launch_count())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("strict", [False, True])
@pytest.mark.parametrize("shape,bias", [((4096, 13696), False), ((4096, 13696), True), ((256, 384), True), ((1024, 2752), False)])
def test_mlp_engine_one_persistent_launch_equals_two(shape, bias, strict, dtype):
    """qlinear_w4g32_mlp_engine (w4_engine.hip: LDS-DMA loader waves + consumer waves per CU, granule hand-off of the row between
    the projections) against the two fused launches it replaces - bit for bit, in both arithmetic modes, over repeated launches
    on one workspace (the launch epoch advances in device memory), real layer size and small sizes (K slices of the first
    projection, fewer quads than CUs); no bounded wait gave up."""
    from chatglm_q_amd.int4.qlinear import DynamicQuantizeLinear
    K, hidden = shape
    if not X.mlp_engine_supported(2 * hidden, K, K):
        pytest.skip("shape not served by the engine on this device")
    g = torch.Generator(device=DEV).manual_seed(K + hidden)
    w_in = DynamicQuantizeLinear(K, 2 * hidden, bias=bias, dtype=dtype, device=DEV)
    w_out = DynamicQuantizeLinear(hidden, K, bias=bias, dtype=dtype, device=DEV)
    for l in (w_in, w_out):
        l.weight.copy_(torch.randint(0, 256, l.weight.shape, dtype=torch.uint8, device=DEV, generator=g))
        l.weight_scale.copy_((torch.rand(l.weight_scale.shape, device=DEV, generator=g) * 0.01 + 0.002).to(dtype))
        if bias:
            l.bias.copy_((torch.randn(l.bias.shape, device=DEV, generator=g) * 0.1).to(dtype))
    ln = (1 + 0.1 * torch.randn(K, device=DEV, generator=g)).to(dtype)
    gp, gb = w_in.gated_packed(hidden)
    po = w_out.prepare()._packed
    ws = X.mlp_engine_workspace(2 * hidden, DEV)
    for rep in range(5):
        x = torch.randn(1, 1, K, device=DEV, generator=g).to(dtype)
        y = h4.w4_forward_fused(_lib.PRO_ADDNORM | _lib.EPI_SILU_GATE, x, gp, 2 * hidden, gb, None, ln, None, 1e-5, strict=strict)
        want = h4.w4_forward_residual(y, po, K, w_out.bias, x, strict=strict)
        before = dev_launches()
        got = X.w4_mlp_engine(x, ln, 1e-5, gp, gb, 2 * hidden, po, w_out.bias, K, ws, strict=strict)
        assert got is not None and dev_launches() == before + 1
        torch.cuda.synchronize()
        assert X.mlp_engine_error(ws) == 0
        assert torch.equal(got, want), (rep, int((got != want).sum()))


def test_decode_with_the_mlp_engine_equals_the_five_launch_step(monkeypatch):
    """enable_mlp_engine(): the decode step with the persistent MLP launch (4 launches per layer) produces the token stream of
    the 5-launch step, eager and from the HIP graph."""
    model, cfg = _tiny()
    if not X.mlp_engine_supported(2 * cfg.inner_hidden_size, cfg.hidden_size, cfg.hidden_size):
        pytest.skip("tiny MLP not served by the engine")
    prefix = [3, 17, 200, 5, 77]
    kw = dict(max_generated_tokens=10, greedy=True, ignore_eos=True)
    want = list(ChatGLMDecoder(None, model).generate_ids(prefix, use_graph=False, **kw))
    X.enable_mlp_engine()
    try:
        before = _lib.launch_count()
        eager = list(ChatGLMDecoder(None, model).generate_ids(prefix, use_graph=False, **kw))
        n_eager = _lib.launch_count() - before
        graph = list(ChatGLMDecoder(None, model).generate_ids(prefix, use_graph=True, **kw))
        assert eager == want and graph == want
    finally:
        X.disable_mlp_experiments()
    before = _lib.launch_count()
    list(ChatGLMDecoder(None, model).generate_ids(prefix, use_graph=False, **kw))
    assert n_eager < _lib.launch_count() - before               # one launch less per layer and step


@pytest.mark.usefixtures("exact_dequant_policy")
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bias", [False, True])
def test_mlp_pair_one_launch_equals_two(dtype, bias):
    """qlinear_w4g32_mlp_pair (experiment: both MLP projections of a one-row decode step in ONE launch, the second one's
    workgroups waiting inside the launch for the first one's row) against the two fused launches it replaces: bit for bit,
    over repeated launches (the arrival counters reset themselves), and no consumer gave up waiting."""
    from chatglm_q_amd.int4 import hip_ops as H4
    from chatglm_q_amd.int4.qlinear import DynamicQuantizeLinear
    g = torch.Generator(device=DEV).manual_seed(91)
    K, hidden = 4096, 13696
    w_in = DynamicQuantizeLinear(K, 2 * hidden, bias=bias, dtype=dtype, device=DEV)
    w_out = DynamicQuantizeLinear(hidden, K, bias=bias, dtype=dtype, device=DEV)
    for l in (w_in, w_out):
        l.weight.copy_(torch.randint(0, 256, l.weight.shape, dtype=torch.uint8, device=DEV, generator=g))
        l.weight_scale.copy_((torch.rand(l.weight_scale.shape, device=DEV, generator=g) * 0.01 + 0.002).to(dtype))
        if bias:
            l.bias.copy_((torch.randn(l.bias.shape, device=DEV, generator=g) * 0.1).to(dtype))
    ln = (1 + 0.1 * torch.randn(K, device=DEV, generator=g)).to(dtype)
    gp, gb = w_in.gated_packed(hidden)
    po = w_out.prepare()._packed
    for rep in range(4):
        h = torch.randn(1, 1, K, device=DEV, generator=g).to(dtype)
        y = H4.w4_forward_fused(_lib.PRO_ADDNORM | _lib.EPI_SILU_GATE, h, gp, 2 * hidden, gb, None, ln, None, 1e-5)
        want = H4.w4_forward_residual(y, po, K, w_out.bias, h)
        got = X.w4_mlp_pair(h, ln, 1e-5, gp, gb, 2 * hidden, po, w_out.bias, K, h)
        assert got is not None and torch.equal(got, want)
    torch.cuda.synchronize()
    assert not X.mlp_pair_timed_out(DEV)
    # other shapes are not served: the caller keeps its two launches
    small = DynamicQuantizeLinear(256, 512, bias=False, dtype=dtype, device=DEV)
    small.weight.fill_(0x88)
    small.weight_scale.fill_(0.01)
    s_out = DynamicQuantizeLinear(256, 256, bias=False, dtype=dtype, device=DEV)
    s_out.weight.fill_(0x88)
    s_out.weight_scale.fill_(0.01)
    x = torch.randn(1, 1, 256, device=DEV, generator=g).to(dtype)
    assert X.w4_mlp_pair(x, ln[:256].contiguous(), 1e-5, small.gated_packed(256)[0], None, 512, s_out.prepare()._packed, None, 256, x) is None


@pytest.mark.parametrize("M,K,N,dt", W4A8_SHAPES)
def test_w4a8_vs_oracle(M, K, N, dt):
    """Two-level parity as for W8A8: (i) the integer stage - quantize_int8 rows composed with the nibble
    decode, one exact int32 sum per 32-deep group - bit-exact (unit scales, operands bounded so the fp32 fold is exact);
    (ii) against the oracle's W4A8 formula to fp32 roundoff / one output rounding; the distance to the weight-only W4A16
    result is REPORTED (activation quantisation error), not claimed.  K = 13696 (428 groups), odd group counts (3, 33,
    65, 130), ragged M / N, every tile height."""
    qw, sc = _rand_w4(K, N, dt, seed=K * 7 + N)
    g = torch.Generator().manual_seed(M + 17)
    a = torch.randn((M, K), generator=g).to(TDT[dt])
    bias = (torch.randn(N, generator=g) * 0.1).to(TDT[dt])
    qd, sd = qw.to(DEV), sc.to(DEV)
    a8 = X.pack_w4a8(qd, sd)
    ref = O.w4a8_matmul(t2n(a), qw.numpy(), t2n(sc), t2n(bias), dtype=dt)
    before = dev_launches()
    out = X.w4a8_forward(a.to(DEV), a8, N, bias.to(DEV))
    assert dev_launches() - before == 2                                  # activation quantiser + GEMM
    err = O.rel_l2(t2n(out), ref)
    assert err <= {"f16": 3e-4, "bf16": 2e-3}[dt], err
    # (i) exact integer stage: unit group scales, unit row scales, small operands -> every fp32 partial sum is an integer < 2^24
    small_a = torch.randint(-15, 16, (M, K), dtype=torch.int8, generator=g)
    ones = torch.ones_like(sc)
    a8_1 = X.pack_w4a8(qd, ones.to(DEV))
    got = X.w4a8_gemm(small_a.to(DEV), torch.ones(M, device=DEV), a8_1, N, TDT[dt]).float().cpu().numpy()
    want = O.w4a8_group_acc_i32(small_a.numpy(), qw.numpy()).sum(axis=0).astype(np.float64)
    want_r = O.round_to(want, dt).astype(np.float64)                 # the output cast is the only rounding
    assert np.array_equal(got.astype(np.float64), want_r)
    # (ii) reported: distance to the weight-only result on the same inputs
    ref16 = O.w4_matmul(t2n(a), qw.numpy(), t2n(sc), t2n(bias), dtype=dt)
    q_err = O.rel_l2(t2n(out), ref16)
    print(f"[w4a8] {M}x{K}x{N} {dt}: vs oracle W4A8 {err:.2e}; vs W4A16 (activation quantisation error) {q_err:.2e}")
    assert q_err < 5e-2
    # the module route (opt-in), row-wise and per-tensor
    layer = q4.DynamicQuantizeLinear(K, N, bias=True, dtype=TDT[dt])
    layer.apply_weights_(qw, sc, bias)
    layer = layer.to(DEV)
    layer.act_quant = True
    X.disable_w4a8()
    with pytest.raises(RuntimeError), torch.no_grad():             # the product alone does not serve (or import) the experiment
        layer(a.to(DEV))
    X.enable_w4a8()
    with torch.no_grad():
        assert torch.equal(layer(a.to(DEV)), out)
    layer.act_quant = "per_tensor"
    ref_t = O.w4a8_matmul(t2n(a), qw.numpy(), t2n(sc), t2n(bias), dtype=dt, per_tensor=True)
    with torch.no_grad():
        assert O.rel_l2(t2n(layer(a.to(DEV))), ref_t) <= {"f16": 3e-4, "bf16": 2e-3}[dt]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,N,bias", [(512, 4096, 4096, False), (300, 384, 264, True), (1000, 13696, 520, False), (256, 128, 256, True)])
def test_dense256_two_launch_gemm_matches_the_fused_kernel(M, K, N, bias, dtype):
    """Round-4 experiment (w4_dense256.hip): the call's int4g32 weights dequantised ONCE into a 16-bit fragment-major image
    (qlinear_dev_dense256_expand, the reference's per-weight rounding) + the dense ring GEMM on it, against the fused 256-tile kernel:
    same dequantised weights, another summation order inside the MFMAs (<= 1e-4 relative in fp16); plain, + residual, SiLU * gate;
    ragged M / N, K with an odd number of 64-deep steps; the int8 weight-only form against its fused twin."""
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd.int4.qlinear import DynamicQuantizeLinear as Q4
    from chatglm_q_amd.int8 import hip_ops as H8
    from chatglm_q_amd.int8.qlinear import DynamicQuantizeLinear as Q8
    g = torch.Generator(device=DEV).manual_seed(M + K + N)
    tol = 1e-4 if dtype == torch.float16 else 2e-3
    layer = Q4(K, N, bias=bias, dtype=dtype, device=DEV)
    layer.weight.copy_(torch.randint(0, 256, layer.weight.shape, dtype=torch.uint8, device=DEV, generator=g))
    layer.weight_scale.copy_((torch.rand(layer.weight_scale.shape, device=DEV, generator=g) * 0.01 + 0.002).to(dtype))
    if bias:
        layer.bias.copy_((torch.randn(N, device=DEV, generator=g) * 0.1).to(dtype))
    x = torch.randn(M, K, device=DEV, generator=g).to(dtype)
    h = torch.randn(M, N, device=DEV, generator=g).to(dtype)
    tiled = layer.tiled()
    want = H4.w4_forward_tiled256(x, tiled, N, layer.bias) if hasattr(H4, "w4_forward_tiled256") else layer(x)
    img = X.dense256_image(tiled, N, K, dtype)
    got = X.dense256_forward(x, img, N, layer.bias)
    assert O.rel_l2(t2n(got), t2n(want)) <= tol
    got_r = X.dense256_forward(x, img, N, layer.bias, residual=h)
    assert O.rel_l2(t2n(got_r), t2n(h + want)) <= tol
    if N % 32 == 0:
        gt, gb = layer.gated_tiled(N // 2)
        got_g = X.dense256_forward(x, X.dense256_image(gt, N, K, dtype), N, gb, gate=True)
        assert O.rel_l2(t2n(got_g), t2n(F_.silu_mul(want, N // 2))) <= 4 * tol
    if K % 64 == 0:
        l8 = Q8(K, N, bias=bias, dtype=dtype, device=DEV)
        l8.weight.copy_(torch.randint(-128, 128, l8.weight.shape, dtype=torch.int8, device=DEV, generator=g))
        l8.weight_scale.copy_((torch.rand(N, device=DEV, generator=g) * 0.001 + 0.0002).to(dtype))
        if bias:
            l8.bias.copy_((torch.randn(N, device=DEV, generator=g) * 0.1).to(dtype))
        t8 = H8.tile_w8(l8.weight)
        want8 = H8.w8_forward_tiled(x, t8, N, l8.weight_scale, l8.bias)
        got8 = X.dense256_forward(x, X.dense256_image(t8, N, K, dtype, scale=l8.weight_scale), N, l8.bias)
        assert O.rel_l2(t2n(got8), t2n(want8)) <= tol


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N", [(512, 4096), (256, 4096), (384, 4608), (1024, 4096)])
def test_w8a8_grid_split_k_equals_the_product_kernel_bit_for_bit(M, N, dtype):
    """Round 6 experiment (BASELINE config 3's shape class): int8 x int8 on 128 x 128 tiles x 2 grid-level K slices with an exact int32
    hand-off between the two workgroups of a tile.  Integer sums and the same epilogue: bit-equal to qlinear_w8a8_fwd_tiled - one call,
    40 back-to-back calls on rotating weights (the tickets' parity / epoch scheme), shapes taking turns on ONE workspace (the layout does
    not depend on the shape: a shape-dependent one turned one shape's flags into another's tickets), and replayed from a graph."""
    from chatglm_q_amd.int8 import hip_ops as h8
    K = 4096
    assert X.w8a8_splitk_serves(M, N, K) > 0 and X.w8a8_splitk_serves(512, 4096, 2048) == 0 and X.w8a8_splitk_serves(500, 4096, 4096) == 0
    g = torch.Generator(device=DEV).manual_seed(13 + M)
    ws = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device=DEV, generator=g) for _ in range(3)]
    tiled = [h8.tile_w8(w) for w in ws]
    sc = (torch.rand(N, device=DEV, generator=g) * 0.01 + 0.001).to(dtype)
    bias = (torch.randn(N, device=DEV, generator=g) * 0.1).to(dtype)
    a_q, a_s = h8.act_quant_rowwise(torch.randn(M, K, device=DEV, dtype=torch.float16))
    want = [h8.w8a8_gemm_tiled(a_q, a_s, t, N, sc, bias) for t in tiled]
    want_nobias = h8.w8a8_gemm_tiled(a_q, a_s, tiled[0], N, sc)
    assert torch.equal(X.w8a8_gemm_tiled_splitk(a_q, a_s, tiled[0], N, sc, bias), want[0])
    assert torch.equal(X.w8a8_gemm_tiled_splitk(a_q, a_s, tiled[0], N, sc), want_nobias)
    outs = [X.w8a8_gemm_tiled_splitk(a_q, a_s, tiled[i % 3], N, sc, bias) for i in range(40)]
    torch.cuda.synchronize()
    assert all(torch.equal(o, want[i % 3]) for i, o in enumerate(outs))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        X.w8a8_gemm_tiled_splitk(a_q, a_s, tiled[0], N, sc, bias)
        s.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            outs = [X.w8a8_gemm_tiled_splitk(a_q, a_s, tiled[i % 3], N, sc, bias) for i in range(8)]
        for _ in range(3):
            gr.replay()
        s.synchronize()
    assert all(torch.equal(o, want[i % 3]) for i, o in enumerate(outs))
    with pytest.raises(ValueError):
        X.w8a8_gemm_tiled_splitk(a_q[:100], a_s[:100], tiled[0], N, sc)
