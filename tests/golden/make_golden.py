#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Runs only in the build container, where the reference checkout is mounted at
/root/reference (read-only).  Nothing from the reference is copied: this script
imports it, feeds it seeded inputs and stores inputs + outputs as .npz data.

Two independent reference routes are recorded for every matmul case:
  * ``out_fallback``  - the reference's torch path on CPU tensors
                        (chatglm_q/int4/qlinear.py:50, chatglm_q/int8/qlinear.py:38),
                        reached through DynamicQuantizeLinear.forward / the functional form.
  * ``out_triton``    - the reference's actual @triton.jit kernels executed by the
                        Triton interpreter (TRITON_INTERPRET=1) on CPU tensors
                        (chatglm_q/int4/triton_ops.py:18-87, chatglm_q/int8/triton_ops.py:13-84).

bf16 arrays are stored as uint16 bit patterns (numpy has no bf16); key suffix ``_bf16bits``.

Every generator pins ``torch.set_num_threads`` (1 for the small fixtures, 8 for the real-dimension model fixtures), so that the
fp16 outputs of torch's CPU GEMM - whose summation order follows the thread split - regenerate byte for byte.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys

os.environ["TRITON_INTERPRET"] = "1"
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

import math  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

from chatglm_q.int4 import qlinear as ref4  # noqa: E402
from chatglm_q.int4 import quantizer as refq4  # noqa: E402
from chatglm_q.int4 import triton_ops as reft4  # noqa: E402
from chatglm_q.int8 import qlinear as ref8  # noqa: E402
from chatglm_q.int8 import quantizer as refq8  # noqa: E402
from chatglm_q.int8 import triton_ops as reft8  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def to_np(t: torch.Tensor):
    if t.dtype == torch.bfloat16:
        return t.contiguous().view(torch.int16).numpy().view(np.uint16)
    return t.contiguous().numpy()


def put(d: dict, name: str, t: torch.Tensor):
    if t.dtype == torch.bfloat16:
        d[name + "_bf16bits"] = to_np(t)
    else:
        d[name] = to_np(t)


def triton_int4(a, bq, bs):
    """Reference int4 Triton kernel under the interpreter (wrapper bypassed: it asserts CUDA).

    Returns None for bf16: with triton 3.6 the kernel's ``int8 * bf16`` promotes to fp32 and
    ``tl.dot`` rejects the mixed operands, so the reference Triton route does not exist for bf16
    in this environment; bf16 cases are pinned by the fallback route only."""
    if a.dtype == torch.bfloat16:
        return None
    out_shape = (*a.shape[:-1], bq.shape[1])
    a2 = a.flatten(0, -2).contiguous()
    M, K = a2.shape
    G, N = bs.shape
    group_k = K // G
    block_k = min(64, group_k)
    c = torch.empty((M, N), dtype=a.dtype)
    grid = (math.ceil(M / 16) * math.ceil(N / 128),)
    kern = reft4._dynamic_quant_matmul_s4_kernel
    kern[grid](a2, bq, bs, c, M, N, K,
               a2.stride(0), a2.stride(1), bq.stride(0), bq.stride(1),
               bs.stride(0), bs.stride(1), c.stride(0), c.stride(1),
               BLOCK_K=block_k, GROUP_K=group_k, allow_tf32=False)
    return c.reshape(out_shape)


def triton_int8(a, b_kn, bs):
    if a.dtype == torch.bfloat16:
        return None     # same triton-3.6 bf16 promotion limitation as triton_int4
    out_shape = (*a.shape[:-1], b_kn.shape[1])
    a2 = a.flatten(0, -2).contiguous()
    M, K = a2.shape
    N = b_kn.shape[1]
    c = torch.empty((M, N), dtype=a.dtype)
    grid = (math.ceil(M / 16) * math.ceil(N / 128), 1)
    kern = reft8._dynamic_quant_matmul_kernel
    kern[grid](a2, b_kn, bs, c, M, N, K,
               a2.stride(0), a2.stride(1), b_kn.stride(0), b_kn.stride(1), bs.stride(0),
               c.stride(0), c.stride(1), allow_tf32=False)
    return c.reshape(out_shape)


def gen_int4():
    torch.set_num_threads(1)                 # byte-reproducible: torch's fp16 CPU GEMM changes its summation order with the thread split
    cases = [
        # name, a-shape, K, N, dtype, bias
        ("ref_test_shape", (32, 512), 512, 256, "f32", False),   # tests/test_triton_ops_int4.py:12-13
        ("decode_k4096", (1, 4096), 4096, 128, "f16", False),
        ("ragged_k448_n384", (1, 448), 448, 384, "f16", True),   # 14 groups, N % 128 != 0
        ("bf16_small", (3, 128), 128, 128, "bf16", True),
        ("m17", (17, 512), 512, 256, "f16", False),
        ("rank3", (2, 3, 256), 256, 128, "f16", True),
        ("f32_bias", (5, 256), 256, 160, "f32", True),
        ("bf16_m9", (9, 512), 512, 64, "bf16", False),
    ]
    d = {}
    names = []
    for i, (name, ashape, K, N, dt, has_bias) in enumerate(cases):
        torch.manual_seed(1000 + i)
        tdt = DT[dt]
        a = torch.randn(ashape).to(tdt)
        w = (torch.randn((K, N)) / math.sqrt(K)).to(tdt)
        bq, bs = refq4.quantize_int4(w)
        assert bs.dtype == tdt
        bias = (torch.randn(N) * 0.1).to(tdt) if has_bias else None
        layer = ref4.DynamicQuantizeLinear(K, N, bias=has_bias, dtype=tdt)
        layer.apply_weights_(bq, bs, bias)
        out_fb = layer(a)                                     # route 1 incl. bias add
        out_tr = triton_int4(a, bq, bs)                       # route 2
        if has_bias and out_tr is not None:
            out_tr += bias                                    # same in-place add as qlinear.py:92-93
        dense = ref4.unpack_int4(bq, bs)
        p = f"{name}/"
        put(d, p + "a", a); d[p + "qweight"] = to_np(bq); put(d, p + "scale", bs)
        if has_bias:
            put(d, p + "bias", bias)
        if K * N <= 65536:                                    # keep the fixture file small
            put(d, p + "dense", dense)
        put(d, p + "out_fallback", out_fb)
        names.append(f"{name}:{dt}:{int(has_bias)}")
        if out_tr is not None:
            put(d, p + "out_triton", out_tr)
            print("int4", name, dt, "fallback-vs-triton max abs",
                  (out_fb.float() - out_tr.float()).abs().max().item())

    # hand-built edge case: nibble 0 (-> -8, never produced by the quantiser but decoded by the
    # kernel), nibble 15 (+7), an all-zero group (scale clamps to 1e-10 in fp32).
    torch.manual_seed(77)
    K, N = 64, 128
    bq = torch.randint(0, 256, (K // 2, N), dtype=torch.uint8)
    bq[0, :] = 0x00
    bq[1, :] = 0xFF
    bq[2, :] = 0x0F
    bq[3, :] = 0xF0
    w0 = torch.randn((K, N)) / 8
    w0[32:, :16] = 0.0                                        # all-zero group for 16 columns
    _, bs = refq4.quantize_int4(w0)
    a = torch.randn((4, K))
    out_fb = ref4.dynamic_quant_matmul(a, bq, bs)
    out_tr = triton_int4(a, bq, bs)
    p = "edge_nibbles/"
    put(d, p + "a", a); d[p + "qweight"] = to_np(bq); put(d, p + "scale", bs)
    put(d, p + "dense", ref4.unpack_int4(bq, bs)); put(d, p + "out_fallback", out_fb); put(d, p + "out_triton", out_tr)
    names.append("edge_nibbles:f32:0")
    d["__cases__"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "int4_matmul.npz"), **d)


def gen_int8():
    torch.set_num_threads(1)                 # byte-reproducible: torch's fp16 CPU GEMM changes its summation order with the thread split
    d = {}
    names = []
    # reference test shape: tests/test_triton_ops.py:10-12 (B given as contiguous (K, N), negative scales)
    torch.manual_seed(2000)
    A = torch.randn((10, 128))
    B = torch.randint(-127, 127, (128, 256), dtype=torch.int8)
    S = torch.randn((256,)) / 256
    p = "ref_test_shape/"
    put(d, p + "a", A); d[p + "w_kn"] = to_np(B); put(d, p + "scale", S)
    put(d, p + "out_fallback", ref8.dynamic_quant_matmul(A, B, S)); put(d, p + "out_triton", triton_int8(A, B, S))
    names.append("ref_test_shape:f32:0:kn")

    cases = [
        ("decode_k4096", (1, 4096), 4096, 64, "f16", False),
        ("m128", (128, 512), 512, 384, "f32", True),
        ("bf16_m3", (3, 256), 256, 128, "bf16", True),
        ("rank3_f16", (2, 5, 384), 384, 192, "f16", True),
        ("k_tail_f16", (7, 200), 200, 96, "f16", False),       # K % 64 != 0 -> masked tail (triton_ops.py:66-69)
    ]
    for i, (name, ashape, K, N, dt, has_bias) in enumerate(cases):
        torch.manual_seed(2001 + i)
        tdt = DT[dt]
        a = torch.randn(ashape).to(tdt)
        w = (torch.randn((N, K)) / math.sqrt(K)).to(tdt)
        wq, ws = refq8.quantize_int8(w)
        bias = (torch.randn(N) * 0.1).to(tdt) if has_bias else None
        layer = ref8.DynamicQuantizeLinear(K, N, bias=has_bias, dtype=tdt)
        layer.apply_weights_(wq, ws, bias)
        out_fb = layer(a)
        out_tr = triton_int8(a, wq.t(), ws)                  # (K, N) view with strides (1, K), as qlinear.py:90
        if has_bias and out_tr is not None:
            out_tr += bias
        p = f"{name}/"
        put(d, p + "a", a); d[p + "weight_nk"] = to_np(wq); put(d, p + "scale", ws)
        if has_bias:
            put(d, p + "bias", bias)
        put(d, p + "out_fallback", out_fb)
        names.append(f"{name}:{dt}:{int(has_bias)}:nk")
        if out_tr is not None:
            put(d, p + "out_triton", out_tr)
            print("int8", name, dt, "fallback-vs-triton max abs",
                  (out_fb.float() - out_tr.float()).abs().max().item())

    # edge: weight -128 (never produced by the quantiser, representable in the buffer)
    torch.manual_seed(88)
    wq = torch.randint(-128, 128, (64, 96), dtype=torch.int8)
    wq[:, 0] = -128
    ws = (torch.rand(64) / 100 + 1e-3)
    a = torch.randn((3, 96))
    p = "edge_m128/"
    put(d, p + "a", a); d[p + "weight_nk"] = to_np(wq); put(d, p + "scale", ws)
    put(d, p + "out_fallback", ref8.dynamic_quant_matmul(a, wq.t(), ws)); put(d, p + "out_triton", triton_int8(a, wq.t(), ws))
    names.append("edge_m128:f32:0:nk")
    d["__cases__"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "int8_matmul.npz"), **d)


def gen_quantizers():
    torch.set_num_threads(1)                 # byte-reproducible: torch's fp16 CPU GEMM changes its summation order with the thread split
    d = {}
    for dt in ("f32", "f16"):
        tdt = DT[dt]
        torch.manual_seed(3000)
        w = (torch.randn((128, 96)) / 8).to(tdt)
        # exact ties: make group 0 of column 0 have max 7.0 so scale == 1 and x.5 values tie
        w[:32, 0] = torch.tensor([7.0, 0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 3.5] * 4).to(tdt)
        w[32:64, 1] = 0.0                                     # all-zero group -> scale floor
        q, s = refq4.quantize_int4(w)
        put(d, f"int4_{dt}/w", w); d[f"int4_{dt}/q"] = to_np(q); put(d, f"int4_{dt}/scale", s)
        x = (torch.randn((24, 160))).to(tdt)
        x[0, :8] = torch.tensor([127.0, 0.5, 1.5, 2.5, -0.5, -1.5, 63.5, -126.5]).to(tdt)
        x[0, 8:] = x[0, 8:].clamp(-100, 100)
        x[1, :] = 0.0
        q8, s8 = refq8.quantize_int8(x)
        put(d, f"int8_{dt}/x", x); d[f"int8_{dt}/q"] = to_np(q8); put(d, f"int8_{dt}/scale", s8)
    np.savez_compressed(os.path.join(OUT, "quantizers.npz"), **d)


def gen_w8a8():
    """Exact-integer stage of the W8A8 semantic: quantize_int8 on activations
    (fp32 arithmetic) composed with an integer matmul; plus the reference's W8A16 output on the
    same inputs so the quantisation error can be reported."""
    torch.set_num_threads(1)                 # byte-reproducible: torch's fp16 CPU GEMM changes its summation order with the thread split
    d = {}
    torch.manual_seed(4000)
    M, K, N = 16, 512, 256
    a = torch.randn((M, K))
    w = torch.randn((N, K)) / math.sqrt(K)
    wq, ws = refq8.quantize_int8(w)
    aq, a_s = refq8.quantize_int8(a)
    acc = aq.to(torch.int32) @ wq.to(torch.int32).t()
    out = acc.float() * (a_s[:, None] * ws[None, :])
    put(d, "a", a); d["weight_nk"] = to_np(wq); put(d, "w_scale", ws)
    d["a_q"] = to_np(aq); put(d, "a_scale", a_s); d["acc_i32"] = to_np(acc); put(d, "out_w8a8", out)
    put(d, "out_w8a16", ref8.dynamic_quant_matmul(a, wq.t(), ws))
    # fp16 activations -> quantised in fp32 arithmetic
    a16 = a.half()
    aq16, as16 = refq8.quantize_int8(a16.float())
    put(d, "a_f16", a16); d["a_q_f16"] = to_np(aq16); put(d, "a_scale_f16", as16)
    # per-tensor symmetric variant: the formula of DynamicQuantizeMatMul.symbolic's second branch
    # (chatglm_q/int8/qlinear.py:64-70) written with torch ops - the reference itself only emits it as ONNX nodes
    # (ReduceMax(Abs(A)) / 127 -> QuantizeLinear zero point 0 -> MatMulInteger -> Cast -> Mul) and no ONNX runtime exists
    # here; QuantizeLinear rounds half to even and saturates, as torch.round + clamp do
    a_scale_t = (a.abs().max() / 127.0).float()
    aq_t = torch.clamp(torch.round(a / a_scale_t), -128, 127).to(torch.int8)
    acc_t = aq_t.to(torch.int32) @ wq.to(torch.int32).t()
    d["pt_a_q"] = to_np(aq_t); put(d, "pt_a_scale", a_scale_t.reshape(1)); d["pt_acc_i32"] = to_np(acc_t)
    put(d, "pt_out", acc_t.float() * (a_scale_t * ws[None, :]))
    np.savez_compressed(os.path.join(OUT, "w8a8.npz"), **d)


def gen_embedding():
    torch.set_num_threads(1)                 # byte-reproducible: torch's fp16 CPU GEMM changes its summation order with the thread split
    d = {}
    torch.manual_seed(5000)
    V, D = 128, 64
    emb = torch.randn((V, D)).half()
    q4, s4 = refq4.quantize_int4(emb)          # packs along the vocabulary axis (quantizer.py:70)
    e4 = ref4.QEmbedding(V, D, dtype=torch.float16)
    e4.apply_weights_(q4, s4)
    ids = torch.tensor([[0, 1, 2, 31, 32, 33, 127], [5, 64, 65, 96, 97, 126, 3]])
    d["ids"] = ids.numpy()
    d["int4/qweight"] = to_np(q4); put(d, "int4/scale", s4); put(d, "int4/out", e4(ids))
    q8, s8 = refq8.quantize_int8(emb.t().float())
    e8 = ref8.QEmbedding(V, D, dtype=torch.float32)
    e8.apply_weights_(q8.t(), s8)
    d["int8/weight"] = to_np(q8.t().contiguous()); put(d, "int8/scale", s8); put(d, "int8/out", e8(ids))
    np.savez_compressed(os.path.join(OUT, "qembedding.npz"), **d)


def gen_model():
    """Tiny-config int4g32 ChatGLM2 (hidden 128, FFN 224 = 7 groups, 2 layers, vocab 256): every buffer, the ids, prefill
    logits, one cached decode step, and the sampler's output on fixed logits.  Pins the build's own
    model graph / decode loop (harness for BASELINE configs 4 and 5)."""
    torch.set_num_threads(1)                 # byte-reproducible: torch's fp16 CPU GEMM changes its summation order with the thread split
    from chatglm_q import model as refm
    from chatglm_q import loader as refl
    from chatglm_q.decoder import top_p_sampling
    cfg = refm.ChatGLM2Config(hidden_size=128, inner_hidden_size=224, head_hidden_size=32,
                              num_multi_query_groups=2, num_attention_heads=4, num_layers=2,
                              vocab_size=256, max_sequence_length=64)
    d = {}
    for dt in ("f32", "f16"):
        tdt = DT[dt]
        torch.manual_seed(6000)
        m = refl.create_quant_int4_model(cfg, dtype=tdt)
        sd = m.state_dict()
        gen = torch.Generator().manual_seed(6001)
        for k, v in sd.items():
            if v.dtype == torch.uint8:
                v.copy_(torch.randint(0, 256, v.shape, dtype=torch.uint8, generator=gen))
            elif k.endswith("weight_scale"):
                v.copy_((torch.rand(v.shape, generator=gen) * 0.02 + 0.005).to(v.dtype))
            elif k.endswith("bias"):
                v.copy_((torch.randn(v.shape, generator=gen) * 0.02).to(v.dtype))
            elif "ln" in k or "norm" in k:
                v.copy_((1.0 + 0.1 * torch.randn(v.shape, generator=gen)).to(v.dtype))
            else:
                v.copy_((torch.randn(v.shape, generator=gen) * 0.02).to(v.dtype))
        m.eval()
        ids = torch.randint(0, 256, (1, 12), generator=gen)
        with torch.no_grad():
            _, logits, kv = m(input_ids=ids)
            nxt = torch.randint(0, 256, (1, 1), generator=gen)
            _, logits2, kv2 = m(input_ids=nxt, past_key_values=kv)
            # chunked prefill: 7 + 5 tokens
            _, la, kva = m(input_ids=ids[:, :7])
            _, lb, kvb = m(input_ids=ids[:, 7:], past_key_values=kva)
        p = f"{dt}/"
        for k, v in sd.items():
            put(d, p + "sd/" + k, v)
        d[p + "ids"] = ids.numpy(); d[p + "next_id"] = nxt.numpy()
        put(d, p + "prefill_logits", logits); put(d, p + "decode_logits", logits2)
        put(d, p + "chunked_last_logits", lb[:, -1])
        put(d, p + "kv0_k", kv2[0][0]); put(d, p + "kv0_v", kv2[0][1])
        print("model", dt, "chunked-vs-full last-logit max abs",
              (lb[:, -1].float() - logits[:, -1].float()).abs().max().item())
    d["config"] = np.array([cfg.hidden_size, cfg.inner_hidden_size, cfg.head_hidden_size,
                            cfg.num_multi_query_groups, cfg.num_attention_heads, cfg.num_layers,
                            cfg.vocab_size, cfg.max_sequence_length])
    # sampler (chatglm_q/decoder.py:12-27)
    torch.manual_seed(6100)
    lg = torch.randn(512) * 3
    probs_in = lg.clone()
    torch.manual_seed(6101)
    picks = [int(top_p_sampling(probs_in, top_k=100, top_p=0.8, temperature=1.0)) for _ in range(8)]
    put(d, "sampler/logits", lg); d["sampler/picks_seed6101"] = np.array(picks)
    # deterministic part of the sampler: the renormalised top-k/top-p distribution
    pr = torch.softmax(lg.float() / 1.0, dim=-1)
    pr, idx = torch.sort(pr, dim=-1, descending=True)
    pr, idx = pr[:100], idx[:100]
    cum = torch.cumsum(pr, dim=-1)
    pr[(cum - pr) > 0.8] = 0.0
    pr = pr / pr.sum()
    put(d, "sampler/probs", pr); d["sampler/indices"] = idx.numpy()
    np.savez_compressed(os.path.join(OUT, "tiny_model.npz"), **d)


def gen_sampler():
    """The reference's ``top_p_sampling`` (chatglm_q/decoder.py:12-27) run on seeded logits rows (tests/_sampler_cases.py): what it
    hands to ``torch.multinomial`` (the filtered, renormalised distribution) and the index order its ``torch.sort`` produced are
    recorded by wrapping those two torch functions around the call - the function under test is the reference's own."""
    sys.path.insert(0, os.path.dirname(OUT))                        # tests/
    import _sampler_cases as SC
    from chatglm_q import decoder as refd
    torch.set_num_threads(1)
    d = {}
    for name, (seed, N, recipe, dtype, top_k, top_p, temperature) in SC.CASES.items():
        lg = torch.from_numpy(SC.logits_for(name)).to(DT[dtype])
        seen = {}
        real_sort, real_multinomial = torch.sort, torch.multinomial

        def sort_spy(*a, **kw):
            out = real_sort(*a, **kw)
            seen["indices"] = out[1].clone()
            return out

        def multinomial_spy(probs, num_samples, *a, **kw):
            seen["probs"] = probs.clone()
            if not torch.isfinite(probs).all():                      # the reference would raise here: record and move on
                return torch.zeros(probs.shape[:-1] + (num_samples,), dtype=torch.long)
            return real_multinomial(probs, num_samples, *a, **kw)

        torch.sort, torch.multinomial = sort_spy, multinomial_spy
        try:
            torch.manual_seed(seed)
            picks = [int(refd.top_p_sampling(lg, top_k, top_p, temperature)) for _ in range(4)]
        finally:
            torch.sort, torch.multinomial = real_sort, real_multinomial
        d[name + "/probs"] = seen["probs"].numpy().astype(np.float32)
        d[name + "/indices"] = seen["indices"][:top_k].numpy().astype(np.int64)
        d[name + "/picks"] = np.array(picks)
        kept = int((seen["probs"] > 0).sum())
        print("sampler", name, "N", N, "k", top_k, "kept", kept, "picks", picks)
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), **d)


def gen_model_real():
    """ChatGLM2-6B LAYER DIMENSIONS (hidden 4096, FFN 13696, 32 heads x 128, 2 groups; 2 layers, vocab 1024), fp16,
    through the reference model on CPU.  Weights come from tests/_golden.py::fill_seeded_ (a pure function of the
    state_dict keys) so that only ids / logits / cache samples are stored.  Cases:
      b1   batch 1: prefill of 9 ids, then 3 cached decode steps (BASELINE config 4's call pattern)
      b4   batch 4, LEFT padded (chatglm_q/model.py:297-318), chunked prefill 3 x 8 positions, 2 decode steps
      big  batch 4 x 2048 positions, left padded, chunked prefill 4 x 512 (BASELINE config 5's workload)"""
    sys.path.insert(0, os.path.dirname(OUT))                        # tests/
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))       # repo root (tests/_golden.py imports oracle/)
    import _golden as G
    from chatglm_q import model as refm
    from chatglm_q import loader as refl
    cfg = refm.ChatGLM2Config(**G.REAL_DIM_CONFIG)
    torch.set_num_threads(8)
    m = refl.create_quant_int4_model(cfg, dtype=torch.float16)
    G.fill_seeded_(m.state_dict())
    m.eval()
    d = {}
    gen = torch.Generator().manual_seed(8200)
    V = cfg.vocab_size

    def run(ids, attn, chunks, past=None):
        """Feed ids[:, c0:c1] chunk by chunk with the growing attention mask; returns (list of logits, kv)."""
        outs, n0 = [], 0 if past is None else past[0][0].shape[1]
        for c0, c1 in chunks:
            with torch.no_grad():
                _, lg, past = m(input_ids=ids[:, c0:c1], attention_mask=None if attn is None else attn[:, : n0 + c1],
                                past_key_values=past)
            outs.append(lg)
        return outs, past

    # ---- b1 ---------------------------------------------------------------------------------------------------
    ids = torch.randint(0, V, (1, 9), generator=gen)
    nxt = torch.randint(0, V, (3,), generator=gen)
    (lg,), kv = run(ids, None, [(0, 9)])
    d["b1/ids"] = ids.numpy(); d["b1/next_ids"] = nxt.numpy()
    put(d, "b1/prefill_logits", lg)
    for t in range(3):
        with torch.no_grad():
            _, lg, kv = m(input_ids=nxt[t].view(1, 1), past_key_values=kv)
        put(d, f"b1/decode_logits_{t}", lg[:, -1])
    put(d, "b1/kv1_k", kv[1][0]); put(d, "b1/kv1_v", kv[1][1])           # (1, 12, 2, 1, 128) each
    print("b1 done")

    # ---- b4: left padded, chunked ------------------------------------------------------------------------------
    S, lens = 24, [24, 17, 9, 21]
    ids = torch.randint(0, V, (4, S), generator=gen)
    attn = torch.zeros(4, S + 2, dtype=torch.long)
    for b, n in enumerate(lens):
        attn[b, S - n:] = 1
        ids[b, : S - n] = 0
    outs, kv = run(ids, attn, [(0, 8), (8, 16), (16, 24)])
    (full,), _ = run(ids, attn, [(0, 24)])
    print("b4 chunked-vs-full last-logit rel", (outs[-1][:, -1].float() - full[:, -1].float()).norm().item()
          / full[:, -1].float().norm().item())
    d["b4/ids"] = ids.numpy(); d["b4/attention_mask"] = attn.numpy()
    put(d, "b4/last_chunk_logits", outs[-1])                              # (4, 8, 1024)
    nxt = torch.randint(0, V, (2, 4), generator=gen)
    d["b4/next_ids"] = nxt.numpy()
    for t in range(2):
        with torch.no_grad():
            _, lg, kv = m(input_ids=nxt[t].view(4, 1), attention_mask=attn[:, : S + t + 1], past_key_values=kv)
        put(d, f"b4/decode_logits_{t}", lg[:, -1])
    put(d, "b4/kv1_k", kv[1][0]); put(d, "b4/kv1_v", kv[1][1])           # (4, 26, 2, 1, 128)
    print("b4 done")

    # ---- big: config 5's workload -------------------------------------------------------------------------------
    S, lens = 2048, [2048, 1900, 1024, 2048]
    ids = torch.randint(0, V, (4, S), generator=gen)
    attn = torch.zeros(4, S, dtype=torch.long)
    for b, n in enumerate(lens):
        attn[b, S - n:] = 1
        ids[b, : S - n] = 0
    import time
    t0 = time.time()
    outs, kv = run(ids, attn, [(0, 512), (512, 1024), (1024, 1536), (1536, 2048)])
    print("big done in", time.time() - t0, "s")
    d["big/ids"] = ids.numpy().astype(np.int16); d["big/lens"] = np.array(lens)
    put(d, "big/last_logits", outs[-1][:, -1])                            # (4, 1024)
    rows = torch.tensor([0, 100, 255, 511])
    d["big/sample_rows"] = rows.numpy()
    put(d, "big/chunk1_logits", outs[1][:, rows])                         # positions 512 + rows: (4, 4, 1024)
    put(d, "big/chunk3_logits", outs[3][:, rows])
    pos = torch.tensor([0, 1023, 1024, 2047])
    d["big/kv_positions"] = pos.numpy()
    put(d, "big/kv1_k", kv[1][0][:, pos]); put(d, "big/kv1_v", kv[1][1][:, pos])
    d["config"] = np.array([G.REAL_DIM_CONFIG[k] for k in ("hidden_size", "inner_hidden_size", "head_hidden_size",
                                                           "num_multi_query_groups", "num_attention_heads", "num_layers",
                                                           "vocab_size", "max_sequence_length")])
    d["seed"] = np.array([G.REAL_DIM_SEED])
    np.savez_compressed(os.path.join(OUT, "real_model.npz"), **d)


def gen_model_r3():
    """Round-3 additions, in a file of their own (tests/golden/model_r3.npz) so that the round-1/2 fixtures stay byte
    identical.  All through the REFERENCE model on CPU, weights from tests/_golden.py seeded fills:
      long   int4g32, fp16, real layer dimensions, batch 1: prefill of 320 ids + 3 cached decode steps - more than 256
             cached positions, i.e. the build's window-split attention + combine launch (decode_ops.hip) against reference logits
      bf16   int4g32, bf16, real layer dimensions, batch 1: prefill of 9 ids + 3 decode steps (bf16 is pinned through the
             reference's torch route only: its Triton kernel does not run in bf16, see triton_int4 above)
      i8     int8 per-channel model (chatglm_q/loader.py:41-50 create_quant_int8_model), fp16, real layer dimensions, batch 1:
             prefill of 9 ids + 3 decode steps
      i8tiny the int8 model at a tiny config in fp32 and fp16 (prefill 12 ids + 1 decode step + chunked prefill 7 + 5)"""
    sys.path.insert(0, os.path.dirname(OUT))
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    import _golden as G
    from chatglm_q import model as refm
    from chatglm_q import loader as refl
    torch.set_num_threads(8)
    d = {}

    def b1_case(m, tag, n_prefill, seed, V):
        gen = torch.Generator().manual_seed(seed)
        ids = torch.randint(0, V, (1, n_prefill), generator=gen)
        nxt = torch.randint(0, V, (3,), generator=gen)
        with torch.no_grad():
            _, lg, kv = m(input_ids=ids)
        d[f"{tag}/ids"] = ids.numpy().astype(np.int16); d[f"{tag}/next_ids"] = nxt.numpy()
        put(d, f"{tag}/prefill_last_logits", lg[:, -1])
        for t in range(3):
            with torch.no_grad():
                _, lg, kv = m(input_ids=nxt[t].view(1, 1), past_key_values=kv)
            put(d, f"{tag}/decode_logits_{t}", lg[:, -1])
        pos = torch.tensor([0, n_prefill // 2, n_prefill - 1, n_prefill + 2])
        d[f"{tag}/kv_positions"] = pos.numpy()
        put(d, f"{tag}/kv1_k", kv[1][0][:, pos]); put(d, f"{tag}/kv1_v", kv[1][1][:, pos])
        print(tag, "done")

    cfg = refm.ChatGLM2Config(**G.REAL_DIM_CONFIG)
    m = refl.create_quant_int4_model(cfg, dtype=torch.float16)
    G.fill_seeded_(m.state_dict())
    m.eval()
    b1_case(m, "long", 320, 8300, cfg.vocab_size)
    del m
    m = refl.create_quant_int4_model(cfg, dtype=torch.bfloat16)
    G.fill_seeded_(m.state_dict())
    m.eval()
    b1_case(m, "bf16", 9, 8301, cfg.vocab_size)
    del m
    m = refl.create_quant_int8_model(cfg, dtype=torch.float16)
    G.fill_seeded_int8_(m.state_dict())
    m.eval()
    b1_case(m, "i8", 9, 8302, cfg.vocab_size)
    del m

    tcfg = refm.ChatGLM2Config(**G.TINY_INT8_CONFIG)
    for dt in ("f32", "f16"):
        m = refl.create_quant_int8_model(tcfg, dtype=DT[dt])
        G.fill_seeded_int8_(m.state_dict(), seed=8400)
        m.eval()
        gen = torch.Generator().manual_seed(8401)
        ids = torch.randint(0, tcfg.vocab_size, (1, 12), generator=gen)
        nxt = torch.randint(0, tcfg.vocab_size, (1, 1), generator=gen)
        with torch.no_grad():
            _, logits, kv = m(input_ids=ids)
            _, logits2, kv2 = m(input_ids=nxt, past_key_values=kv)
            _, la, kva = m(input_ids=ids[:, :7])
            _, lb, kvb = m(input_ids=ids[:, 7:], past_key_values=kva)
        p = f"i8tiny/{dt}/"
        d[p + "ids"] = ids.numpy(); d[p + "next_id"] = nxt.numpy()
        put(d, p + "prefill_logits", logits); put(d, p + "decode_logits", logits2)
        put(d, p + "chunked_last_logits", lb[:, -1])
        put(d, p + "kv0_k", kv2[0][0]); put(d, p + "kv0_v", kv2[0][1])
    np.savez_compressed(os.path.join(OUT, "model_r3.npz"), **d)


def gen_model_full():
    """Round 5 (tests/golden/model_full.npz): the WHOLE ChatGLM2-6B geometry - 28 layers, hidden 4096, FFN 13696, 32 heads x 128,
    2 groups, vocabulary 65024 (chatglm_q/model.py:17-33 defaults) - int4g32, fp16, through the reference model on CPU
    (chatglm_q/decoder.py:65-108's call pattern: one prefill, then cached one-token steps).  Weights from tests/_golden.py::fill_seeded_
    (a pure function of the state_dict keys: nothing of the 3.4 GB is stored).  Stored: 32 prompt ids, 4 teacher-forced next ids, the
    last-position logits of the prefill and of every decode step (fp16, 65024 each), the reference's own greedy choice per step and
    four cache rows of the LAST layer."""
    sys.path.insert(0, os.path.dirname(OUT))
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    import time
    import _golden as G
    from chatglm_q import model as refm
    from chatglm_q import loader as refl
    torch.set_num_threads(8)
    cfg = refm.ChatGLM2Config(**G.FULL_CONFIG)
    t0 = time.time()
    m = refl.create_quant_int4_model(cfg, dtype=torch.float16)
    G.fill_seeded_(m.state_dict(), seed=G.FULL_SEED)
    m.eval()
    print("model filled in", round(time.time() - t0, 1), "s")
    d = {}
    gen = torch.Generator().manual_seed(8500)
    V = cfg.vocab_size
    ids = torch.randint(0, V, (1, 32), generator=gen)
    nxt = torch.randint(0, V, (4,), generator=gen)
    d["ids"] = ids.numpy().astype(np.int32); d["next_ids"] = nxt.numpy().astype(np.int32)
    t0 = time.time()
    with torch.no_grad():
        _, lg, kv = m(input_ids=ids)
    print("prefill", round(time.time() - t0, 1), "s")
    put(d, "prefill_last_logits", lg[:, -1])
    greedy = [int(lg[0, -1].float().argmax())]
    for t in range(4):
        t0 = time.time()
        with torch.no_grad():
            _, lg, kv = m(input_ids=nxt[t].view(1, 1), past_key_values=kv)
        put(d, f"decode_logits_{t}", lg[:, -1])
        greedy.append(int(lg[0, -1].float().argmax()))
        print("decode", t, round(time.time() - t0, 1), "s")
    d["greedy_ids"] = np.array(greedy, dtype=np.int32)
    pos = torch.tensor([0, 15, 31, 35])
    d["kv_positions"] = pos.numpy()
    L = cfg.num_layers - 1
    put(d, "kv_last_k", kv[L][0][:, pos]); put(d, "kv_last_v", kv[L][1][:, pos])
    d["config"] = np.array([G.FULL_CONFIG[k] for k in ("hidden_size", "inner_hidden_size", "head_hidden_size",
                                                       "num_multi_query_groups", "num_attention_heads", "num_layers",
                                                       "vocab_size", "max_sequence_length")])
    d["seed"] = np.array([G.FULL_SEED])
    np.savez_compressed(os.path.join(OUT, "model_full.npz"), **d)


def gen_backward():
    """grad_A of both quantized matmuls through the reference's own autograd functions (CPU route:
    chatglm_q/int4/qlinear.py:53-64, chatglm_q/int8/qlinear.py:41-52)."""
    torch.set_num_threads(1)                 # byte-reproducible: torch's fp16 CPU GEMM changes its summation order with the thread split
    d = {}
    names = []
    cases = [("i4_f16_m5", 4, (5, 256), 256, 192, "f16"), ("i4_bf16_m3", 4, (3, 128), 128, 64, "bf16"),
             ("i4_f32_m4", 4, (4, 128), 128, 96, "f32"), ("i4_f16_rank3", 4, (2, 3, 192), 192, 128, "f16"),
             ("i8_f16_m5", 8, (5, 256), 256, 192, "f16"), ("i8_bf16_m3", 8, (3, 128), 128, 64, "bf16"),
             ("i8_f32_m4", 8, (4, 96), 96, 80, "f32")]
    for i, (name, bits, ashape, K, N, dt) in enumerate(cases):
        torch.manual_seed(7000 + i)
        tdt = DT[dt]
        a = torch.randn(ashape).to(tdt).requires_grad_(True)
        w = (torch.randn((K, N)) / math.sqrt(K)).to(tdt)
        go = torch.randn((*ashape[:-1], N)).to(tdt)
        p = f"{name}/"
        if bits == 4:
            bq, bs = refq4.quantize_int4(w)
            out = ref4.dynamic_quant_matmul(a, bq, bs)
            d[p + "qweight"] = to_np(bq)
        else:
            wq, bs = refq8.quantize_int8(w.t().contiguous())          # (N, K) int8, (N) scales: the module's buffers
            out = ref8.dynamic_quant_matmul(a, wq.t(), bs)
            d[p + "weight_nk"] = to_np(wq)
        out.backward(go)
        put(d, p + "a", a.detach()); put(d, p + "scale", bs); put(d, p + "grad_out", go); put(d, p + "grad_a", a.grad)
        names.append(f"{name}:{bits}:{dt}")
    d["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "backward.npz"), **d)


def gen_loader():
    """config.json text, state_dict key order and the greedy shard plan of the reference for a tiny model
    (chatglm_q/loader.py:16-38,139-150) - pins the build's checkpoint-folder writer/reader."""
    import json
    from chatglm_q.loader import ChatGLMLoadConfig, create_quant_int4_model
    from chatglm_q.model import ChatGLM2Config
    cfg = ChatGLM2Config(hidden_size=128, inner_hidden_size=224, head_hidden_size=32, num_multi_query_groups=2,
                         num_attention_heads=4, num_layers=2, vocab_size=256, max_sequence_length=64)
    lc = ChatGLMLoadConfig(model_config=cfg, quant_type="int4g32",
                           weight_files=["model_weights_0.safetensors", "model_weights_1.safetensors"], torch_dtype="float16")
    m = create_quant_int4_model(cfg, dtype=torch.float16)
    mapping, idx, size = {}, 0, 0
    for name, w in m.state_dict().items():           # the loop of save_model_and_tokenizer with a 40 kB shard limit
        s = w.element_size() * w.numel()
        if size + s > 40000:
            idx += 1
            size = 0
        size += s
        mapping[name] = f"model_weights_{idx}.safetensors"
    json.dump({"config_json": lc.to_json(), "shard_mapping_40000": mapping, "state_keys": list(m.state_dict().keys())},
              open(os.path.join(OUT, "loader_format.json"), "w"), indent=1)


if __name__ == "__main__":
    which = sys.argv[1:] or ["int4", "int8", "quantizers", "w8a8", "embedding", "model", "model_real", "model_r3", "model_full", "backward", "loader", "sampler"]
    torch.set_num_threads(4)
    for w in which:
        {"int4": gen_int4, "int8": gen_int8, "quantizers": gen_quantizers, "w8a8": gen_w8a8,
         "embedding": gen_embedding, "model": gen_model, "model_real": gen_model_real, "model_r3": gen_model_r3, "model_full": gen_model_full, "backward": gen_backward, "loader": gen_loader, "sampler": gen_sampler}[w]()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
