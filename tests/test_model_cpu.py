"""The build's ChatGLM2 graph (caller of the QLinear path) against logits produced by the reference model
itself on a tiny int4g32 configuration (tests/golden/tiny_model.npz).  CPU branch of the modules."""
import numpy as np
import pytest
import torch

import _golden as G
from oracle import qlinear_oracle as O
from chatglm_q_amd import model as M

Z = G.load("tiny_model.npz")
TDT = {"f32": torch.float32, "f16": torch.float16}


def build(dt, device="cpu"):
    c = [int(v) for v in Z["config"]]
    cfg = M.ChatGLM2Config(hidden_size=c[0], inner_hidden_size=c[1], head_hidden_size=c[2], num_multi_query_groups=c[3],
                           num_attention_heads=c[4], num_layers=c[5], vocab_size=c[6], max_sequence_length=c[7])
    model = M.create_quant_int4_model(cfg, dtype=TDT[dt])
    sd = model.state_dict()
    pre = f"{dt}/sd/"
    keys = [k[len(pre):] for k in Z.files if k.startswith(pre)]
    assert sorted(keys) == sorted(sd.keys())              # same parameter / buffer names as the reference
    for k in keys:
        sd[k].copy_(torch.from_numpy(Z[pre + k]))
    return model.to(device).eval(), cfg


def t2n(t):
    return t.detach().float().cpu().numpy()


# The CPU branch runs the reference's own torch formula: measured 0.0 (fp16, bit-identical) and 2.3e-7 (fp32) in the
# build container.  The bars leave room only for a different BLAS summation order on another host: one fp16 rounding of
# a dot product per layer output (~1e-3, the same bar as the real-dimension fixture), fp32 roundoff.
TOL = {"f32": 2e-5, "f16": 2e-3}


@pytest.mark.parametrize("dt", ["f32", "f16"])
def test_prefill_decode_and_chunked_match_reference(dt):
    model, cfg = build(dt)
    ids = torch.from_numpy(Z[f"{dt}/ids"])
    nxt = torch.from_numpy(Z[f"{dt}/next_id"])
    with torch.no_grad():
        _, logits, kv = model(input_ids=ids)
        assert O.rel_l2(t2n(logits), Z[f"{dt}/prefill_logits"].astype(np.float32)) < TOL[dt]
        _, logits2, kv2 = model(input_ids=nxt, past_key_values=kv)
        assert O.rel_l2(t2n(logits2), Z[f"{dt}/decode_logits"].astype(np.float32)) < TOL[dt]
        assert kv2[0][0].shape == Z[f"{dt}/kv0_k"].shape          # (B, T, groups, 1, d_head)
        assert O.rel_l2(t2n(kv2[0][0]), Z[f"{dt}/kv0_k"].astype(np.float32)) < TOL[dt]
        assert O.rel_l2(t2n(kv2[0][1]), Z[f"{dt}/kv0_v"].astype(np.float32)) < TOL[dt]
        _, la, kva = model(input_ids=ids[:, :7])
        _, lb, _ = model(input_ids=ids[:, 7:], past_key_values=kva)
        assert O.rel_l2(t2n(lb[:, -1]), Z[f"{dt}/chunked_last_logits"].astype(np.float32)) < TOL[dt]


def test_static_cache_step_equals_reference_shaped_forward():
    model, cfg = build("f32")
    ids = torch.from_numpy(Z["f32/ids"])
    S = ids.shape[1]
    cap = 24
    with torch.no_grad():
        _, ref_logits, kv = model(input_ids=ids)
        cache = model.new_cache(1, cap)
        t = torch.arange(cap)
        pos = torch.arange(1, S + 1)[None]                       # positions start at 1
        mask = ((t[None, :] > torch.arange(S)[:, None]).float() * -1e10)[None]
        out = model.step(ids, cache, torch.arange(S), pos, mask)
        assert torch.allclose(out, ref_logits, atol=1e-5, rtol=1e-5)
        cache.length = S
        nxt = torch.from_numpy(Z["f32/next_id"])
        _, ref2, _ = model(input_ids=nxt, past_key_values=kv)
        mask1 = ((t > S).float() * -1e10)[None, None]
        out2 = model.step(nxt, cache, torch.tensor([S]), torch.tensor([[S + 1]]), mask1, last_only=True)
        assert torch.allclose(out2, ref2, atol=1e-5, rtol=1e-5)
        assert O.rel_l2(t2n(out2), Z["f32/decode_logits"].astype(np.float32)) < 2e-4


def test_sampler_matches_reference():
    from chatglm_q_amd.decoder import filtered_distribution, top_p_sampling
    lg = torch.from_numpy(Z["sampler/logits"])
    probs, idx = filtered_distribution(lg, top_k=100, top_p=0.8, temperature=1.0)
    assert np.array_equal(idx.numpy(), Z["sampler/indices"])
    assert np.allclose(probs.numpy(), Z["sampler/probs"], rtol=1e-6, atol=1e-8)
    torch.manual_seed(6101)                                      # same CPU generator stream as the fixture
    picks = [int(top_p_sampling(lg, top_k=100, top_p=0.8, temperature=1.0)) for _ in range(8)]
    assert picks == Z["sampler/picks_seed6101"].tolist()


def test_generate_greedy_equals_reference_shaped_loop():
    from chatglm_q_amd.decoder import ChatGLMDecoder
    model, cfg = build("f32")
    prefix = Z["f32/ids"][0].tolist()
    # reference-shaped loop: full forward with concatenated cache, argmax
    with torch.no_grad():
        ids = torch.tensor([prefix])
        _, logits, kv = model(input_ids=ids)
        want = [int(logits[0, -1].argmax())]
        for _ in range(9):
            _, logits, kv = model(input_ids=torch.tensor([[want[-1]]]), past_key_values=kv)
            want.append(int(logits[0, -1].argmax()))
    dec = ChatGLMDecoder(None, model)
    got = list(dec.generate_ids(prefix, max_generated_tokens=10, greedy=True, ignore_eos=True))
    assert got == want
    got_chunked = list(dec.generate_ids(prefix, max_generated_tokens=10, greedy=True, ignore_eos=True, prefill_chunk=5))
    assert got_chunked == want
    s = dec.last_stats
    assert s["prefix"] == len(prefix) and s["generated"] == 10 and s["gen_tok_per_s"] > 0
    # EOS stops the loop
    stop = next(i for i, t in enumerate(want) if i > 0 and t not in want[:i])
    dec_eos = ChatGLMDecoder(None, model, eos_token_id=want[stop])
    assert list(dec_eos.generate_ids(prefix, max_generated_tokens=10, greedy=True)) == want[:stop + 1]


def test_reference_training_contract_labels_loss_backward():
    """chatglm_q/model.py:384-390: forward(labels=...) returns a loss that backpropagates (P-tuning / LoRA through the
    frozen quantized weights).  The inference shortcuts of the attention core (bmm(out=), fused softmax launch) must
    step aside when autograd is recording."""
    model, cfg = build("f32")
    model.train()
    ids = torch.from_numpy(Z["f32/ids"])
    loss, logits, kv = model(input_ids=ids, labels=ids)
    assert loss.requires_grad and torch.isfinite(loss)
    loss.backward()
    g = model.layers[0].attn_ln.weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().sum() > 0
    emb = model.word_embedding(ids).detach().requires_grad_(True)
    loss2, _, _ = model(input_embeddings=emb, labels=ids)
    loss2.backward()
    assert emb.grad is not None and emb.grad.abs().sum() > 0
    with torch.no_grad():
        _, ref_logits, _ = model(input_ids=ids)
    assert torch.allclose(logits, ref_logits, atol=1e-5, rtol=1e-5)      # same function with and without the graph


def test_decoder_reuses_its_session_across_generations():
    """every generate_ids() used to allocate a new cache (and re-capture a HIP graph on the GPU).  One session
    is now kept while the capacity fits; a second, different prompt through the reused session equals a fresh decoder."""
    from chatglm_q_amd.decoder import ChatGLMDecoder
    model, cfg = build("f32")
    p1 = Z["f32/ids"][0].tolist()
    p2 = [(7 * t + 3) % cfg.vocab_size for t in range(5)]
    dec = ChatGLMDecoder(None, model)
    a1 = list(dec.generate_ids(p1, max_generated_tokens=6, greedy=True, ignore_eos=True))
    sess = dec._session
    a2 = list(dec.generate_ids(p2, max_generated_tokens=6, greedy=True, ignore_eos=True))
    assert dec._session is sess                                   # reused, not rebuilt
    fresh = list(ChatGLMDecoder(None, model).generate_ids(p2, max_generated_tokens=6, greedy=True, ignore_eos=True))
    assert a2 == fresh and len(a1) == 6
    sess.capacity = 8                                             # pretend the kept session is too small for the next request
    list(dec.generate_ids(p1, max_generated_tokens=6, greedy=True, ignore_eos=True))
    assert dec._session is not sess and dec._session.capacity >= len(p1) + 6
