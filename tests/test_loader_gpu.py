"""N2 end to end on the GPU: a reference-format checkpoint folder -> ``load_model(device="cuda")`` -> DecodeSession
prefill + HIP-graph decode steps, against logits the REFERENCE model produced for the same seeded weights
(tests/golden/real_model.npz); then other weights loaded INTO THE SAME model object: the graph must be captured again and the
logits must follow.  The path under test is the reference's ``load_model`` (chatglm_q/loader.py:90-104: ``copy_`` of every tensor into
``state_dict()`` buffers) meeting this build's lazily derived layouts and captured launches (chatglm_q/decoder.py:51-58 is the caller)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import _golden as G  # noqa: E402
from oracle import qlinear_oracle as O  # noqa: E402
from chatglm_q_amd import loader as L  # noqa: E402
from chatglm_q_amd import model as M  # noqa: E402
from chatglm_q_amd.decoder import DecodeSession  # noqa: E402
from test_model_real_cpu import R, f32, t2n  # noqa: E402

DEV = "cuda:0"
TOL = 2e-3


def _folder(tmp_path, seed, name):
    cfg = M.ChatGLM2Config(**G.REAL_DIM_CONFIG)
    lc = L.ChatGLMLoadConfig(model_config=cfg, quant_type="int4g32", torch_dtype="float16")
    model = L.build_model(lc)
    G.fill_seeded_(model.state_dict(), seed)
    path = tmp_path / name
    L.save_model(path, lc, model, shard=True, max_shard_bytes=48 * 1024 * 1024)     # several shards (2 layers of real width: ~210 MB)
    assert len(lc.weight_files) > 2
    return path


def _run(sess, ids, next_ids, steps=3):
    last = sess.prefill(ids)
    sess.tok.fill_(int(next_ids[0]))
    sess.capture(greedy=False)
    outs = [last]
    for t in range(steps):
        outs.append(sess.decode_step(torch.from_numpy(next_ids[t:t + 1]).view(1, 1), greedy=False).clone())
    return outs


def test_checkpoint_folder_to_graph_decode_and_reload_into_same_model(tmp_path):
    seed = int(R["seed"][0])
    lc, model = L.load_model(_folder(tmp_path, seed, "a"), device=DEV)
    assert lc.quant_type == "int4g32" and next(model.parameters()).is_cuda
    model.eval()
    ids, nxt = torch.from_numpy(R["b1/ids"]), R["b1/next_ids"]
    sess = DecodeSession(model, 1, 64, use_graph=True)
    outs = _run(sess, ids, nxt)
    assert O.rel_l2(t2n(outs[0]), f32("b1/prefill_logits")[:, -1]) < TOL
    for t in range(3):
        assert O.rel_l2(t2n(outs[1 + t]), f32(f"b1/decode_logits_{t}")) < TOL
    graph_a = sess.graph
    assert graph_a is not None

    # other weights INTO THE SAME model object, the way the reference's loader fills a model: copy_ into state_dict() buffers
    other = L.load_model(_folder(tmp_path, seed + 1, "b"))[1]
    with torch.no_grad():
        state = model.state_dict()
        for k, v in other.state_dict().items():
            state[k].copy_(v.to(state[k].device))
    sess.reset()
    outs_b = _run(sess, ids, nxt)
    assert sess.graph is not None and sess.graph is not graph_a                  # captured again: the old launches read dropped layouts
    fresh = DecodeSession(other.to(DEV).eval(), 1, 64, use_graph=False)
    want_b = _run(fresh, ids, nxt)
    for got, want in zip(outs_b, want_b):
        assert torch.equal(got, want)                                               # same weights, same kernels: bit for bit
    assert O.rel_l2(t2n(outs_b[0]), f32("b1/prefill_logits")[:, -1]) > 0.1          # ... and no longer model A's logits


def test_low_footprint_mode_drops_the_canonical_copy_and_round_trips_byte_for_byte(tmp_path):
    """Round 5: ``DecodeSession(low_footprint=True)`` drops the canonical GPU buffers of every int4 module once
    the derived layouts exist; decode and prefill keep producing the same bits, ``state_dict()`` / ``save_model`` rebuild the
    canonical tensors from part 1 (the gate-interleaved part 1 for a first MLP projection whose plain copy a decode-only session
    released) byte for byte, a reload of the saved folder equals the original checkpoint (chatglm_q/loader.py:90-104)."""
    from chatglm_q_amd.int4.qlinear import DynamicQuantizeLinear as Q4
    from chatglm_q_amd.loader import load_model, save_model
    config, model = L.load_model(_folder(tmp_path, int(R["seed"][0]) + 7, "orig"))
    want = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    _, m = load_model(tmp_path / "orig", device=DEV)
    m.eval()
    ids = torch.randint(0, config.model_config.vocab_size, (1, 12), generator=torch.Generator().manual_seed(5))
    ref = DecodeSession(m, 1, 32, use_graph=True)
    ref_last = ref.prefill(ids).clone()
    ref.capture(greedy=True)
    ref_steps = [ref.decode_step().clone() for _ in range(3)]
    resident_before = sum(sum(q.derived_nbytes().values()) for q in m.modules() if isinstance(q, Q4))

    _, m2 = load_model(tmp_path / "orig", device=DEV)
    m2.eval()
    low = DecodeSession(m2, 1, 32, use_graph=True, decode_only=True, low_footprint=True)
    assert torch.equal(low.prefill(ids), ref_last)
    low.capture(greedy=True)
    for want_logits in ref_steps:
        assert torch.equal(low.decode_step(), want_logits)
    q4s = [q for q in m2.modules() if isinstance(q, Q4)]
    assert q4s and all(q.canonical_dropped and q.derived_nbytes()["canonical"] == 0 for q in q4s)
    resident_low = sum(sum(q.derived_nbytes().values()) for q in q4s)
    canonical = sum(v.numel() * v.element_size() for k, v in want.items() if k.endswith(("weight", "weight_scale")) and "ln" not in k
                    and "word_embedding" not in k)
    assert resident_low <= 1.1 * canonical < resident_before           # one copy of the weights instead of two to three and a half
    got = m2.state_dict()
    assert list(got.keys()) == list(want.keys())
    for k, v in want.items():
        assert torch.equal(got[k].cpu(), v), k                     # incl. modules whose only resident copy is gate-interleaved
    save_model(tmp_path / "low", config, m2)
    _, m3 = load_model(tmp_path / "low", device="cpu")
    for k, v in m3.state_dict().items():
        assert torch.equal(v, want[k]), k
    assert all(q.canonical_dropped for q in q4s)                   # serving the state_dict did not bring the copies back
    # a second prefill in low-footprint mode rebuilds what the decode-only session released, without the canonical copy
    low2 = DecodeSession(m2, 1, 32, use_graph=False, low_footprint=True)
    assert torch.equal(low2.prefill(ids), ref_last)
    # loading new weights into a dropped module: the buffers come back, the data is the new data
    q = q4s[0]
    sd = {k: v.clone() for k, v in q.state_dict().items()}
    sd["weight"] = (sd["weight"] ^ 0x11)
    q.load_state_dict(sd)
    assert not q.canonical_dropped and torch.equal(q.weight, sd["weight"])
    x = torch.randn(2, q.in_features, device=DEV, dtype=q.weight_scale.dtype)
    y = q(x)
    q.drop_canonical()
    assert torch.equal(q(x), y) and torch.equal(q.state_dict()["weight"], sd["weight"])
    assert q.to("cpu").weight.shape == sd["weight"].shape and not q.canonical_dropped and torch.equal(q.weight, sd["weight"].cpu())


def test_from_pretrained_on_a_gpu_is_low_footprint_and_round_trips(tmp_path):
    """Round 6 hygiene: ``ChatGLMDecoder.from_pretrained`` on a GPU defaults to the low-footprint mode - after a
    generation no int4 module holds its canonical buffers any more - and ``save_pretrained`` still writes the checkpoint byte for byte;
    ``low_footprint=False`` keeps the canonical copies; both decoders generate the same tokens (greedy and seeded sampling)."""
    from chatglm_q_amd.decoder import ChatGLMDecoder
    seed = int(R["seed"][0])
    folder = _folder(tmp_path, seed, "lf")
    dec = ChatGLMDecoder.from_pretrained(folder, device=DEV)
    assert dec.low_footprint is True
    full = ChatGLMDecoder.from_pretrained(folder, device=DEV, low_footprint=False)
    prompt = [3, 17, 200, 5, 77, 400, 9]
    kw = dict(max_generated_tokens=6, ignore_eos=True)
    a = list(dec.generate_ids(prompt, greedy=True, **kw))
    b = list(full.generate_ids(prompt, greedy=True, **kw))
    assert a == b
    assert list(dec.generate_ids(prompt, seed=5, **kw)) == list(full.generate_ids(prompt, seed=5, **kw))
    int4 = [m for m in dec.model.modules() if hasattr(m, "canonical_dropped")]
    assert int4 and all(m.canonical_dropped for m in int4)
    assert not any(m.canonical_dropped for m in full.model.modules() if hasattr(m, "canonical_dropped"))
    held = lambda d: sum(sum(m.derived_nbytes().values()) + (0 if m.canonical_dropped else m.weight.numel() + 2 * m.weight_scale.numel())  # noqa: E731
                         for m in d.model.modules() if hasattr(m, "canonical_dropped"))
    assert held(dec) < 0.8 * held(full)
    dec.save_pretrained(tmp_path / "again")
    lc2, back = L.load_model(tmp_path / "again", device="cpu")
    want = L.load_model(folder, device="cpu")[1].state_dict()
    for k, v in back.state_dict().items():
        assert torch.equal(v, want[k]), k
