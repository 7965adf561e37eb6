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
