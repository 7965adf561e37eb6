"""Round-3 reference pins (tests/golden/model_r3.npz, make_golden.py::gen_model_r3) through the CPU branch of the modules:
the int8 model at a tiny config (fp32 / fp16) and at real layer dimensions, the bf16 int4g32 model.  The GPU twin
(test_model_r3_gpu.py) adds the 320-position context; its builders live here."""
import numpy as np
import pytest
import torch

import _golden as G
from oracle import qlinear_oracle as O
from chatglm_q_amd import model as M
from chatglm_q_amd.decoder import DecodeSession

R3 = G.load("model_r3.npz")
TDT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def dec(name):
    """fixture array as float32 (bf16 entries are stored as bit patterns)."""
    if name + "_bf16bits" in R3.files:
        return O.bf16_bits_to_f32(R3[name + "_bf16bits"])
    return R3[name].astype(np.float32)


def t2n(t):
    return t.detach().float().cpu().numpy()


def build_r3(tag, device="cpu"):
    """The build's model for a fixture case, filled by the same seeded generators the reference model was filled with."""
    if tag.startswith("i8tiny/"):
        cfg = M.ChatGLM2Config(**G.TINY_INT8_CONFIG)
        model = M.create_quant_int8_model(cfg, dtype=TDT[tag.split("/")[1]])
        G.fill_seeded_int8_(model.state_dict(), seed=8400)
    elif tag == "i8":
        cfg = M.ChatGLM2Config(**G.REAL_DIM_CONFIG)
        model = M.create_quant_int8_model(cfg, dtype=torch.float16)
        G.fill_seeded_int8_(model.state_dict())
    else:
        cfg = M.ChatGLM2Config(**G.REAL_DIM_CONFIG)
        model = M.create_quant_int4_model(cfg, dtype=torch.bfloat16 if tag == "bf16" else torch.float16)
        G.fill_seeded_(model.state_dict())
    for m in model.modules():
        if hasattr(m, "invalidate"):
            m.invalidate()
    return model.to(device).eval(), cfg


@pytest.mark.parametrize("dt", ["f32", "f16"])
def test_int8_tiny_model_matches_reference(dt):
    model, cfg = build_r3(f"i8tiny/{dt}")
    p = f"i8tiny/{dt}/"
    ids, nxt = torch.from_numpy(R3[p + "ids"]), torch.from_numpy(R3[p + "next_id"])
    tol = {"f32": 2e-5, "f16": 2e-3}[dt]
    with torch.no_grad():
        _, logits, kv = model(input_ids=ids)
        assert O.rel_l2(t2n(logits), dec(p + "prefill_logits")) < tol
        _, logits2, kv2 = model(input_ids=nxt, past_key_values=kv)
        assert O.rel_l2(t2n(logits2), dec(p + "decode_logits")) < tol
        assert O.rel_l2(t2n(kv2[0][0]), dec(p + "kv0_k")) < tol and O.rel_l2(t2n(kv2[0][1]), dec(p + "kv0_v")) < tol
        sess = DecodeSession(model, 1, 32, use_graph=False)
        last = sess.prefill(ids, chunk=7)
        assert O.rel_l2(t2n(last), dec(p + "chunked_last_logits")) < tol


@pytest.mark.parametrize("tag,tol", [("i8", 2e-3), ("bf16", 1.6e-2)])
def test_real_dim_b1_matches_reference(tag, tol):
    torch.set_num_threads(8)
    model, cfg = build_r3(tag)
    ids = torch.from_numpy(R3[f"{tag}/ids"].astype(np.int64))
    nxt = R3[f"{tag}/next_ids"]
    with torch.no_grad():
        _, logits, kv = model(input_ids=ids)
        assert O.rel_l2(t2n(logits[:, -1]), dec(f"{tag}/prefill_last_logits")) < tol
        for t in range(3):
            _, lg, kv = model(input_ids=torch.from_numpy(nxt[t:t + 1]).view(1, 1), past_key_values=kv)
            assert O.rel_l2(t2n(lg[:, -1]), dec(f"{tag}/decode_logits_{t}")) < tol
