"""Developer tool: chunked prefill (batch 4 x 2048, one pass of 8192 rows) of the synthetic ChatGLM2-6B with int8 per-channel weights:
weight-only (the reference's int8 forward) beside int8 ACTIVATIONS (module.act_quant = True: row-wise act-quant by the norm /
SiLU * gate launches, int8 x int8 MFMA GEMMs - chatglm_q/int8/qlinear.py:56-62)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from chatglm_q_amd import model as M  # noqa: E402
from chatglm_q_amd.decoder import DecodeSession  # noqa: E402


def run(torch, dev, B=4, S=2048):
    cfg = M.ChatGLM2Config()
    with torch.device(dev):
        m = M.create_quant_int8_model(cfg, dtype=torch.float16)
    M.fill_synthetic_(m, 0)
    m.eval()
    ids = torch.randint(0, cfg.vocab_size, (B, S), device=dev)
    out = {}
    for label, aq in (("weight_only", None), ("int8_activations", True)):
        for mod in m.modules():
            if hasattr(mod, "act_quant") and hasattr(mod, "weight_scale") and mod is not m.lm_head:
                mod.act_quant = aq
        logits = None
        for rep in range(2):                         # first pass: lazy layouts, allocator
            sess = DecodeSession(m, B, S, use_graph=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            logits = sess.prefill(ids, S)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out[label] = {"seconds": round(dt, 4), "tokens_per_s": round(B * S / dt, 1)}
        out[label + "_logits"] = logits.float()
    a, b = out.pop("weight_only_logits"), out.pop("int8_activations_logits")
    out["rel_l2_between_paths"] = float(((a - b).norm() / a.norm()).item())
    return out


if __name__ == "__main__":
    print(run(torch, torch.device("cuda:0")))
