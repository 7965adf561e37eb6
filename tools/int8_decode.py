"""Developer tool: greedy decode of the synthetic ChatGLM2-6B with int8 per-channel weights (weight-only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chatglm_q_amd import model as M
from chatglm_q_amd.decoder import ChatGLMDecoder

dev = torch.device("cuda:0")
cfg = M.ChatGLM2Config()
with torch.device(dev):
    m = M.create_quant_int8_model(cfg, dtype=torch.float16)
M.fill_synthetic_(m, 0)
m.eval()
dec = ChatGLMDecoder(None, m)
prompt = [(37 * i + 11) % cfg.vocab_size for i in range(32)]
for label, kw in (("graph", dict(use_graph=True, sync_every_token=True)), ("device_loop", dict(use_graph=True, sync_every_token=False))):
    toks = list(dec.generate_ids(prompt, max_generated_tokens=96, greedy=True, ignore_eos=True, **kw))
    print(label, {k: round(v, 4) if isinstance(v, float) else v for k, v in dec.last_stats.items()})
