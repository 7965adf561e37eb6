"""A/B of a library build on the whole decode step: greedy generate() of the synthetic ChatGLM2-6B int4g32, graph-replayed, 3 x 128 tokens
(QLINEAR_LIB_PATH picks the build; one process per build, interleave the processes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras
from chatglm_q_amd.decoder import ChatGLMDecoder

dev = torch.device("cuda:0")
model, cfg = bench_extras._chatglm2_6b(torch, dev, torch.float16)
dec = ChatGLMDecoder(None, model)
prompt = [(37 * i + 11) % cfg.vocab_size for i in range(32)]
list(dec.generate_ids(prompt, max_generated_tokens=8, greedy=True, ignore_eos=True, use_graph=True))
res = []
for _ in range(3):
    toks = list(dec.generate_ids(prompt, max_generated_tokens=128, greedy=True, ignore_eos=True, use_graph=True))
    res.append(round(dec.last_stats["gen_tok_per_s"], 1))
print(os.path.basename(os.environ.get("QLINEAR_LIB_PATH", "product")), res, "tokens", toks[:6])
