"""Developer tool: run a short graph-replayed decode of the synthetic ChatGLM2-6B int4g32 model (for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras
from chatglm_q_amd.decoder import ChatGLMDecoder

dev = torch.device("cuda:0")
model, cfg = bench_extras._chatglm2_6b(torch, dev, torch.float16)
for m in model.modules():
    if hasattr(m, "prepare"):
        m.prepare()
dec = ChatGLMDecoder(None, model)
prompt = [(37 * i + 11) % cfg.vocab_size for i in range(32)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
greedy = "sampled" not in sys.argv[2:]                 # `sampled`: the reference's default mode (device sampler inside the graph)
toks = list(dec.generate_ids(prompt, max_generated_tokens=n, greedy=greedy, ignore_eos=True, use_graph=True, seed=1))
print("greedy" if greedy else "sampled", dec.last_stats)
