import sys, os
sys.path.insert(0, "/root/repo")
import torch, bench_extras
from chatglm_q_amd import decoder as Dm
dev = torch.device("cuda:0")
model, cfg = bench_extras._chatglm2_6b(torch, dev, torch.float16)
for m in model.modules():
    if hasattr(m, "prepare"):
        m.prepare()
dec = Dm.ChatGLMDecoder(None, model)
prompt = [(37 * i + 11) % cfg.vocab_size for i in range(32)]
for rep in range(4):
    for flag in (True, False):
        Dm.AHEAD_LAUNCH = flag
        list(dec.generate_ids(prompt, max_generated_tokens=200, greedy=True, ignore_eos=True, use_graph=True))
        print("ahead" if flag else "plain", round(dec.last_stats["gen_tok_per_s"], 1), flush=True)
for rep in range(3):                                       # upper bound: no read-back at all until the end (sync_every_token=False)
    list(dec.generate_ids(prompt, max_generated_tokens=200, greedy=True, ignore_eos=True, use_graph=True, sync_every_token=False))
    print("device loop (read back at the end)", round(dec.last_stats["gen_tok_per_s"], 1), flush=True)
