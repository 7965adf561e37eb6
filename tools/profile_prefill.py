"""Developer tool: one chunked prefill (2048 x 4, chunks of PREFILL_CHUNK = 2048) of the synthetic ChatGLM2-6B int4g32 model (for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras
from chatglm_q_amd.decoder import DecodeSession

dev = torch.device("cuda:0")
model, cfg = bench_extras._chatglm2_6b(torch, dev, torch.float16)
B, S, CH = 4, 2048, int(os.environ.get("PREFILL_CHUNK", 2048))
ids = torch.randint(0, cfg.vocab_size, (B, S), device=dev)
sess = DecodeSession(model, B, S, use_graph=False)
sess.prefill(ids[:, :CH], CH)
torch.cuda.synchronize()
sess = DecodeSession(model, B, S, use_graph=False)
sess.prefill(ids, CH)
torch.cuda.synchronize()
