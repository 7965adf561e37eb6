"""Developer tool: batch-1 greedy decode speed of the synthetic ChatGLM2-6B int4g32 model at long contexts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras
from chatglm_q_amd.decoder import DecodeSession
dev = torch.device("cuda:0")
model, cfg = bench_extras._chatglm2_6b(torch, dev, torch.float16)
for m in model.modules():
    if hasattr(m, "prepare"):
        m.prepare()
for ctx in [int(x) for x in os.environ.get("CONTEXTS", "128,1024,4096,8000").split(",")]:
    cap = -(-(ctx + 72) // 64) * 64
    ids = torch.randint(0, cfg.vocab_size, (1, ctx), device=dev)
    sess = DecodeSession(model, 1, cap, use_graph=True)
    logits = sess.prefill(ids, 512)
    sess.tok.copy_(logits.argmax(-1, keepdim=True))
    sess.capture(greedy=True)
    sess.decode_step(greedy=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(48):
        sess.decode_step(greedy=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 48
    print(f"context {ctx:5d} (capacity {cap}): {dt * 1e3:.3f} ms/token  {1 / dt:.1f} tok/s  (split from {os.environ.get('QLINEAR_SPLIT_ATTENTION_FROM', '448')})", flush=True)
    del sess
