"""Developer tool: aggregate greedy decode throughput of the synthetic ChatGLM2-6B int4g32 model for several batch
sizes (one DecodeSession per batch size, graph-replayed step, 32-token prompts, 64 steps)."""
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
steps = 64
for B in [int(x) for x in os.environ.get("BATCHES", "1,2,4,8,16,32,64").split(",")]:
    ids = torch.randint(0, cfg.vocab_size, (B, 32), device=dev)
    sess = DecodeSession(model, B, 128, use_graph=True)
    logits = sess.prefill(ids)
    sess.tok.copy_(logits.argmax(-1, keepdim=True))
    sess.capture(greedy=True)
    sess.decode_step(greedy=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sess.decode_step(greedy=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"batch {B:3d}: {dt / steps * 1e3:7.3f} ms/step  {B * steps / dt:9.1f} tok/s aggregate", flush=True)
    del sess
