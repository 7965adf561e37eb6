"""Developer tool: chunked prefill (2048 x 4) time against the chunk length (MODEL=int8: the int8 weight-only model)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras
from chatglm_q_amd.decoder import DecodeSession
dev = torch.device("cuda:0")
if os.environ.get("MODEL") == "int8":
    from chatglm_q_amd import model as M
    cfg = M.ChatGLM2Config()
    with torch.device(dev):
        model = M.create_quant_int8_model(cfg, dtype=torch.float16)
    M.fill_synthetic_(model, 0)
    model.eval()
else:
    model, cfg = bench_extras._chatglm2_6b(torch, dev, torch.float16)
B, S = 4, 2048
ids = torch.randint(0, cfg.vocab_size, (B, S), device=dev)
for CH in (int(c) for c in os.environ.get("CHUNKS", "256,512,1024,2048").split(",")):
    sess = DecodeSession(model, B, S, use_graph=False)
    sess.prefill(ids[:, :CH], CH)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        sess = DecodeSession(model, B, S, use_graph=False)
        t0 = time.perf_counter()
        sess.prefill(ids, CH)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"chunk {CH}: {best:.4f} s  {B * S / best:.0f} tok/s", flush=True)
