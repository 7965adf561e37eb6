"""A/B harness for BASELINE config 5 on the whole prefill: ChatGLM2-6B int4g32 (synthetic), batch 4 x 2048 positions in one pass, 6 timed repetitions
(one process per setting - environment switches / QLINEAR_LIB_PATH - interleave the processes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras
from chatglm_q_amd.decoder import DecodeSession
from chatglm_q_amd import model as M

for kv in filter(None, os.environ.get("AB_SET", "").split(",")):      # e.g. AB_SET=GATED_PREFILL=0,RESIDUAL_PREFILL=0 (module attributes of model.py)
    name, val = kv.split("=")
    assert hasattr(M, name), name
    setattr(M, name, bool(int(val)))

dev = torch.device("cuda:0")
model, cfg = bench_extras._chatglm2_6b(torch, dev, torch.float16)
B, S = 4, 2048
CH = int(os.environ.get("PREFILL_CHUNK", 2048))
ids = torch.randint(0, cfg.vocab_size, (B, S), device=dev)
DecodeSession(model, B, S, use_graph=False).prefill(ids, CH)
torch.cuda.synchronize()
ts = []
for _ in range(6):
    sess = DecodeSession(model, B, S, use_graph=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sess.prefill(ids, CH)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
    del sess
ts.sort()
print(sys.argv[1] if len(sys.argv) > 1 else "default", "prefill 4 x 2048: min %.4f s  median %.4f s" % (ts[0], ts[len(ts) // 2]))
