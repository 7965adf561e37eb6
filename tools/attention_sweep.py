"""Developer tool: time the decode attention launch(es) alone (ChatGLM2 geometry: 32 heads, 2 key/value groups, D = 128)
over 28 rotating caches inside one HIP graph, for several capacities and both split-mode kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chatglm_q_amd import fused_ops as F_
from chatglm_q_amd import model as M

dev = torch.device("cuda:0")
B = int(os.environ.get("BATCH", "1"))
H, G, D, L = 32, 2, 128, 28
for cap in [int(x) for x in os.environ.get("CAPS", "256,512,1152,4224,8128").split(",")]:
    n = cap - 20
    qkv = torch.randn(B, 1, (H + 2 * G) * D, device=dev).half()
    table = M.rotary_table(D, cap + 8).to(dev).half().reshape(cap + 8, -1).contiguous()
    pos = torch.full((B, 1), n + 1, dtype=torch.long, device=dev)
    widx = torch.tensor([n], dtype=torch.long, device=dev)
    mask = torch.full((B, 1, cap), -1e10, device=dev)
    mask[:, :, : n + 1] = 0
    caches = [(torch.randn(B, cap, G, D, device=dev).half(), torch.randn(B, cap, G, D, device=dev).half()) for _ in range(L)]
    line = f"capacity {cap:5d} batch {B}:"
    for name, split, gqa in (("one block/head", False, "0"), ("split per head", True, "0"), ("group kernel (MFMA)", cap > 256, "1")):
        os.environ["QLINEAR_ATTENTION_MFMA"] = gqa
        def run():
            for k, v in caches:
                F_.decode_attention_rope(qkv, table, pos, widx, k, v, mask, H, G, D, split=split)
        run()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            run()
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        line += f"  {name} {e0.elapsed_time(e1) / 20 / L * 1e3:7.2f} us"
    print(line, flush=True)
