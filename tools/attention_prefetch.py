"""Developer tool: the decode attention launch followed by the attention output projection (int4g32 4096 -> 4096, one
row), 28 rotating layers in one HIP graph, with and without the prefetch workgroups (QLINEAR_ATTENTION_PREFETCH is
read per call here)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras
from chatglm_q_amd import _lib, fused_ops as F_
from chatglm_q_amd import model as M

dev = torch.device("cuda:0")
B, H, G, D, L = 1, 32, 2, 128, 28
gen = torch.Generator(device=dev).manual_seed(1)
layers = [bench_extras._w4_layer(torch, dev, 4096, 4096, False, gen) for _ in range(L)]
big = [bench_extras._w4_layer(torch, dev, 4096, 27392, False, gen) for _ in range(6)]   # evicts the caches between graphs
for cap in (256, 1152, 4224):
    n = cap - 20
    qkv = torch.randn(B, 1, (H + 2 * G) * D, device=dev).half()
    table = M.rotary_table(D, cap + 8).to(dev).half().reshape(cap + 8, -1).contiguous()
    pos = torch.full((B, 1), n + 1, dtype=torch.long, device=dev)
    widx = torch.tensor([n], dtype=torch.long, device=dev)
    mask = torch.full((B, 1, cap), -1e10, device=dev)
    mask[:, :, : n + 1] = 0
    caches = [(torch.randn(B, cap, G, D, device=dev).half(), torch.randn(B, cap, G, D, device=dev).half()) for _ in range(L)]
    line = f"capacity {cap:5d}:"
    for name, pf, with_o in (("attention", False, False), ("attention+pf", True, False), ("attention, o_proj", False, True),
                             ("attention+pf, o_proj", True, True)):
        F_.PREFETCH_NEXT = pf
        def run():
            for (k, v), l in zip(caches, layers):
                nxt = (l.prepare()._packed, _lib.NEXT_W4G32_PACKED, 4096, 4096)
                a = F_.decode_attention_rope(qkv, table, pos, widx, k, v, mask, H, G, D, prefetch=nxt)
                if with_o:
                    with torch.no_grad():
                        l(a)
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
        line += f"  {name} {e0.elapsed_time(e1) / 20 / L * 1e3:6.2f} us"
    print(line, flush=True)
