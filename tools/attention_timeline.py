#!/usr/bin/env python3
"""Developer tool: where the decode attention launch (ChatGLM2 geometry, capacity 256: two workgroups of 8 waves) spends its time.
Reads the QL_ATT_STAMPS build of decode_ops.hip (tools/ab/build_variant.sh attstamps decode_ops.hip -DQL_ATT_STAMPS;
QLINEAR_LIB_PATH=tools/ab/libqlinear_hip_attstamps.so): the 100 MHz wall clock at the stations of the kernel's dependent chain, first and
last wave of workgroup 0, beside the launch-to-launch time of the same kernel over 28 rotating caches in one HIP graph."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chatglm_q_amd import _lib, fused_ops as F_
from chatglm_q_amd import model as M

dev = torch.device("cuda:0")
B, H, G, D, L = 1, 32, 2, 128, 28
cap = int(os.environ.get("CAP", "256"))
n = cap - 20
qkvs = [torch.randn(B, 1, (H + 2 * G) * D, device=dev).half() for _ in range(L)]
table = M.rotary_table(D, cap + 8).to(dev).half().reshape(cap + 8, -1).contiguous()
pos = torch.full((B, 1), n + 1, dtype=torch.long, device=dev)
widx = torch.tensor([n], dtype=torch.long, device=dev)
mask = torch.full((B, 1, cap), -1e10, device=dev)
mask[:, :, : n + 1] = 0
caches = [(torch.randn(B, cap, G, D, device=dev).half(), torch.randn(B, cap, G, D, device=dev).half()) for _ in range(L)]
evict = torch.empty(512 << 20, dtype=torch.uint8, device=dev)          # rotates the caches out of L2 / MALL between replays


def run():
    for x, (k, v) in zip(qkvs, caches):
        F_.decode_attention_rope(x, table, pos, widx, k, v, mask, H, G, D)


run()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    run()
def read_stamps(lib):
    buf = np.zeros((2, 16), dtype=np.uint64)
    lib.qlinear_att_stamps_read.argtypes = [ctypes.c_void_p]
    assert lib.qlinear_att_stamps_read(buf.ctypes.data) == 0
    names = ["entry", "loads issued", "loads landed", "rotary done", "behind barrier 1", "K Q^T done", "softmax done", "P V done",
             "partials in LDS", "behind barrier 2", "stores issued", "stores acknowledged"]
    print(f"  the last wave entered {(int(buf[1, 0]) - int(buf[0, 0])) * 0.01:+.2f} us after the first; stations of the last wave on the first wave's clock: "
          + ", ".join(f"{(int(buf[1, i]) - int(buf[0, 0])) * 0.01:.2f}" for i in (15, 12, 2, 3, 5, 6, 7, 9, 11)) + " (loads issued, first landed, all landed, rotary, scores, softmax, PV, barrier 2, end)")
    for w, label in ((0, "first wave"), (1, "last wave")):
        t = buf[w, :12].astype(np.int64)
        print(f"  {label}: " + ", ".join(f"{nm} +{(t[i] - t[i - 1]) * 0.01:.2f}" for i, nm in enumerate(names) if i > 0)
              + f" | q / k / v loads issued +{(int(buf[w, 15]) - t[0]) * 0.01:.2f}, first query load landed +{(int(buf[w, 12]) - t[0]) * 0.01:.2f} after entry | entry -> end {(t[11] - t[0]) * 0.01:.2f} us at "
              f"{(int(buf[w, 14]) - int(buf[w, 13])) / max((t[11] - t[0]) * 0.01, 1e-9) / 1e3:.2f} GHz")


if os.environ.get("IN_DECODE") == "1":               # the last attention launch of a real graph-replayed decode (28 layers, the GEMVs between)
    import bench_extras
    from chatglm_q_amd.decoder import ChatGLMDecoder
    model, cfg = bench_extras._chatglm2_6b(torch, dev, torch.float16)
    for m in model.modules():
        if hasattr(m, "prepare"):
            m.prepare()
    dec = ChatGLMDecoder(None, model)
    prompt = [(37 * i + 11) % cfg.vocab_size for i in range(32)]
    list(dec.generate_ids(prompt, max_generated_tokens=64, greedy=True, ignore_eos=True, use_graph=True))
    print("inside a graph-replayed decode step (layer 28's attention launch of the last token):", dec.last_stats)
    read_stamps(_lib.get_lib())
    sys.exit(0)

for cold in (False, True):
    ts = []
    for _ in range(10):
        if cold:
            evict.add_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / L * 1e3)
    print(f"capacity {cap}: launch to launch {np.median(ts):.2f} us per attention launch ({'operands evicted between replays' if cold else 'operands cache-hot'})")
    lib = _lib.get_lib()
    if hasattr(lib, "qlinear_att_stamps_read"):
        read_stamps(lib)
