"""Developer experiment: what the per-replay cost of the one-step decode graph is - the same greedy step captured N times per
graph (tokens of the intermediate steps are not logged here: timing only) against one step per graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras
from chatglm_q_amd.decoder import DecodeSession

dev = torch.device("cuda:0")
model, cfg = bench_extras._chatglm2_6b(torch, dev, torch.float16)
ids = torch.randint(0, cfg.vocab_size, (1, 32), device=dev)
for N in (1, 2, 4, 8):
    sess = DecodeSession(model, 1, 256, use_graph=True)
    sess.prefill(ids)
    sess.capture(greedy=True)                       # warm-up + layouts
    g = torch.cuda.CUDAGraph()
    saved = (sess.tok.clone(), sess.write_index.clone(), sess.pos.clone(), sess.mask.clone())
    with torch.cuda.graph(g):
        for _ in range(N):
            sess._step_body(True)
    for dst, src in zip((sess.tok, sess.write_index, sess.pos, sess.mask), saved):
        dst.copy_(src)
    reps = 128 // N
    g.replay()
    for dst, src in zip((sess.tok, sess.write_index, sess.pos, sess.mask), saved):
        dst.copy_(src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{N} steps per graph: {dt / (reps * N) * 1e3:.4f} ms per token, {reps * N / dt:.1f} tok/s", flush=True)
