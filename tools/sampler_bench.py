"""Per-launch time of qlinear_top_p_sample (csrc/sampler.hip) beside qlinear_greedy_advance on logits rows of several shapes: graph of 200
dependent launches, HIP events.  Rows: seeded normal logits (tests/_sampler_cases.py recipes) and the logits a synthetic ChatGLM2-6B emits."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from chatglm_q_amd import fused_ops
import _sampler_cases as SC

dev = torch.device("cuda:0")


def timed(fn, n=200, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        best = 1e9
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s); g.replay(); b.record(s); b.synchronize()
            best = min(best, a.elapsed_time(b) / n * 1e3)
    return round(best, 2)


out = {}
rows = {}
for name in ("vocab_default", "vocab_peaked", "vocab_flat_k256", "vocab_bf16", "ties_coarse", "masked_inf", "sorted_ascending", "k1024", "small_n"):
    dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[SC.CASES[name][3]]
    rows[name] = (torch.from_numpy(SC.logits_for(name)).to(dt).to(dev)[None].contiguous(),) + tuple(SC.CASES[name][4:])
if "--model" in sys.argv:
    import bench_extras
    model, cfg = bench_extras._chatglm2_6b(torch, dev, torch.float16)
    from chatglm_q_amd.decoder import DecodeSession
    sess = DecodeSession(model, 1, 64, use_graph=False)
    lg = sess.prefill(torch.tensor([[(37 * i + 11) % cfg.vocab_size for i in range(32)]], device=dev))
    rows["model_logits"] = (lg.clone().contiguous(), 100, 0.8, 1.0)
    x = lg.float()
    out["model_logits_stats"] = {"std": float(x.std()), "max": float(x.max()), "distinct": int(torch.unique(x).numel())}
    del model, sess
for name, (lg, k, p, T) in rows.items():
    B, N = lg.shape
    tok = torch.zeros(B, 1, dtype=torch.int64, device=dev)
    wi = torch.zeros(1, dtype=torch.int64, device=dev)
    pos = torch.zeros(B, 1, dtype=torch.int64, device=dev)
    mask = torch.zeros(B, 1, 4096, dtype=torch.float32, device=dev)
    rng = fused_ops.new_rng_state(B, 5, dev)

    def samp():
        wi.zero_()
        fused_ops.top_p_sample(lg, tok.view(-1), rng, k, p, T, write_index=wi, pos=pos.view(-1), mask=mask)

    def greedy():
        wi.zero_()
        fused_ops.greedy_advance(lg, tok, wi, pos, mask)

    def zero():
        wi.zero_()

    z = timed(zero)
    out[name] = {"N": N, "top_k": k, "sample_us": round(timed(samp) - z, 2), "greedy_us": round(timed(greedy) - z, 2)}
print(json.dumps(out, indent=1))
