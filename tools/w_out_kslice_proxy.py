"""Round 6, VERDICT r5 "Next 5" priced before it is built.  A grid-level K split of w_out (13696 -> 4096) by S = 4 - workgroup = 4 column
quads x ONE K slice, staging 6.8 KB of the row instead of all 27 KB - has exactly the weight stream, the staging pattern and the workgroup
count of a plain GEMV of shape 3424 -> 16384 (one "column" per (column, slice)); only the output differs (fp32 partial rows instead of
fp16).  So that shape, on the shipped kernel, is the optimistic stand-in for the producer half of the scheme; S = 2: 6848 -> 8192.
Interleaved with the real w_out under bench_extras' protocol (weight sets beyond the memory-side cache, graph replay)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras as BE

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(9)
shapes = [("w_out 13696->4096 (shipped, KS=4 inside the workgroup)", 13696, 4096),
          ("S=2 stand-in 6848->8192", 6848, 8192), ("S=4 stand-in 3424->16384", 3424, 16384)]
res = {n: [] for n, _, _ in shapes}
for rnd in range(3):
    for name, K, N in shapes:
        per = K * N // 2 + (K // 32) * N * 2
        n_sets = 24
        layers = [BE._w4_layer(torch, dev, K, N, False, gen) for _ in range(n_sets)]
        x = torch.randn(1, K, device=dev, dtype=torch.float16)

        def fn():
            with torch.no_grad():
                for _ in range(4):
                    for l in layers:
                        l(x)

        ms = BE._graph_time(torch, dev, fn) / (4 * n_sets)
        res[name].append(round(ms * 1e3, 3))
        del layers
        torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
