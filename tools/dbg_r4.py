import sys, torch
sys.path.insert(0, '.')
from chatglm_q_amd.int8 import hip_ops as h8
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for (M, K, N) in [(256, 256, 256)]:
    w = torch.randint(-127, 128, (N, K), dtype=torch.int8, generator=g)
    sc = (torch.rand(N, generator=g) * 0.01 + 0.001).half()
    a = torch.randn((M, K), generator=g).half()
    tiled = h8.tile_w8(w.to(dev))
    a_q, a_s = h8.act_quant_rowwise(a.to(dev))
    for bias in (None, (torch.randn(N, generator=g) * 0.1).half().to(dev)):
        o1 = h8.w8a8_gemm256(a_q, a_s, tiled, N, sc.to(dev), bias).float().cpu()
        o2 = h8.w8a8_gemm_tiled(a_q, a_s, tiled, N, sc.to(dev), bias).float().cpu()
        acc = (a_q.cpu().double() @ w.double().t())
        ref = (acc * (a_s.cpu().double()[:, None] * sc.double()[None, :]))
        bad = (o1 != o2).nonzero()
        print("bias" if bias is not None else "nobias", "diffs", len(bad), "of", M * N)
        for i, j in bad[:10].tolist():
            f32 = torch.tensor(acc[i, j].item(), dtype=torch.float32) * (a_s[i].cpu() * sc[j].float())
            print(i, j, "r4", o1[i, j].item(), "old", o2[i, j].item(), "exact", ref[i, j].item(), "f32 formula", f32.item(), "->f16", f32.half().item())
