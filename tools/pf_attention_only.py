"""The one-launch prefill attention alone, config 5's layer: batch 4 x 2048 positions, causal, fp16 (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatglm_q_amd import fused_ops as F_

H, G, D, B, S = 32, 2, 128, 4, 2048
dev, dt = "cuda", torch.float16
k = torch.randn(B, S, G, D, device=dev).to(dt)
v = torch.randn(B, S, G, D, device=dev).to(dt)
q = torch.randn(B, S, H * D, device=dev).to(dt)
t = torch.arange(S, device=dev)
mask = ((t[None, None, :] > t[None, :, None]).expand(B, S, S).float() * -1e10).contiguous()
flags = F_.attention_tile_flags(mask)
for _ in range(int(os.environ.get("REPS", 12))):
    F_.prefill_attention(q, k, v, mask, flags, S, H, G, D)
torch.cuda.synchronize()
print("done")
