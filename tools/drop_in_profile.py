#!/usr/bin/env python3
"""Host-side profile of the drop-in mode (reference-shaped forward(), eager QLinear launches): where the time of one
decode token goes.  Prints cProfile's top entries by own time and the per-token wall time."""
import cProfile
import io
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_extras as BX  # noqa: E402

dev = torch.device("cuda:0")
model, cfg = BX._chatglm2_6b(torch, dev, torch.float16)
prompt = [(37 * i + 11) % cfg.vocab_size for i in range(32)]
print("warm", BX.drop_in_generate(torch, model, prompt, 8))
pr = cProfile.Profile()
pr.enable()
res = BX.drop_in_generate(torch, model, prompt, 24)
pr.disable()
print(res)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
