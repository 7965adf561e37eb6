"""Phase timeline of the sampler's one workgroup (make -C chatglm_q_amd/csrc strace; QLINEAR_LIB_PATH=.../libqlinear_hip_strace.so):
thread 0 stamps the 100 MHz clock at the phase boundaries; median over launches, microseconds from kernel entry."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("QLINEAR_LIB_PATH", os.path.join(ROOT, "chatglm_q_amd", "csrc", "libqlinear_hip_strace.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from chatglm_q_amd import _lib
import _sampler_cases as SC

dev = torch.device("cuda:0")
lib = _lib.get_lib()
names = ["entry", "pass1+t0", "pass2+Z", "(slow path)", "select", "rank", "final wave"]
out = {}
for name in sys.argv[1:] or ["vocab_default", "vocab_flat_k256", "small_n", "k1024"]:
    dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[SC.CASES[name][3]]
    lg = torch.from_numpy(SC.logits_for(name)).to(dt).to(dev)[None].contiguous()
    k, p, T = SC.CASES[name][4:]
    tok = torch.zeros(1, dtype=torch.int64, device=dev)
    rng = torch.zeros(2, dtype=torch.int64, device=dev)
    u = torch.zeros(64, dtype=torch.float32, device=dev)
    rows = []
    for it in range(40):
        st = lib.qlinear_top_p_sample(lg.data_ptr(), 1, lg.shape[1], lg.shape[1], k, p, T, None, rng.data_ptr(), tok.data_ptr(), None, None, None, 0,
                                      None, None, u.data_ptr(), 0, _lib.dtype_code(lg.dtype), _lib.stream_ptr(dev))
        assert st == 0, st
        torch.cuda.synchronize()
        stamps = u.cpu().numpy().view(np.uint64)[1:8].astype(np.int64)
        rows.append((stamps - stamps[0]) / 100.0)
    med = np.median(np.array(rows[5:]), axis=0)
    out[name] = {n: round(float(v), 2) for n, v in zip(names, med)}
print(json.dumps(out, indent=1))
