"""Developer tool: where the one-row int4g32 GEMV's time goes, per layer shape.  Needs the developer build
(`make -C chatglm_q_amd/csrc dev`, then copy libqlinear_hip_dev.so over libqlinear_hip.so in a scratch copy) and
QL_VARIANT: 0 product kernel, 1 loads only (no math), 2 constant activation (no LDS reads), 3 activations from
global, 4 no activation staging at all."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_extras
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
for K, N, NL in ((4096, 4608, 48), (4096, 4096, 48), (4096, 27392, 8), (13696, 4096, 16)):
    layers = [bench_extras._w4_layer(torch, dev, K, N, False, gen) for _ in range(NL)]
    a = torch.randn(1, K, device=dev, dtype=torch.float16)
    def f():
        with torch.no_grad():
            for l in layers:
                l(a)
    print(f"QL_VARIANT={os.environ.get('QL_VARIANT', '0')} KSPLIT={os.environ.get('QLINEAR_W4_KSPLIT', 'auto')} {K}->{N}: {bench_extras._graph_time(torch, dev, f) / NL * 1e3:.2f} us", flush=True)
    del layers
