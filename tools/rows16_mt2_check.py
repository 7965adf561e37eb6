"""Developer tool: the two-row-tile form of w4_rows16_kernel (17..32 rows; developer library, QLINEAR_ROWS16_MAX=32) against the oracle-checked
few-row kernel (QLINEAR_DISPATCH=norows16): relative L2 difference and time per call at the o_proj / qkv_proj shapes."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import bench_extras
    dev = torch.device("cuda:0")
    for K, N, NL in ((4096, 4096, 24), (4096, 4608, 24)):
        gen = torch.Generator(device=dev).manual_seed(1)
        layers = [bench_extras._w4_layer(torch, dev, K, N, True, gen) for _ in range(NL)]
        row = []
        for M in (17, 24, 32):
            a = torch.randn(M, K, device=dev, dtype=torch.float16, generator=torch.Generator(device=dev).manual_seed(M))
            def f():
                with torch.no_grad():
                    for l in layers:
                        l(a)
            us = bench_extras._graph_time(torch, dev, f) / NL * 1e3
            with torch.no_grad():
                y = layers[0](a).float()
            torch.save(y.cpu(), f"/tmp/r16_{sys.argv[2]}_{K}_{N}_{M}.pt")
            row.append(f"M{M}: {us:5.1f} us")
        print(f"{K}->{N}: " + "  ".join(row), flush=True)
    sys.exit(0)
import torch
env0 = dict(os.environ, QLINEAR_LIB_PATH="chatglm_q_amd/csrc/libqlinear_hip_dev.so")
for tag, extra in (("fewrow", {"QLINEAR_DISPATCH": "norows16"}), ("rows16mt2", {"QLINEAR_ROWS16_MAX": "32"})):
    r = subprocess.run([sys.executable, __file__, "child", tag], env=dict(env0, **extra), capture_output=True, text=True, timeout=600)
    print(tag, "|", " | ".join(l for l in r.stdout.strip().splitlines()) or r.stderr[-400:], flush=True)
for K, N in ((4096, 4096), (4096, 4608)):
    for M in (17, 24, 32):
        a, b = torch.load(f"/tmp/r16_fewrow_{K}_{N}_{M}.pt"), torch.load(f"/tmp/r16_rows16mt2_{K}_{N}_{M}.pt")
        print(f"{K}->{N} M={M}: rel-L2 between the kernels {float((a - b).norm() / a.norm()):.2e}")
