"""Developer tool: the 3..16-row kernel on part 1 (w4_rows16.hip) per (column tiles, waves) configuration beside the few-row kernel on part 2
(QLINEAR_DISPATCH=norows16), four ChatGLM2-6B layer shapes, weights rotated.  Developer library: QLINEAR_ROWS16_NT / QLINEAR_ROWS16_KW."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import bench_extras
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(1)
    out = []
    for K, N, NL in ((4096, 4608, 24), (4096, 4096, 24), (4096, 27392, 6), (13696, 4096, 10)):
        layers = [bench_extras._w4_layer(torch, dev, K, N, False, gen) for _ in range(NL)]
        row = []
        for M in (3, 5, 8, 16):
            a = torch.randn(M, K, device=dev, dtype=torch.float16)
            def f():
                with torch.no_grad():
                    for l in layers:
                        l(a)
            row.append(f"{bench_extras._graph_time(torch, dev, f) / NL * 1e3:5.1f}")
        out.append(f"{K}->{N}: " + " ".join(row))
        del layers
    print(" | ".join(out), flush=True)
    sys.exit(0)
env0 = dict(os.environ, QLINEAR_LIB_PATH="chatglm_q_amd/csrc/libqlinear_hip_dev.so")
print("us at M = 3 5 8 16")
for name, extra in [("few-row kernel (part 2)", {"QLINEAR_DISPATCH": "norows16"}), ("product rule", {}), ("product rule, no wide kernel", {"QLINEAR_ROWS16_WIDE": "0"})] + [
        (f"rows16 NT={nt} KW={kw} D={d or 'default'}", {"QLINEAR_ROWS16_NT": str(nt), "QLINEAR_ROWS16_KW": str(kw), "QLINEAR_ROWS16_D": str(d)})
        for nt, kw, d in ((1, 8, 0), (1, 8, 3), (1, 8, 2), (1, 4, 0), (2, 8, 0), (2, 4, 0))]:
    r = subprocess.run([sys.executable, __file__, "child"], env=dict(env0, **extra), capture_output=True, text=True, timeout=600)
    print(f"{name:28s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
