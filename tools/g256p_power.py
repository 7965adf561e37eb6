"""Developer tool (round 5): is the 256-tile int4 GEMM power bound?  Runs one shape for ~3 s per setting while rocm-smi samples socket power
and the shader clock; with QLINEAR_G256_PGRID=n (developer library) the persistent kernel runs on n CUs only - if half the CUs deliver
much more than half the throughput, the full chip is held back by its power budget, not by what a CU does per cycle.

  QLINEAR_LIB_PATH=chatglm_q_amd/csrc/libqlinear_hip_dev.so QLINEAR_G256_PGRID=128 python tools/g256p_power.py [shape]
"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_extras  # noqa: E402

dev = torch.device("cuda:0")
shapes = {"o_proj": (4096, 4096), "w_in": (4096, 27392), "w_out": (13696, 4096), "qkv_proj": (4096, 4608)}
name = sys.argv[1] if len(sys.argv) > 1 else "o_proj"
K, N = shapes[name]
M = int(os.environ.get("M", 8192))
g = torch.Generator(device=dev).manual_seed(5)
layers = [bench_extras._w4_layer(torch, dev, K, N, False, g) for _ in range(4)]
x = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
with torch.no_grad():
    for l in layers:
        l(x)
torch.cuda.synchronize()
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append(out)
        except Exception as e:  # noqa: BLE001
            samples.append(f"ERR {e}")
        time.sleep(0.2)


th = threading.Thread(target=sampler)
th.start()
t0 = time.perf_counter()
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
with torch.no_grad():
    while time.perf_counter() - t0 < 3.0:
        for _ in range(20):
            for l in layers:
                l(x)
        n += 80
        torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
stop = True
th.join()
us = e0.elapsed_time(e1) * 1e3 / n
print(f"{name} {M}x{K}x{N} PGRID={os.environ.get('QLINEAR_G256_PGRID', '-')} PERSIST={os.environ.get('QLINEAR_G256_PERSIST', '-')}: "
      f"{us:.1f} us per launch (host loop, {n} launches) = {2.0 * M * N * K / us / 1e6:.0f} TF")
mid = samples[len(samples) // 2] if samples else ""
print("rocm-smi sample (middle of the run):")
print(mid.strip()[:1200])
