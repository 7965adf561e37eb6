"""Developer tool (round 5): the int4g32 256-tile launch with its half-tile last round (small rounds: QLINEAR_G256_PGRID) against whole tiles
only, bit for bit; also used across MFMA bodies (QLINEAR_G256_MI16=0 / 1: close, not equal).

  QLINEAR_LIB_PATH=chatglm_q_amd/csrc/libqlinear_hip_dev.so QLINEAR_GEMM_256_MIN_BLOCKS=1 QLINEAR_G256_TAIL=0 python tools/g256p_check.py save
  ... QLINEAR_G256_PGRID=16 python tools/g256p_check.py check

(the knobs are read once per process: two runs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from chatglm_q_amd.int4 import hip_ops as h4  # noqa: E402

dev = torch.device("cuda:0")
mode = sys.argv[1]
path = os.environ.get("G256P_FILE", "/tmp/g256p_ref.pt")
CASES = [  # kind, M, K, N, dtype, bias
    ("plain", 2048, 1024, 4608, torch.float16, True),       # 144 tiles
    ("plain", 3000, 1024, 2000, torch.bfloat16, False),     # 96 tiles, ragged M and N
    ("plain", 1000, 4096, 1000, torch.bfloat16, False),
    ("plain", 777, 1024, 264, torch.float16, True),
    ("plain", 1536, 2048, 2048, torch.float16, False),      # 48 tiles
    ("plain", 2900, 1152, 1288, torch.float16, True),       # 72 tiles, 18 K tiles, ragged: the last row tile's second half holds 84 rows
    ("resid", 2048, 1024, 1024, torch.float16, True),       # 32 tiles
    ("resid", 2200, 1024, 1544, torch.bfloat16, False),     # 63 tiles, ragged
    ("gated", 1024, 1024, 2048, torch.float16, True),       # 32 tiles
    ("gated", 2500, 2048, 1056, torch.bfloat16, False),     # 50 tiles, ragged
]
out = {}
for i, (kind, M, K, N, dt, has_bias) in enumerate(CASES):
    g = torch.Generator().manual_seed(100 + i)
    qw = torch.randint(0, 256, (K // 2, N), dtype=torch.uint8, generator=g).to(dev)
    sc = (torch.rand((K // 32, N), generator=g) * 0.02 + 0.002).to(dt).to(dev)
    a = torch.randn((M, K), generator=g).to(dt).to(dev)
    bias = (torch.randn(N, generator=g) * 0.1).to(dt).to(dev) if has_bias else None
    tiled = h4.tile_w4g32(h4.repack_w4g32_gemv(qw, sc), N, K, dt)
    if kind == "plain":
        y = h4.w4_gemm256(a, tiled, N, bias)
    elif kind == "resid":
        r = torch.randn((M, N), generator=g).to(dt).to(dev)
        y = h4.w4_forward_tiled_residual(a, tiled, N, bias, r)
    else:
        y = h4.w4_forward_gated(a, tiled, N, bias)      # any copy serves as a "gate-interleaved" one: same arithmetic
    assert y is not None, (kind, M, K, N)
    torch.cuda.synchronize()
    out[i] = y.cpu()
if mode == "save":
    torch.save(out, path)
    print("saved", len(out), "cases")
else:
    ref = torch.load(path)
    bad = 0
    for i, c in enumerate(CASES):
        same = torch.equal(ref[i], out[i])
        fin = bool(torch.isfinite(out[i].float()).all())
        nd = int((ref[i] != out[i]).sum())
        rel = float((ref[i].float() - out[i].float()).norm() / ref[i].float().norm())
        print(("OK  " if same else "DIFF"), c[:4], str(c[4]).split(".")[-1], "finite" if fin else "NONFINITE", "mismatches", nd, f"rel-L2 {rel:.2e}", flush=True)
        bad += 0 if same else 1
    print("RESULT", "all bit-equal" if bad == 0 else f"{bad} cases differ")
    sys.exit(1 if bad else 0)
