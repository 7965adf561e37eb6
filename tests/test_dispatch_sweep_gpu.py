"""Dispatch-boundary sweep: one ``module.forward`` can land in six int4 kernels (GEMV, 4x4x4-MFMA rows kernel,
few-row kernel, 128-row-tile GEMM, 256 x 256-tile GEMM, 256-tile + 128-tile peel) and four int8 ones, and the gated / residual
entry points exist on a subset of them.  This walks the row count M across EVERY switch of the library's own dispatch table
(``qlinear_w4g32_rows_on_tiled``, ``qlinear_tiled_dispatch``: few-row limit, whole-round rule of the 256-tile kernel, peel rule)
at ChatGLM2-6B's four real layer shapes (chatglm_q/model.py:111-112,194-195) for int4g32 and int8, checks

  * the result against the oracle (every 13th row + the rows at tile / peel boundaries; all columns, or a spread subset of
    the 27 392 columns of w_in - columns are independent), at the tolerance of tests/test_parity_gpu.py;
  * that the kernel family that actually ran (``qlinear_last_dispatch``) is the one the table names - so a flip in the table
    is a flip on the device;
  * that the gated / residual entry points serve exactly the row counts the table says and equal projection + ``silu_mul`` /
    ``+ residual`` (bit for bit where one kernel family computes both).
"""
import ctypes
import functools

import numpy as np
import pytest
import torch

from oracle import qlinear_oracle as O

pytestmark = pytest.mark.gpu

from chatglm_q_amd import _lib  # noqa: E402
from chatglm_q_amd import fused_ops as F_  # noqa: E402
from chatglm_q_amd.int4 import hip_ops as h4  # noqa: E402
from chatglm_q_amd.int4 import qlinear as q4  # noqa: E402
from chatglm_q_amd.int8 import hip_ops as h8  # noqa: E402
from chatglm_q_amd.int8 import qlinear as q8  # noqa: E402

DEV = "cuda:0"
# (K, N, bias, role)   chatglm_q/model.py:111 (qkv_proj, bias), :112 (o_proj), :194 (w_in -> SiLU * gate), :195 (w_out -> + residual)
SHAPES = {"qkv_proj": (4096, 4608, True, None), "o_proj": (4096, 4096, False, "residual"),
          "w_in": (4096, 27392, False, "gated"), "w_out": (13696, 4096, False, "residual"),
          "lm_head": (4096, 65024, False, None)}            # chatglm_q/model.py:262-263,382: 254 column tiles of 256
K_OTHER, K_W4_GEMV, K_W4_ROWS4, K_W4_FEWROW, K_W4_GEMM128, K_W4_GEMM256 = 1, 2, 3, 4, 5, 6
K_SPLITK, K_W8_GEMV, K_W8_FEWROW, K_W8_GEMM128, K_W8_GEMM256 = 9, 10, 11, 12, 13
K_W4_ROWS16 = 19


def part1_family(M, N, K, dtype, strict):
    """The family that serves a call from part 1 of the derived layout (0: part 2 does) - the library's own table."""
    return _lib.get_lib().qlinear_w4g32_packed_dispatch(M, N, K, _lib.dtype_code(dtype), 1 if strict else 0)
M_MAX = 8192
BASE_M = sorted(set(range(1, 10)) | {15, 16, 17, 31, 32, 33, 63, 64, 65, 66, 127, 128, 129, 130, 255, 256, 257, 258, M_MAX})


def t2n(t):
    t = t.detach().cpu()
    return t.float().numpy() if t.dtype == torch.bfloat16 else t.numpy()


def table(bits, M, N, K):
    """(family of the first launch, rows it serves) of the tile-major route - the library's dispatch table as a function."""
    first = ctypes.c_int64(0)
    fam = _lib.get_lib().qlinear_tiled_dispatch(bits, M, N, K, ctypes.addressof(first))
    return fam, first.value


@functools.lru_cache(maxsize=None)
def m_values(bits, N, K):
    """BASE_M + every M = F where the table switches, with F - 1 and F + 1; where the kernel FAMILY switches (few-row limit,
    256-tile rule) also F - 255 and F + 255 (the peel of w_in goes on and off 23 times below 8192 rows: F - 1, F, F + 1 there)."""
    ms, prev = set(BASE_M), None
    for M in range(1, M_MAX + 1):
        fam, first = table(bits, M, N, K)
        cur = (fam, first < M)
        if prev is not None and cur != prev:
            near = (M - 255, M - 1, M, M + 1, M + 255) if cur[0] != prev[0] else (M - 1, M, M + 1)
            ms.update(m for m in near if 1 <= m <= M_MAX)
        prev = cur
    return tuple(sorted(ms))


def families():
    """Kernel families of the launches since the last reset, oldest first, without repacks / reductions."""
    log, out = int(_lib.get_lib().qlinear_last_dispatch()), []
    while log:
        out.append(log & 0xFF)
        log >>= 8
    return [f for f in reversed(out) if f not in (K_OTHER, K_SPLITK)]


def expected(bits, M, N, K, dtype):
    if bits == 4:
        strict = _lib.strict_for(dtype)
        if not h4.rows_on_tiled(M, N, K, dtype, strict):
            fam = part1_family(M, N, K, dtype, strict)
            assert fam in (K_W4_GEMV, K_W4_ROWS4, K_W4_ROWS16) and (fam != K_W4_ROWS4 or not strict) and (fam != K_W4_GEMV or M <= 2 or M > 16)
            return [fam]
        fam, first = table(4, M, N, K)
        return [fam] + ([K_W4_GEMM128] if first < M else [])
    if M <= 2:
        return [K_W8_GEMV]
    return [table(8, M, N, K)[0]]


def sample_rows(M, first):
    rows = set(range(0, M, 13)) | {M - 1}
    for edge in (first, (M // 256) * 256):                    # peel boundary, last full row tile
        rows.update(r for r in (edge - 1, edge) if 0 <= r < M)
    return torch.tensor(sorted(rows), dtype=torch.long)


def sample_cols(N, role):
    """All columns of the narrow layers; for w_in 64-column runs spread over the matrix (first, last and ragged 256-tile included)."""
    if N <= 8192:
        return None
    if role != "gated":                                       # lm_head: runs over the whole width, both ends and the last 256-tile
        starts = sorted({0, 192, N - 256, N - 64} | set(range(1000, N - 64, 3391)))
        return torch.unique(torch.cat([torch.arange(s, s + 64) for s in starts]))
    hidden = N // 2
    starts = sorted({0, 192, 4096 - 32, hidden - 64, hidden - 256 + 32} | set(range(1000, hidden - 64, 1709)))
    cols = torch.cat([torch.arange(s, s + 64) for s in starts])
    cols = torch.unique(cols[cols < hidden])
    return torch.cat([cols, cols + hidden]) if role == "gated" else cols   # SiLU * gate pairs column c with hidden + c


class Layer:
    """A seeded module of one layer shape on the GPU + the oracle's view of (a column subset of) its weights."""

    def __init__(self, name, bits, dtype):
        K, N, has_bias, role = SHAPES[name]
        self.K, self.N, self.role, self.bits, self.dtype = K, N, role, bits, dtype
        self.dt = "f16" if dtype == torch.float16 else "bf16"
        g = torch.Generator().manual_seed(N * 3 + K + bits)
        bias = (torch.randn(N, generator=g) * 0.1).to(dtype) if has_bias else None
        if bits == 4:
            qw = torch.randint(0, 256, (K // 2, N), dtype=torch.uint8, generator=g)
            sc = (torch.rand((K // 32, N), generator=g) * 0.02 + 0.002).to(dtype)
            self.mod = q4.DynamicQuantizeLinear(K, N, bias=has_bias, dtype=dtype)
            self.mod.apply_weights_(qw, sc, bias)
        else:
            qw = torch.randint(-128, 128, (N, K), dtype=torch.int8, generator=g)
            sc = ((torch.rand(N, generator=g) * 0.01 + 0.001) * torch.where(torch.rand(N, generator=g) < 0.1, -1.0, 1.0)).to(dtype)
            self.mod = q8.DynamicQuantizeLinear(K, N, bias=has_bias, dtype=dtype)
            self.mod.apply_weights_(qw, sc, bias)
        self.mod = self.mod.to(DEV)
        self.cols = sample_cols(N, role)
        c = slice(None) if self.cols is None else self.cols
        self.o_bias = None if bias is None else t2n(bias[c])
        # the oracle's dequantised weights (K, columns), computed once per layer: w4_matmul / w8_matmul spend seconds per call on
        # them at these sizes; oracle() below is their remaining three lines (product in fp64, one rounding, bias as a second
        # rounded add - oracle/qlinear_oracle.py:136-153,209-221) and is pinned to the functions themselves on a few rows here
        if bits == 4:
            o_w, o_s = np.ascontiguousarray(qw[:, c].numpy()), np.ascontiguousarray(t2n(sc[:, c]))
            self.w64 = O.as_f64(O.unpack_int4(o_w, o_s, self.dt))
        else:
            o_w, o_s = np.ascontiguousarray(qw[c].numpy().T), np.ascontiguousarray(t2n(sc[c]))
            self.w64 = O.as_f64(O.w8_dequant(o_w, o_s, self.dt))
        self.x = torch.randn((M_MAX, K), generator=torch.Generator().manual_seed(K + 1)).to(dtype)
        self.xd = self.x.to(DEV)
        probe = torch.tensor([0, 5, M_MAX - 1])
        full = (O.w4_matmul if bits == 4 else O.w8_matmul)(t2n(self.x[probe]), o_w, o_s, self.o_bias, dtype=self.dt)
        assert np.array_equal(self.oracle(probe), full)

    def oracle(self, rows):
        out = O.round_to(O.as_f64(t2n(self.x[rows])) @ self.w64, self.dt)
        if self.o_bias is not None:
            out = O.round_to(O.as_f64(out) + O.as_f64(self.o_bias)[None, :], self.dt)
        return out


def check_forward(L, M):
    lib = _lib.get_lib()
    x = L.xd[:M]
    lib.qlinear_dispatch_reset()
    with torch.no_grad():
        y = L.mod(x)
    ran = families()
    want = expected(L.bits, M, L.N, L.K, L.dtype)
    assert ran == want, f"M={M}: kernel families {ran}, dispatch table says {want}"
    first = table(L.bits, M, L.N, L.K)[1] if want[0] in (K_W4_GEMM256,) else M
    rows = sample_rows(M, first)
    got = y[rows.to(DEV)]
    got = t2n(got if L.cols is None else got[:, L.cols.to(DEV)])
    ref = L.oracle(rows)
    err = O.rel_l2(got, ref)
    # one / two bf16 rows in the default arithmetic would be the documented 4e-3 exception; the policy makes bf16 strict
    assert err <= 1e-3, f"M={M} {want}: rel-L2 {err:.2e} vs oracle"
    assert np.isfinite(got).all()
    return y, want, first


def check_gated(L, M, y, want, first):
    """w_in: the SiLU * gate epilogue entry points against forward() + silu_mul."""
    hidden = L.N // 2
    lib = _lib.get_lib()
    x = L.xd[:M]
    serves256 = bool(lib.qlinear_gemm256_serves(M, L.N, L.K))
    lib.qlinear_dispatch_reset()
    if L.bits == 4:
        part1 = not h4.rows_on_tiled(M, L.N, L.K, L.dtype, _lib.strict_for(L.dtype))
        if part1 and M == 1:
            return
        gp, gb = L.mod.gated_packed(hidden) if part1 else L.mod.gated_tiled(hidden)
        out = h4.w4_forward_gated(x, gp, L.N, gb, part1=part1)
        should = part1 or M <= 32 or serves256
        fam = part1_family(M, L.N, L.K, L.dtype, _lib.strict_for(L.dtype)) if part1 else (K_W4_FEWROW if M <= 32 else K_W4_GEMM256)
    else:
        if M <= 2:
            return
        gt, gs, gb = L.mod.gated_tiled(hidden)
        out = h8.w8_forward_tiled_gated(x, gt, L.N, gs, gb)
        should, fam = serves256, K_W8_GEMM256
    assert (out is not None) == should, f"gated entry at M={M}: served={out is not None}, table says {should}"
    if out is None:
        return
    assert families() == [fam], (M, families(), fam)
    ref = F_.silu_mul(y, hidden)
    same_kernel = want == [fam]
    if same_kernel:
        assert torch.equal(out, ref), f"gated entry != projection + silu_mul at M={M} (same kernel family {fam})"
    else:                                                   # forward() peeled / used another family: same sums in another order
        lim = min(first, M)
        if want[0] == fam and lim > 0:
            assert torch.equal(out[:lim], ref[:lim]), f"gated entry differs on the rows both routes gave to family {fam} (M={M})"
        assert O.rel_l2(t2n(out), t2n(ref)) <= 2e-3, (M, O.rel_l2(t2n(out), t2n(ref)))


def check_residual(L, M, y, want):
    lib = _lib.get_lib()
    x = L.xd[:M]
    resid = torch.randn((M, L.N), generator=torch.Generator().manual_seed(M)).to(L.dtype).to(DEV)
    serves256 = bool(lib.qlinear_gemm256_serves(M, L.N, L.K))
    lib.qlinear_dispatch_reset()
    if L.bits == 4:
        if M < 3:
            return
        out = h4.w4_forward_tiled_residual(x, L.mod.tiled(), L.N, L.mod.bias, resid)
        fam = K_W4_GEMM256
    else:
        if M < 3:
            return
        out = h8.w8_forward_tiled_residual(x, L.mod.prepare()._tiled, L.N, L.mod.weight_scale, L.mod.bias, resid)
        fam = K_W8_GEMM256
    assert (out is not None) == serves256, f"residual entry at M={M}: served={out is not None}, table says {serves256}"
    if out is None:
        return
    assert families() == [fam]
    ref = y + resid
    if want == [fam]:
        assert torch.equal(out, ref), f"residual entry != projection + add at M={M}"
    else:
        assert O.rel_l2(t2n(out), t2n(ref)) <= 2e-3


def run_sweep(name, bits, dtype):
    L = Layer(name, bits, dtype)
    seen = set()
    for M in m_values(bits, L.N, L.K):
        y, want, first = check_forward(L, M)
        seen.add(tuple(want))
        if L.role == "gated":
            check_gated(L, M, y, want, first)
        elif L.role == "residual":
            check_residual(L, M, y, want)
        del y
    return seen


@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("name", list(SHAPES))
def test_dispatch_boundaries_fp16(name, bits):
    seen = run_sweep(name, bits, torch.float16)
    if bits == 4:
        need = {(K_W4_GEMV,), (K_W4_ROWS4,), (K_W4_FEWROW,), (K_W4_GEMM128,), (K_W4_GEMM256,)}
    else:
        need = {(K_W8_GEMV,), (K_W8_FEWROW,), (K_W8_GEMM128,), (K_W8_GEMM256,)}
    assert need <= seen, f"{name}: kernel families never reached: {need - seen}"


@pytest.mark.parametrize("bits", [4, 8])
def test_dispatch_boundaries_bf16(bits):
    """One bf16 pass (strict per-weight rounding by policy: no 4x4x4-MFMA kernel, GEMV for 1..2 rows)."""
    seen = run_sweep("o_proj", bits, torch.bfloat16)
    assert ((K_W4_GEMM256,) if bits == 4 else (K_W8_GEMM256,)) in seen


def test_last_round_half_tiles_replace_the_peel(monkeypatch):
    """Round 5: the int4g32 256-tile launch runs a last round that fills at most half the chip as half tiles of its own, so the dispatch
    table no longer peels - one launch, one summation order for all rows (any K: 64 and 65 K tiles here).  QLINEAR_DISPATCH=nohalf brings
    the older peel back (last row tiles on the 128-row-tile kernel)."""
    lib = _lib.get_lib()
    M, N = 8192, 4608                                       # 576 tiles of 256 x 256 = 2.25 rounds of 256 workgroups
    g = torch.Generator().manual_seed(77)
    for K in (4096, 4160):
        fam, first = table(4, M, N, K)
        assert fam == K_W4_GEMM256 and first == M           # no peel: qkv_proj at config 5's row count is ONE launch
        qw = torch.randint(0, 256, (K // 2, N), dtype=torch.uint8, generator=g)
        sc = (torch.rand((K // 32, N), generator=g) * 0.02 + 0.002).half()
        mod = q4.DynamicQuantizeLinear(K, N, bias=False, dtype=torch.float16)
        mod.apply_weights_(qw, sc, None)
        mod = mod.to(DEV)
        x = torch.randn((M, K), generator=g).half()
        lib.qlinear_dispatch_reset()
        with torch.no_grad():
            y = mod(x.to(DEV))
        assert families() == [K_W4_GEMM256], (K, families())
        rows = torch.tensor([0, 255, 4095, 7167, 7168, 7295, 7296, 7423, 7424, 8191])     # both halves of the last round's tiles
        ref = O.w4_matmul(t2n(x[rows]), qw.numpy(), t2n(sc), None, dtype="f16")
        assert O.rel_l2(t2n(y[rows.to(DEV)]), ref) <= 1.5e-4
        try:
            monkeypatch.setenv("QLINEAR_DISPATCH", "nohalf")
            lib.qlinear_dispatch_reload()
            fam, first = table(4, M, N, K)
            assert fam == K_W4_GEMM256 and 0 < first < M
            lib.qlinear_dispatch_reset()
            with torch.no_grad():
                y2 = mod(x.to(DEV))
            assert families() == [K_W4_GEMM256, K_W4_GEMM128]
            assert torch.equal(y2[:first], y[:first])       # the rows whole tiles serve in both: the same bits
            assert O.rel_l2(t2n(y2[rows.to(DEV)]), ref) <= 1.5e-4
        finally:
            monkeypatch.delenv("QLINEAR_DISPATCH")
            lib.qlinear_dispatch_reload()


def test_dispatch_override_switches_families(monkeypatch):
    """QLINEAR_DISPATCH (the one environment variable the library reads) removes families from the table - and from the device."""
    lib = _lib.get_lib()
    K, N = 4096, 4096
    try:
        monkeypatch.setenv("QLINEAR_DISPATCH", "no256,nofewrow,norows4,norows16")
        lib.qlinear_dispatch_reload()
        assert lib.qlinear_gemm256_serves(8192, N, K) == 0
        assert table(4, 8192, N, K)[0] == K_W4_GEMM128 and table(4, 8, N, K)[0] == K_W4_GEMM128
        assert h4.rows_on_tiled(3, N, K, torch.float16, False) is True
        assert part1_family(8, N, K, torch.float16, False) == 0
        monkeypatch.setenv("QLINEAR_DISPATCH", "norows4")
        lib.qlinear_dispatch_reload()
        assert part1_family(3, N, K, torch.float16, False) == K_W4_ROWS16      # 3..4 rows fall to the 16-row kernel, still on part 1
    finally:
        monkeypatch.delenv("QLINEAR_DISPATCH")
        lib.qlinear_dispatch_reload()
    assert lib.qlinear_gemm256_serves(8192, N, K) == 1 and table(4, 8, N, K)[0] == K_W4_FEWROW
    assert part1_family(8, N, K, torch.float16, False) == K_W4_ROWS16 and part1_family(3, N, K, torch.float16, False) == K_W4_ROWS4
    assert part1_family(8, 27392, K, torch.float16, False) == K_W4_ROWS16      # the wide first MLP projection: rows in LDS, still part 1
    assert part1_family(16, 27392, 2 * K, torch.float16, False) == 0           # ... while they fit: otherwise the few-row kernel on part 2
