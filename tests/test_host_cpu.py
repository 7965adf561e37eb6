"""CPU tests of the host side: module API mirrors the reference (names, buffers, dtypes, CPU branch),
quantizer writers are format-exact, the C ABI library loads and exports everything the header declares.
No compute call goes to the GPU here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import _golden as G
from oracle import qlinear_oracle as O

from chatglm_q_amd import _lib
from chatglm_q_amd.int4 import qlinear as q4, quantizer as z4
from chatglm_q_amd.int8 import qlinear as q8, quantizer as z8

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INT4 = G.load("int4_matmul.npz")
INT8 = G.load("int8_matmul.npz")
TDT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
REL = {"f32": 1e-5, "f16": 1e-3, "bf16": 1e-3}


def t2n(t):
    return t.float().numpy() if t.dtype == torch.bfloat16 else t.numpy()


def test_public_surface_matches_reference_names():
    for name in ("DEFAULT_GROUP_SIZE", "KERNEL_IMPL", "check_input", "unpack_int4", "DynamicQuantizeMatMul",
                 "dynamic_quant_matmul", "DynamicQuantizeLinear", "QEmbedding"):
        assert hasattr(q4, name), name
    for name in ("KERNEL_IMPL", "check_input", "DynamicQuantizeMatMul", "dynamic_quant_matmul",
                 "DynamicQuantizeLinear", "QEmbedding"):
        assert hasattr(q8, name), name
    assert q4.DEFAULT_GROUP_SIZE == 32
    assert q4.KERNEL_IMPL in ("hip", "none")
    assert q4.check_input(torch.zeros(1)) is False


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    assert _lib.available(), "libqlinear_hip.so must be built (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "qlinear_hip.h")).read()
    declared = set(re.findall(r"\b(qlinear_\w+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    # the developer library exports the product surface + the experiments of include/qlinear_hip_dev.h, and only it does
    dev_header = open(os.path.join(ROOT, "include", "qlinear_hip_dev.h")).read()
    dev_declared = set(re.findall(r"^(?:int|size_t)\s+(qlinear_\w+)\s*\(", dev_header, re.M))
    assert dev_declared == set(_lib.DEV_EXPORTS), dev_declared ^ set(_lib.DEV_EXPORTS)
    dev = ctypes.CDLL(_lib.DEV_LIB_PATH)
    for name in declared | dev_declared:
        assert getattr(dev, name) is not None
    for name in dev_declared:
        assert not hasattr(lib, name), name
    assert _lib.get_lib().qlinear_abi_version() == 3
    assert _lib.get_lib().qlinear_status_string(0) == b"ok"
    assert b"group" in _lib.get_lib().qlinear_status_string(-4)
    # host-only queries (no GPU needed)
    # column-major part (GEMV) + tile-major part (MFMA kernels), each units + scales
    assert _lib.get_lib().qlinear_w4g32_packed_bytes(4096, 4096, 32, 1) == 2 * (4096 * 128 * 16 + 4096 * 128 * 2)
    assert _lib.get_lib().qlinear_w4g32_packed_bytes(4096, 4096, 64, 1) == 0
    assert _lib.get_lib().qlinear_workspace_bytes(1, 1, 4096, 4096, 32) == 8 * 1 * 4096 * 4
    assert _lib.get_lib().qlinear_workspace_bytes(1, 1, 4096, 512, 32) == 0


def test_argument_errors_come_back_as_status_codes_without_a_gpu():
    lib = _lib.get_lib()
    buf = (ctypes.c_char * 64)()
    p = ctypes.addressof(buf)
    assert lib.qlinear_w4g32_fwd(None, None, None, None, None, 1, 8, 64, 32, 64, 8, 1, None, 0, None) == -1
    assert lib.qlinear_w4g32_fwd(p, p, p, None, p, 0, 8, 64, 32, 64, 8, 1, None, 0, None) == -2
    assert lib.qlinear_w4g32_fwd(p, p, p, None, p, 1, 8, 64, 32, 64, 8, 7, None, 0, None) == -3
    assert lib.qlinear_w4g32_fwd(p, p, p, None, p, 1, 8, 64, 24, 64, 8, 1, None, 0, None) == -4
    assert lib.qlinear_w4g32_fwd_packed(p, p, None, p, 1, 8, 64, 64, 64, 8, 1, 0, None, 0, None) == -4
    assert lib.qlinear_w8a8_fwd(p, p, p, p, None, p, 1, 8, 40, 8, 1, None, 0, None) == -7      # K % 16 != 0
    with pytest.raises(ValueError, match="group"):
        _lib.check(-4, "x")


@pytest.mark.parametrize("entry", [":".join(c) for c in G.cases(INT4)])
def test_int4_module_cpu_branch_matches_reference(entry):
    name, dt, has_bias = entry.split(":")
    c = G.case(INT4, name, dt)
    K, N = c["qweight"].shape[0] * 2, c["qweight"].shape[1]
    layer = q4.DynamicQuantizeLinear(K, N, bias=has_bias == "1", dtype=TDT[dt])
    assert [k for k, _ in layer.named_buffers()] == (["weight", "weight_scale", "bias"] if has_bias == "1" else ["weight", "weight_scale"])
    assert layer.weight.shape == (K // 2, N) and layer.weight.dtype == torch.uint8
    assert layer.weight_scale.shape == (K // 32, N) and layer.weight_scale.dtype == TDT[dt]
    layer.apply_weights_(torch.from_numpy(c["qweight"]), G.to_torch(c["scale"], dt),
                         G.to_torch(c["bias"], dt) if has_bias == "1" else None)
    out = layer(G.to_torch(c["a"], dt))
    assert out.dtype == TDT[dt]
    assert O.rel_l2(t2n(out), c["out_fallback"]) <= REL[dt]
    if "dense" in c:
        dense = q4.unpack_int4(layer.weight, layer.weight_scale)
        assert np.array_equal(t2n(dense).astype(np.float32), np.asarray(c["dense"], np.float32))


@pytest.mark.parametrize("entry", [":".join(c) for c in G.cases(INT8)])
def test_int8_module_cpu_branch_matches_reference(entry):
    name, dt, has_bias, layout = entry.split(":")
    c = G.case(INT8, name, dt)
    a = G.to_torch(c["a"], dt)
    if layout == "kn":
        out = q8.dynamic_quant_matmul(a, torch.from_numpy(c["w_kn"]), G.to_torch(c["scale"], dt))
    else:
        N, K = c["weight_nk"].shape
        layer = q8.DynamicQuantizeLinear(K, N, bias=has_bias == "1", dtype=TDT[dt])
        assert layer.weight.shape == (N, K) and layer.weight.dtype == torch.int8
        assert layer.weight_scale.shape == (N,)
        layer.apply_weights_(torch.from_numpy(c["weight_nk"]), G.to_torch(c["scale"], dt),
                             G.to_torch(c["bias"], dt) if has_bias == "1" else None)
        out = layer(a)
    assert O.rel_l2(t2n(out), c["out_fallback"]) <= REL[dt]


def test_baseline_config1_int8_128x4096x4096_cpu_plumbing():
    """BASELINE.json configs[0]: int8 QLinear forward (128 x 4096 -> 4096) on CPU through the module API."""
    torch.manual_seed(0)
    a = torch.randn(128, 4096)
    w = torch.randn(4096, 4096) / 64
    layer = z8.get_quant_int8_linear(torch.nn.Linear(4096, 4096, bias=True))
    q, s = z8.quantize_int8(w)
    layer.apply_weights_(q, s, torch.zeros(4096))
    out = layer(a)
    assert out.shape == (128, 4096) and out.dtype == torch.float32
    ref = O.w8_matmul(a.numpy()[:4], np.ascontiguousarray(q.numpy().T), s.numpy(), np.zeros(4096, np.float32), dtype="f32")
    assert np.allclose(out.numpy()[:4], ref, atol=1e-4, rtol=1e-4)
    sd = layer.state_dict()
    assert sorted(sd) == ["bias", "weight", "weight_scale"] and sd["weight"].dtype == torch.int8


@pytest.mark.parametrize("dt", ["f32", "f16"])
def test_quantizer_writers_are_format_exact(dt):
    z = G.load("quantizers.npz")
    q, s = z4.quantize_int4(torch.from_numpy(z[f"int4_{dt}/w"]))
    assert np.array_equal(q.numpy(), z[f"int4_{dt}/q"])
    assert np.array_equal(s.numpy().astype(np.float32), z[f"int4_{dt}/scale"].astype(np.float32))
    q8_, s8 = z8.quantize_int8(torch.from_numpy(z[f"int8_{dt}/x"]))
    assert np.array_equal(q8_.numpy(), z[f"int8_{dt}/q"])
    assert np.array_equal(s8.numpy().astype(np.float32), z[f"int8_{dt}/scale"].astype(np.float32))


def test_qembedding_cpu_branch():
    z = G.load("qembedding.npz")
    ids = torch.from_numpy(z["ids"])
    e4 = q4.QEmbedding(128, 64, dtype=torch.float16)
    e4.apply_weights_(torch.from_numpy(z["int4/qweight"]), torch.from_numpy(z["int4/scale"]))
    assert np.array_equal(e4(ids).numpy(), z["int4/out"])
    e8 = q8.QEmbedding(128, 64, dtype=torch.float32)
    e8.apply_weights_(torch.from_numpy(z["int8/weight"]), torch.from_numpy(z["int8/scale"]))
    assert np.array_equal(e8(ids).numpy(), z["int8/out"])


def test_loader_style_in_place_fill_and_converters():
    """The reference loader copies tensors straight into state_dict() (chatglm_q/loader.py:90-104)."""
    layer = q4.DynamicQuantizeLinear(64, 16, bias=True, dtype=torch.float16)
    sd = layer.state_dict()
    sd["weight"].copy_(torch.full((32, 16), 0x88, dtype=torch.uint8))
    sd["weight_scale"].copy_(torch.ones(2, 16).half())
    sd["bias"].copy_(torch.arange(16).half())
    out = layer(torch.randn(3, 64).half())
    assert torch.equal(out, torch.arange(16).half().expand(3, 16))      # all weights decode to zero
    lin = torch.nn.Linear(64, 16)
    ql = z4.get_quant_int4_linear(lin)
    assert ql.weight.shape == (32, 16) and ql.bias is not None
    assert torch.allclose(ql(torch.eye(64))[:4], lin(torch.eye(64))[:4], atol=0.05)
    with pytest.raises(AssertionError):
        q4.DynamicQuantizeLinear(100, 16, group_size=32)


def test_autograd_through_cpu_branch():
    torch.manual_seed(1)
    a = torch.randn(4, 64, requires_grad=True)
    w = torch.randn(64, 16) / 8
    q, s = z4.quantize_int4(w)
    q4.dynamic_quant_matmul(a, q, s).sum().backward()
    assert torch.allclose(a.grad, torch.ones(4, 16) @ q4.unpack_int4(q, s).t(), atol=1e-5)


def test_gpu_tensor_without_library_raises(monkeypatch):
    """The product path must fail loudly, never fall back, when the HIP extension is missing."""
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_load_error", OSError("simulated missing library"))
    with pytest.raises(_lib.QLinearLibraryMissing):
        _lib.get_lib()


def test_derived_layout_keys_survive_inference_tensors_and_invalidate():
    """Tensor._version raises on inference tensors and misses .data writes; the cache key must not."""
    from chatglm_q_amd import _lib
    from chatglm_q_amd.int4.qlinear import DynamicQuantizeLinear as Q4
    from chatglm_q_amd.int8.qlinear import DynamicQuantizeLinear as Q8
    with torch.inference_mode():
        l4 = Q4(64, 32, bias=True, dtype=torch.float16)
        l8 = Q8(64, 32, bias=True, dtype=torch.float16)
        l4.weight.zero_(); l4.weight_scale.fill_(1); l4.bias.zero_()
        l8.weight.zero_(); l8.weight_scale.fill_(1); l8.bias.zero_()
    k4 = l4._canonical_key()                 # must not raise "Inference tensors do not track version counter"
    k8 = _lib.buffer_key(l8.weight, l8.weight_scale, l8.bias)
    assert k4 == l4._canonical_key() and k8 == _lib.buffer_key(l8.weight, l8.weight_scale, l8.bias)
    n4 = Q4(64, 32, bias=True, dtype=torch.float16)
    before = n4._canonical_key()
    n4.weight.copy_(torch.ones_like(n4.weight))
    assert n4._canonical_key() != before     # the loader's in-place copy_ is seen
    # explicit invalidation exists on both modules and is what load_state_dict / apply_weights_ / .to() trigger
    for mod in (n4, Q8(64, 32, dtype=torch.float16)):
        mod._gated, mod._gated_key = ("stale",), "k"
        mod.invalidate()
        assert mod._gated is None and mod._gated_key is None
        mod._gated = ("stale",)
        mod.load_state_dict(mod.state_dict())
        assert mod._gated is None
        mod._gated = ("stale",)
        mod.to(torch.float16)
        assert mod._gated is None


def test_lib_load_is_retried_when_the_file_changes(tmp_path, monkeypatch):
    """a failed dlopen must not be cached forever (package imported before build())."""
    from chatglm_q_amd import _lib
    real = _lib.LIB_PATH
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_load_error", None)
    monkeypatch.setattr(_lib, "_load_stamp", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    assert not _lib.available()
    with pytest.raises(_lib.QLinearLibraryMissing):
        _lib.get_lib()
    monkeypatch.setattr(_lib, "LIB_PATH", real)          # "the build finished": path now resolves, stamp differs
    assert _lib.available()


def test_attention_tile_flags_against_brute_force():
    """fused_ops.attention_tile_flags (host side of qlinear_prefill_attention): 0 only where every entry of the tile is <= -1e9 AND
    every query row of the block sees an entry >= -1e6 somewhere, 2 only where the tile is all zero, 1 otherwise - for causal chunks,
    left pads (rows with no visible key forbid skipping in their block), ragged shapes and a soft bias."""
    import torch
    from chatglm_q_amd import fused_ops as F_
    QB, KB = F_.prefill_attention_tiles()
    assert (QB, KB) == (16, 64)
    g = torch.Generator().manual_seed(0)
    cases = []
    for (B, S, T, first) in [(2, 64, 192, 128), (1, 37, 42, 5), (3, 48, 130, 60)]:
        t = torch.arange(T)
        rows = torch.arange(first, first + S)
        blocked = (t[None, None, :] > rows[None, :, None]).expand(B, S, T).clone()
        pads = torch.randint(0, T // 2, (B,), generator=g)
        for b in range(B):
            blocked[b, :, : int(pads[b])] = True
        cases.append(blocked.float() * -1e10)
    cases.append(-0.05 * torch.rand(1, 40, 100, generator=g))          # soft bias: nothing skippable, nothing all-zero
    cases.append(torch.zeros(1, 16, 64))                                # one all-zero tile
    for mask in cases:
        B, S, T = mask.shape
        flags = F_.attention_tile_flags(mask)
        nq, nk = (S + QB - 1) // QB, (T + KB - 1) // KB
        assert flags.shape == (B, nq, nk) and flags.dtype == torch.uint8
        for b in range(B):
            for qb in range(nq):
                rows = mask[b, qb * QB:(qb + 1) * QB]
                rows_ok = bool((rows >= -1e6).any(dim=-1).all())
                for kt in range(nk):
                    tile = rows[:, kt * KB:(kt + 1) * KB]
                    want = 0 if (bool((tile <= -1e9).all()) and rows_ok) else 2 if bool((tile == 0).all()) else 1
                    assert int(flags[b, qb, kt]) == want, (b, qb, kt)


def test_tile_orders_are_bijective_for_any_grid(tmp_path):
    """csrc/ql_common.h: xcd_tile / xcd_tile_super map a workgroup id to an output tile; every tile of the grid must be produced exactly
    once for ANY (column tiles, row tiles) - round 4 opened the grouped order to column counts that are not multiples of 8 (w_in: 107).
    The two functions are __host__ __device__: compiled for the host by hipcc (no GPU needed) and swept here."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = tmp_path / "tiles.hip"
    src.write_text('''
#include "%s/chatglm_q_amd/csrc/ql_common.h"
#include <cstdio>
#include <vector>
int main() {
    long bad = 0, grids = 0;
    for (int nbx = 1; nbx <= 130; ++nbx)
        for (int nby = 1; nby <= 40; ++nby) {
            const unsigned total = (unsigned)(nbx * nby);
            for (int mode = 0; mode < 4; ++mode) {                      // row-major, column-major, groups of 4 rows, groups of 2 rows
                std::vector<char> seen(total, 0);
                for (unsigned id = 0; id < total; ++id) {
                    const ql::TileXY t = mode == 0 ? ql::xcd_tile(id, total, nbx) : mode == 1 ? ql::xcd_tile(id, total, -nbx)
                                                   : ql::xcd_tile_super(id, total, nbx, mode == 2 ? 4 : 2);
                    if (t.x < 0 || t.x >= nbx || t.y < 0 || t.y >= nby || seen[t.y * nbx + t.x]++) ++bad;
                }
                ++grids;
            }
        }
    // what the grouped order is for: the 32 tiles an XCD runs together (consecutive positions of one XCD) span few operand panels
    const unsigned total = 107 * 32;
    int worst = 0;
    for (unsigned first = 0; first + 32 * 8 <= total; first += 32 * 8) {   // ids first + 8 i + c, i < 32: 32 consecutive positions of XCD c
        bool col[107] = {false}, row[32] = {false};
        for (unsigned i = 0; i < 32; ++i) { const ql::TileXY t = ql::xcd_tile_super(first + 8 * i, total, 107, 4); col[t.x] = row[t.y] = true; }
        int panels = 0;
        for (bool b : col) panels += b;
        for (bool b : row) panels += b;
        if (panels > worst) worst = panels;
    }
    printf("%%ld %%ld %%d\\n", bad, grids, worst);
    return 0;
}
''' % ROOT)
    exe = tmp_path / "tiles.bin"
    subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", str(src), "-o", str(exe)], check=True, capture_output=True)
    bad, grids, worst = (int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split())
    assert bad == 0 and grids == 130 * 40 * 4
    assert worst <= 13          # 8 or 9 columns x 4 rows (+ a group boundary): 33 with whole rows
