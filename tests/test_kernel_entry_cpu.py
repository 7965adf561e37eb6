"""Static check on the built library (no GPU): in the hot one-row kernels nothing may wait for a scalar load between a
wave's first instruction and its first weight load (issuing one is fine).  The kernel arguments those loads need are preloaded into SGPRs
(-amdgpu-kernarg-preload-count, csrc/Makefile) and the signatures are ordered for it; a late argument used early puts
an `s_load` + `s_waitcnt lgkmcnt(0)` in front of the weight stream of every wave (measured: 4.0 -> 4.4 us on the
headline shape).  The device code is taken from libqlinear_hip.so's fat binary and disassembled with llvm-objdump."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "chatglm_q_amd", "csrc", "libqlinear_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _device_code_objects(path):
    """ELF code objects for gfx950 inside the clang offload bundles of a host binary."""
    data = open(path, "rb").read()
    out, at = [], 0
    while True:
        at = data.find(MAGIC, at)
        if at < 0:
            return out
        n = struct.unpack_from("<Q", data, at + 24)[0]
        p = at + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(data[at + off:at + off + size])
        at += len(MAGIC)


def _disassemble(symbol_re, tmp_path):
    if not (os.path.exists(LIB) and os.path.exists(OBJDUMP)):
        pytest.skip("library or llvm-objdump not available")
    found = {}
    for i, blob in enumerate(_device_code_objects(LIB)):
        co = tmp_path / f"co{i}.elf"
        co.write_bytes(blob)
        syms = subprocess.run([OBJDUMP, "-t", str(co)], capture_output=True, text=True).stdout
        names = sorted({ln.split()[-1] for ln in syms.splitlines() if re.search(symbol_re, ln) and " F " in ln})
        for name in names:
            txt = subprocess.run([OBJDUMP, "-d", f"--disassemble-symbols={name}", str(co)], capture_output=True, text=True).stdout
            found[name] = [ln.split("//")[0].strip() for ln in txt.splitlines() if "\t" in ln]
    return found


def _body_after_preload_prologue(lines):
    """Instructions of the kernel proper: the backward-compatibility prologue (s_load of the preloaded arguments,
    s_waitcnt, s_branch over padding) that firmware with kernarg preload skips is dropped."""
    ops = [ln.split("\t")[-1].strip() if "\t" in ln else ln for ln in lines]
    for i, op in enumerate(ops[:40]):
        if op.startswith("s_branch"):
            rest = ops[i + 1:]
            while rest and rest[0].startswith("s_nop"):
                rest = rest[1:]
            return rest
    return ops


@pytest.mark.parametrize("pattern,what", [
    (r"w4_packed_gemv_16_kernelIDF16_Li1ELi\dELi\dELb0ELi0ELi0E", "int4g32 one-row forward (plain / residual)"),
    (r"w4_packed_gemv_16_kernelIDF16_Li1ELi\dELi\dELb0ELi0ELi3E", "int4g32 one-row forward with the RMSNorm prologue"),
    (r"w4_packed_gemv_16_kernelIDF16_Li1ELi\dELi\dELb0ELi0ELi2E", "int4g32 one-row forward with the add + RMSNorm prologue"),
    (r"w4_packed_gemv_16_kernelIDF16bLi1ELi\dELi\dELb0ELi0ELi[03]E", "int4g32 one-row forward, bf16"),
    (r"w8_gemv_f16_kernelILi1ELi\dELi\dELb0ELi0E", "int8 one-row forward (plain / residual)"),
    (r"w8_gemv_f16_kernelILi1ELi\dELi\dELb0ELi[23]E", "int8 one-row forward with the RMSNorm prologues"),
    (r"w4_fewrow_kernelIDF16", "int4g32 few-row forward"),
    (r"w8_fewrow_kernelIDF16", "int8 few-row forward"),
])
def test_no_scalar_load_wait_before_the_first_weight_load(pattern, what, tmp_path):
    kernels = _disassemble(pattern, tmp_path)
    assert kernels, f"no kernel matches {pattern}: the check would be vacuous"
    for name, lines in kernels.items():
        body = _body_after_preload_prologue(lines)
        first_nt = next((i for i, op in enumerate(body) if op.startswith("global_load_dwordx4") and " nt" in op), None)
        assert first_nt is not None, f"{what}: no streaming weight load found in {name}"
        # a scalar load may be ISSUED early (its latency then hides under the weight stream); it must not be WAITED for
        pending, stalls = [], []
        for op in body[:first_nt]:
            if op.startswith("s_load") or op.startswith("s_buffer_load"):
                pending.append(op)
            elif op.startswith("s_waitcnt") and "lgkmcnt" in op and pending:
                stalls.append((op, list(pending)))
                pending = []
        assert not stalls, f"{what} ({name}): waits for scalar loads before the first weight load: {stalls}"
