"""ctypes binding of libqlinear_hip.so (C ABI in include/qlinear_hip.h).

The library is built in-tree (``make -C chatglm_q_amd/csrc`` or ``__graft_entry__.build()``).
There is no software fallback for device tensors: if the library is missing, every op that
receives a GPU tensor raises ``QLinearLibraryMissing``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# QLINEAR_LIB_PATH: developer override (tools/: ablation builds of the same library); the package default is the in-tree build
LIB_PATH = os.environ.get("QLINEAR_LIB_PATH") or os.path.join(_HERE, "csrc", "libqlinear_hip.so")
# product + experiments + tuning knobs as environment variables; QLINEAR_DEV_LIB_PATH: another developer build (tools/ab/)
DEV_LIB_PATH = os.environ.get("QLINEAR_DEV_LIB_PATH") or os.path.join(_HERE, "csrc", "libqlinear_hip_dev.so")
ABI_VERSION = 3

DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}

FLAG_STRICT_ROUNDING = 1
FLAG_ACT_PER_TENSOR = 2

# Per-weight rounding policy of the one-to-four-row int4 / int8 kernels (the MFMA and canonical kernels always round like the
# reference).  QLINEAR_STRICT=1: the reference's rounding sequence bit for bit in every kernel; =0: never (exact-dequant
# arithmetic everywhere); unset ("auto"): strict for bf16 - skipping a rounding worth 2^-9 / sqrt(3) per weight cannot land
# within 1e-3 of a reference that performs it - and exact-dequant for fp16 / fp32 (1.6e-4 measured at 1 x 4096 -> 4096).
_STRICT_ENV = os.environ.get("QLINEAR_STRICT", "auto").strip().lower()
STRICT_MODE = "auto" if _STRICT_ENV in ("auto", "") else ("off" if _STRICT_ENV in ("0", "false") else "on")
STRICT_DEFAULT = STRICT_MODE == "on"


def strict_for(dtype) -> bool:
    """Whether a few-row call on ``dtype`` activations takes the reference's per-weight rounding under the current policy."""
    return STRICT_MODE == "on" or (STRICT_MODE == "auto" and dtype == torch.bfloat16)


PRO_SILU = 1
PRO_ADDNORM = 2
NEXT_W4G32_PACKED, NEXT_W8_ROWS = 1, 2     # qlinear_decode_attention_rope_prefetch
EPI_SILU_GATE = 0x100
FUSED_STRICT = 0x200            # QL_FUSED_STRICT: OR-ed into the prologue code of qlinear_w4g32_fwd_packed_fused
ERR_UNSUPPORTED = -7            # QL_ERR_UNSUPPORTED

OP_W4G32_FWD = 1
OP_W4G32_FWD_PACKED = 2
OP_W8_FWD = 3
OP_W8A8_FWD = 4
OP_W8_FWD_TILED = 5
OP_W8A8_LINEAR_TILED = 6
OP_W4A8_LINEAR = 7

EXPORTS = {
    # name: (restype, argtypes)
    "qlinear_abi_version": (c_int, []),
    "qlinear_status_string": (c_char_p, [c_int]),
    "qlinear_launch_count": (c_uint64, []),
    "qlinear_last_dispatch": (c_uint64, []),
    "qlinear_dispatch_reset": (None, []),
    "qlinear_dispatch_reload": (None, []),
    "qlinear_dispatch_flags": (ctypes.c_uint, []),
    "qlinear_workspace_bytes": (c_size_t, [c_int, c_int64, c_int64, c_int64, c_int64]),
    "qlinear_w4g32_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                  c_int64, c_int64, c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "qlinear_w4g32_bwd_input": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64,
                                        c_int64, c_int, c_void_p]),
    "qlinear_w4g32_packed_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int]),
    "qlinear_w4g32_repack": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "qlinear_w4g32_rows_on_tiled": (c_int, [c_int64, c_int64, c_int64, c_int, c_int]),
    "qlinear_w4g32_packed_dispatch": (c_int, [c_int64, c_int64, c_int64, c_int, c_int]),
    "qlinear_gated_serves": (c_int, [c_int64, c_int64, c_int64, c_int, c_int]),
    "qlinear_w4g32_gemv_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int]),
    "qlinear_w4g32_tiled_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int]),
    "qlinear_w4g32_repack_gemv": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "qlinear_w4g32_unpack_gemv": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "qlinear_w4g32_tile": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "qlinear_w4g32_fwd_tiled": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64,
                                        c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "qlinear_w4g32_fwd_tiled256": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64,
                                           c_int, c_void_p]),
    "qlinear_gemm256_serves": (c_int, [c_int64, c_int64, c_int64]),
    "qlinear_tiled_dispatch": (c_int, [c_int, c_int64, c_int64, c_int64, c_void_p]),
    "qlinear_w4g32_fwd_tiled_gated": (c_int, [c_void_p] * 4 + [c_int64] * 5 + [c_int, c_void_p]),
    "qlinear_w4g32_fwd_packed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                         c_int64, c_int64, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "qlinear_w4g32_fwd_packed_fused": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                                               c_void_p, c_void_p, c_float, c_int, c_void_p]),
    "qlinear_w8_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                               c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "qlinear_w8_tiled_bytes": (c_size_t, [c_int64, c_int64]),
    "qlinear_w8_tile": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "qlinear_w8_fwd_tiled256": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64,
                                        c_int, c_void_p]),
    "qlinear_w8_fwd_tiled": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64,
                                     c_int, c_void_p, c_size_t, c_void_p]),
    "qlinear_w8_fwd_fused": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                     c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p]),
    "qlinear_w8_bwd_input": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64,
                                     c_int, c_void_p]),
    "qlinear_act_quant_i8_rowwise": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "qlinear_act_quant_i8": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p]),
    "qlinear_w8a8_fwd_tiled": (c_int, [c_void_p] * 6 + [c_int64] * 4 + [c_int, c_void_p]),
    "qlinear_w8a8_fwd_tiled_gated": (c_int, [c_void_p] * 6 + [c_int64] * 4 + [c_int, c_void_p]),
    "qlinear_w8a8_fwd_tiled256": (c_int, [c_void_p] * 6 + [c_int64] * 4 + [c_int, c_void_p]),
    "qlinear_w8a8_linear_tiled": (c_int, [c_void_p] * 5 + [c_int64] * 5 + [c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "qlinear_w8a8_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                 c_int64, c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "qlinear_qembedding_w4": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                      c_int, c_void_p]),
    "qlinear_qembedding_w8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int,
                                      c_void_p]),
    "qlinear_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_float, c_int,
                                c_void_p]),
    "qlinear_add_rmsnorm": (c_int, [c_void_p] * 5 + [c_int64, c_int64, c_int64, c_float, c_int, c_void_p]),
    "qlinear_rope_kv_write": (c_int, [c_void_p] * 7 + [c_int64] * 7 + [c_int, c_void_p]),
    "qlinear_decode_attention": (c_int, [c_void_p] * 5 + [c_int64] * 5 + [c_int, c_void_p]),
    "qlinear_decode_attention_split_bytes": (c_size_t, [c_int64] * 4),
    "qlinear_decode_attention_rope": (c_int, [c_void_p] * 8 + [c_int64] * 6 + [c_int, c_void_p, c_size_t, c_void_p]),
    "qlinear_w4g32_fwd_packed_gated": (c_int, [c_void_p] * 4 + [c_int64] * 5 + [c_int, c_void_p]),
    "qlinear_w4g32_fwd_rows_fused": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                             c_void_p, c_float, c_int, c_void_p]),
    "qlinear_w4g32_fwd_packed_residual": (c_int, [c_void_p] * 5 + [c_int64, c_int64, c_int, c_int, c_void_p]),
    "qlinear_w8_fwd_residual": (c_int, [c_void_p] * 6 + [c_int64, c_int64, c_int64, c_int, c_void_p]),
    "qlinear_decode_attention_rope_prefetch": (c_int, [c_void_p] * 8 + [c_int64] * 6 + [c_int, c_void_p, c_size_t, c_void_p, c_int,
                                                        c_int64, c_int64, c_void_p]),
    "qlinear_greedy_advance": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                       c_int, c_void_p]),
    "qlinear_top_p_sample": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "qlinear_masked_softmax": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64,
                                       c_int, c_void_p]),
    "qlinear_silu_mul": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "qlinear_w4g32_fwd_tiled_residual": (c_int, [c_void_p] * 5 + [c_int64] * 6 + [c_int, c_void_p]),
    "qlinear_w8_fwd_tiled_gated": (c_int, [c_void_p] * 5 + [c_int64] * 5 + [c_int, c_void_p]),
    "qlinear_w8_fwd_tiled_residual": (c_int, [c_void_p] * 6 + [c_int64] * 6 + [c_int, c_void_p]),
    "qlinear_prefill_attention_tiles": (c_int, [c_void_p, c_void_p]),
    "qlinear_prefill_attention": (c_int, [c_void_p] * 6 + [c_int64] * 8 + [c_int, c_void_p]),
    "qlinear_rmsnorm_quant_i8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                         c_int64, c_float, c_int, c_void_p]),
    "qlinear_silu_mul_quant_i8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p]),
}

# entry points of libqlinear_hip_dev.so only (include/qlinear_hip_dev.h): recorded experiments, chatglm_q_amd/dev/
DEV_EXPORTS = {
    "qlinear_w4a8_packed_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int]),
    "qlinear_w4a8_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "qlinear_w4a8_fwd": (c_int, [c_void_p] * 5 + [c_int64] * 4 + [c_int, c_void_p]),
    "qlinear_w4a8_linear": (c_int, [c_void_p] * 4 + [c_int64] * 5 + [c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "qlinear_w4g32_mlp_pair_workspace_bytes": (c_size_t, []),
    "qlinear_w4g32_mlp_pair": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "qlinear_w4g32_mlp_engine_workspace_bytes": (c_size_t, [c_int64]),
    "qlinear_w4g32_mlp_engine_supported": (c_int, [c_int64, c_int64, c_int64]),
    "qlinear_w4g32_mlp_engine": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64,
                                         c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "qlinear_dev_dense256_image_bytes": (c_size_t, [c_int64, c_int64]),
    "qlinear_dev_dense256_expand": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "qlinear_dev_w8a8_splitk_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "qlinear_dev_w8a8_fwd_tiled_splitk": (c_int, [c_void_p] * 6 + [c_int64] * 4 + [c_int, c_void_p, c_size_t, c_void_p]),
    "qlinear_dev_dense256_fwd": (c_int, [c_void_p] * 5 + [c_int64] * 6 + [c_int, c_int, c_void_p]),
}


class QLinearLibraryMissing(RuntimeError):
    pass


_lib = None
_load_error: Exception | None = None
_load_stamp = None      # (exists, mtime) of LIB_PATH when the last attempt failed


def _lib_stamp():
    try:
        return (True, os.stat(LIB_PATH).st_mtime_ns)
    except OSError:
        return (False, 0)


# Which kernel a call takes - and so which derived layout it reads - follows QLINEAR_DISPATCH as the library parsed it; a reload
# (qlinear_dispatch_reload: tests, A/B tools) can change it under every routing answer Python has cached and under every pre-bound
# launch.  The handle's `qlinear_dispatch_reload` attribute is wrapped so that a reload bumps this counter; plans compare it per call
# (a Python int), cached routing answers are keyed on the flags themselves (int4/hip_ops.py).  ADVICE r5 (medium).
_dispatch_epoch = [0]


def dispatch_epoch() -> int:
    return _dispatch_epoch[0]


def _wrap_dispatch_reload(lib):
    raw = lib.qlinear_dispatch_reload

    def qlinear_dispatch_reload():
        raw()
        _dispatch_epoch[0] += 1

    lib.qlinear_dispatch_reload = qlinear_dispatch_reload


def _try_load():
    """dlopen once; a FAILED attempt is retried when the file appears or changes (e.g. the package was imported
    before ``__graft_entry__.build()`` ran in the same process)."""
    global _lib, _load_error, _load_stamp
    if _lib is not None:
        return
    stamp = _lib_stamp()
    if _load_error is not None and stamp == _load_stamp:
        return
    _load_error, _load_stamp = None, stamp
    try:
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        ver = lib.qlinear_abi_version()
        if ver != ABI_VERSION:
            raise OSError(f"{LIB_PATH}: ABI version {ver}, expected {ABI_VERSION}")
        _wrap_dispatch_reload(lib)
        _lib = lib
    except (OSError, AttributeError) as e:  # missing file, missing dependency or missing symbol
        _load_error = e


def available() -> bool:
    """True when libqlinear_hip.so is built and loads in this process."""
    _try_load()
    return _lib is not None


def get_lib():
    _try_load()
    if _lib is None:
        raise QLinearLibraryMissing(
            f"libqlinear_hip.so is not available ({_load_error}). Build it with "
            f"`make -C {os.path.join(_HERE, 'csrc')}` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no software fallback for GPU tensors.")
    return _lib


_dev_lib = None


def dev_available() -> bool:
    """True when the developer library (``make -C chatglm_q_amd/csrc dev``) is built."""
    return os.path.exists(DEV_LIB_PATH)


def get_dev_lib():
    """libqlinear_hip_dev.so: everything the product library exports plus ``DEV_EXPORTS`` (chatglm_q_amd/dev/)."""
    global _dev_lib
    if _dev_lib is None:
        try:
            lib = ctypes.CDLL(DEV_LIB_PATH)
            for name, (res, args) in {**EXPORTS, **DEV_EXPORTS}.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            if lib.qlinear_abi_version() != ABI_VERSION:
                raise OSError(f"{DEV_LIB_PATH}: ABI version {lib.qlinear_abi_version()}, expected {ABI_VERSION}")
        except (OSError, AttributeError) as e:
            raise QLinearLibraryMissing(f"libqlinear_hip_dev.so is not available ({e}); build it with "
                                        f"`make -C {os.path.join(_HERE, 'csrc')} dev`") from None
        _wrap_dispatch_reload(lib)
        _dev_lib = lib
    return _dev_lib


def check(status: int, what: str):
    if status == 0:
        return
    msg = get_lib().qlinear_status_string(status).decode()
    if status < 0:
        raise ValueError(f"{what}: {msg} (status {status})")
    raise RuntimeError(f"{what}: HIP error {status}: {msg}")


def launch_count() -> int:
    return int(get_lib().qlinear_launch_count())


def dtype_code(dtype: torch.dtype) -> int:
    try:
        return DTYPE_CODE[dtype]
    except KeyError:
        raise TypeError(f"unsupported activation dtype {dtype}; expected float32, float16 or bfloat16") from None


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# ---- pre-bound launches (the module call must not cost 10x its kernel) -----------------------------------
# A plan is built by the CHECKED path after it served a call: every argument of the C entry point that cannot change while the
# module's buffers stay put is converted to its ctypes object once; per call only the activation pointer, the output pointer
# and the stream change.  The plan re-validates cheaply what can change behind its back (buffer identity + version counters,
# shape / dtype / device / layout of the input) and returns None to send the call back to the checked path.
_raw_stream = torch._C._cuda_getCurrentRawStream if hasattr(torch._C, "_cuda_getCurrentRawStream") else None
_cur_device = torch._C._cuda_getDevice if hasattr(torch._C, "_cuda_getDevice") else None


def _version_of(t):
    try:
        return t._version
    except RuntimeError:                # inference tensors keep no version counter: identity only (see buffer_key)
        return None


def make_plan(name: str, values, a_slot: int, c_slot: int, st_slot: int, rows: int, in_cols: int, n_cols: int, dtype, device,
              guards, ws_slot: int | None = None, ws_bytes: int = 0, keep=(), extras=(), dev: bool = False):
    """Closure ``run(input, *extra_tensors) -> Tensor | None`` around the C entry point ``name``.  ``values``: one Python
    value per argument (``EXPORTS[name]``), with None in the variable slots; ``guards``: the tensors whose identity and
    version the plan depends on (canonical buffers, bias, norm weights); ``keep``: tensors whose ADDRESSES are baked into
    ``values`` (derived layouts); ``extras``: ((slot, numel), ...) of further per-call operands in the activation dtype
    (residual, delta, hout), each passed to ``run`` as a tensor or None."""
    if _raw_stream is None or _cur_device is None:
        return None
    lib = get_dev_lib() if dev else get_lib()
    res, argtypes = (DEV_EXPORTS if dev else EXPORTS)[name]
    fn = lib._FuncPtr((name, lib))      # a private pointer object: no argtypes, arguments arrive as ctypes objects
    fn.restype = res
    tmpl = tuple(t(v) for t, v in zip(argtypes, values))
    idx = device.index if device.index is not None else torch.cuda.current_device()
    numel = rows * in_cols
    g = tuple((t, _version_of(t), t.data_ptr()) for t in guards if t is not None)
    kept = tuple(keep)
    empty, c_void = torch.empty, c_void_p
    uint8 = torch.uint8
    extras = tuple(extras)
    single_device = torch.cuda.device_count() == 1          # then the current device cannot be another one
    # the three usual guards (weight, scale, bias) unrolled; identity first (a replaced storage), then the version counter
    (g0, v0, p0), (g1, v1, p1), (g2, v2, p2) = (g + ((None, None, 0),) * 3)[:3]
    g_rest = g[3:]
    shape_memo = [None, None]                               # last input shape seen -> its output shape
    epoch0, epoch = _dispatch_epoch[0], _dispatch_epoch     # the routing this plan was built under

    def run(x, *ex):
        if epoch[0] != epoch0:                              # QLINEAR_DISPATCH reloaded: another kernel / layout may serve this call now
            return None
        if (x.dtype is not dtype or x.numel() != numel or x.get_device() != idx
                or not x.is_contiguous() or (x.requires_grad and torch.is_grad_enabled())):
            return None
        shp = x.shape
        if shp != shape_memo[0]:
            if shp[-1] != in_cols:
                return None
            shape_memo[0], shape_memo[1] = shp, shp[:-1] + (n_cols,)
        if g0 is not None and (g0.data_ptr() != p0 or (v0 is not None and g0._version != v0)):
            return None
        if g1 is not None and (g1.data_ptr() != p1 or (v1 is not None and g1._version != v1)):
            return None
        if g2 is not None and (g2.data_ptr() != p2 or (v2 is not None and g2._version != v2)):
            return None
        for t, ver, p in g_rest:
            if t.data_ptr() != p or (ver is not None and t._version != ver):
                return None
        a_ptr = x.data_ptr()
        if a_ptr & 15:
            return None
        args = list(tmpl)
        for (slot, n), e in zip(extras, ex):
            if e is not None:
                ep = e.data_ptr()
                if e.dtype is not dtype or e.numel() != n or not e.is_contiguous() or e.get_device() != idx or ep & 15:
                    return None
                args[slot] = c_void(ep)
        out = empty(shape_memo[1], dtype=dtype, device=device)
        args[a_slot] = c_void(a_ptr)
        args[c_slot] = c_void(out.data_ptr())
        args[st_slot] = c_void(_raw_stream(idx))
        if ws_slot is not None:
            ws = empty(ws_bytes, dtype=uint8, device=device)    # split-K slabs of the few-row kernels: per call, like the checked path
            args[ws_slot] = c_void(ws.data_ptr())
        if single_device or _cur_device() == idx:
            st = fn(*args)
        else:
            with torch.cuda.device(idx):
                st = fn(*args)
        if st:
            check(st, name)
        return out

    run.kept = kept
    return run


def ptr(t: torch.Tensor | None):
    return None if t is None else t.data_ptr()


_layout_epoch = 0


def layout_epoch() -> int:
    """Process-wide count of derived-layout builds and drops (every module bumps it): a captured HIP graph that baked a
    derived buffer's address in compares it with the value at capture time (chatglm_q_amd/decoder.py)."""
    return _layout_epoch


def bump_layout_epoch() -> None:
    global _layout_epoch
    _layout_epoch += 1


def buffer_key(*tensors):
    """Identity + version of the canonical buffers a derived (non-persistent) layout was built from.

    ``Tensor._version`` notices every in-place op that goes through autograd's bookkeeping - the reference loader's
    ``state_dict()[k].copy_(...)`` (chatglm_q/loader.py:103), ``apply_weights_``, ``.to()`` (new storage).  It does NOT
    notice writes through ``.data`` / raw pointers, and inference tensors (created under ``torch.inference_mode()``) have
    no version counter at all: for those the key falls back to identity only, and whoever rewrites such a buffer in
    place must call the module's ``invalidate()``."""
    key = []
    for t in tensors:
        if t is None:
            key.append(None)
            continue
        try:
            ver = t._version
        except RuntimeError:            # "Inference tensors do not track version counter"
            ver = -1
        key.append((t.data_ptr(), ver, t.dtype, t.device, tuple(t.shape)))
    return tuple(key)
