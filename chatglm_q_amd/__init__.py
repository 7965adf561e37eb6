"""chatglm_q_amd - MI355X-native (gfx950) quantized-linear forward path.

Drop-in replacement for the ``DynamicQuantizeLinear`` + Triton ``dynamic_quant_matmul`` path of
K024/chatglm-q (``chatglm_q.int4`` / ``chatglm_q.int8``): same module API and buffer format, with
the device work done by hand-written HIP kernels behind a C ABI (``include/qlinear_hip.h``,
``chatglm_q_amd/csrc``).  See DESIGN.md and INTEGRATION.md.
"""
from . import _lib  # noqa: F401

__all__ = ["int4", "int8"]
__version__ = "0.1.0"
