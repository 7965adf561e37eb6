from . import qlinear, quantizer  # noqa: F401
from .qlinear import DynamicQuantizeLinear, QEmbedding, dynamic_quant_matmul, unpack_int4  # noqa: F401
from .quantizer import quantize_int4  # noqa: F401
