from . import qlinear, quantizer  # noqa: F401
from .qlinear import DynamicQuantizeLinear, QEmbedding, dynamic_quant_matmul  # noqa: F401
from .quantizer import quantize_int8  # noqa: F401
