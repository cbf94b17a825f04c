"""quantization_amd -- MI355X-native weight quantize -> dequantize -> error (MSE / P99) path.

Layout:
  csrc/            hand-written gfx950 kernels and the C ABI (include/mq.h) -> csrc/libmq.so
  _lib.py          ctypes binding of libmq.so (fails loudly when the library is missing)
  fused.py         host side of the fused pass
  quant_weight.py  the reference's quant_utils/quant_weight.py API on top of it
  write_results.py the reference's quant_utils/write_results.py
"""
from . import quant_weight  # noqa: F401
from .fused import QuantResult, quantize  # noqa: F401
from .write_results import write_results  # noqa: F401

__version__ = "0.1.0"
