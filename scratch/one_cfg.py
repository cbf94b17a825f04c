"""One statistics-only configuration, a few launches: for counter collection (argv: group size or 0, format, mse flag)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _lib
from quantization_amd.fused import quantize_batch
gs = int(sys.argv[1]) or None
fmt = sys.argv[2] if len(sys.argv) > 2 else "int8"
torch.cuda.set_device(0)
g = torch.Generator(device='cuda').manual_seed(0)
w = (torch.randn(16, 4096, 4096, generator=g, device='cuda') * 0.02).half()
code = {"int8": _lib.INT_SYM, "int8_asym": _lib.INT_ASYM}.get(fmt, _lib.LUT_FP)
res = None
for _ in range(4):
    res = quantize_batch(w, code, 8, gs, fmt if code == _lib.LUT_FP else None, out=res, deq=False, codes=False, scale=False,
                         seg_mse=True, totals=True)
torch.cuda.synchronize()
