"""Cost of the numpy-order modes: 16 x 4096^2 bf16, int8, per-channel and g = 128 / 256, list entries in float32
(bandwidth kernels) against numpy's order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantization_amd import _lib
from quantization_amd.fused import quantize_batch
w = (torch.randn(16, 4096, 4096, device="cuda") * 0.02).bfloat16()
for gs in (None, 128, 256, 1024):
    for mode in (True, "numpy"):
        f = lambda: quantize_batch(w, _lib.INT_SYM, 8, gs, deq=False, codes=False, scale=False, seg_mse=mode, totals=True)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        print(f"g={gs} seg_mse={mode}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us per 16 x 4096^2", flush=True)
