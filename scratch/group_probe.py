"""Timing probes for the group statistics kernels: which outputs / grid sizes cost what (16 x 4096^2 fp16, int8 sym)."""
import os, sys, torch
sys.path.insert(0, '.')
from quantization_amd import _lib
from quantization_amd.fused import quantize_batch
torch.cuda.set_device(0)
g = torch.Generator(device='cuda').manual_seed(0)
w = (torch.randn(16, 4096, 4096, generator=g, device='cuda') * 0.02).half()
def t(gs, **outs):
    res = None
    for _ in range(3):
        res = quantize_batch(w, _lib.INT_SYM, 8, gs, None, out=res, **outs)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            res = quantize_batch(w, _lib.INT_SYM, 8, gs, None, out=res, **outs)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    return best
for gs in (128, 32, None):
    full = t(gs, deq=False, codes=False, scale=True, seg_mse=True, totals=True)
    nosc = t(gs, deq=False, codes=False, scale=False, seg_mse=True, totals=True)
    noseg = t(gs, deq=False, codes=False, scale=False, seg_mse=False, totals=True)
    print(f"MQ_MAX_BLOCKS={os.environ.get('MQ_MAX_BLOCKS')} gs={gs}: scale+seg_sse+totals {full:6.1f} us | seg_sse+totals {nosc:6.1f} | totals only {noseg:6.1f}", flush=True)
