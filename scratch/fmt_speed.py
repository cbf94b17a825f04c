import sys, torch
sys.path.insert(0, '.')
import quantization_amd as qa
from quantization_amd import _lib
from quantization_amd.fused import quantize_batch
torch.cuda.set_device(0)
g = torch.Generator(device='cuda').manual_seed(0)
w = (torch.randn(16, 4096, 4096, generator=g, device='cuda') * 0.02).half()
def t(fmt, dt, gs, bits=8, **outs):
    res = None
    for _ in range(2):
        res = quantize_batch(w, fmt, bits, gs, dt, out=res, **outs)
    best = 1e9
    for _ in range(3):   # best of 3 runs of 20 launches (clocks move between short runs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            res = quantize_batch(w, fmt, bits, gs, dt, out=res, **outs)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    return best
stats = dict(deq=False, codes=False, scale=True, seg_sse=True, totals=True)
api = dict(deq=True, codes=True, scale=True)
for name, fmt, dt, bits in [("int8", _lib.INT_SYM, None, 8), ("int8_asym", _lib.INT_ASYM, None, 8), ("int4", _lib.INT_SYM, None, 4),
                            ("fp8_e4m3", _lib.LUT_FP, "fp8_e4m3", 8), ("fp8_e2m5", _lib.LUT_FP, "fp8_e2m5", 8),
                            ("fp4", _lib.LUT_FP, "fp4", 4), ("fp6_e3m2", _lib.LUT_FP, "fp6_e3m2", 6)]:
    for gs in (None, 128, 64, 32):
        a = t(fmt, dt, gs, bits, **stats)
        b = t(fmt, dt, gs, bits, **api) if gs in (None, 128) else float("nan")
        print(f"{name:10s} gs={gs}: stats-only {a:7.1f} us ({w.numel()*2/a/1e6:6.2f} TB/s of weights)   deq+codes {b:7.1f} us ({w.numel()*2/b/1e6:6.2f} TB/s)", flush=True)
