"""Group statistics kernels only: int8 / int8_asym / fp8_e4m3 x g = 128 / 64 / 32, fp16 and bf16, 16 x 4096^2."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _lib
from quantization_amd.fused import quantize_batch
torch.cuda.set_device(0)
g = torch.Generator(device='cuda').manual_seed(0)
w32 = torch.randn(16, 4096, 4096, generator=g, device='cuda') * 0.02
stats = dict(deq=False, codes=False, scale=True, seg_mse=True, totals=True)
def t(w, fmt, dt, gs, bits=8):
    res = None
    for _ in range(2):
        res = quantize_batch(w, fmt, bits, gs, dt, out=res, **stats)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            res = quantize_batch(w, fmt, bits, gs, dt, out=res, **stats)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    return best
for dname, w in (("fp16", w32.half()), ("bf16", w32.to(torch.bfloat16))):
    for name, fmt, dt in [("int8", _lib.INT_SYM, None), ("int8_asym", _lib.INT_ASYM, None), ("fp8_e4m3", _lib.LUT_FP, "fp8_e4m3"), ("fp8_e3m4", _lib.LUT_FP, "fp8_e3m4")]:
        print(f"{dname} {name:10s}" + "".join(f"  g={gs}: {t(w, fmt, dt, gs):6.1f} us" for gs in (None, 128, 64, 32)), flush=True)
