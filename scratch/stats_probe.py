"""A handful of statistics-only launches for profiling (rocprofv3 --kernel-trace / --pmc):
16 x 4096 x 4096 fp16 and bf16, sym int8 per-channel / g=128 / g=32, asym g=128, fp8_e4m3 per-channel."""
import sys
import torch
sys.path.insert(0, '.')
from quantization_amd import _lib
from quantization_amd.fused import quantize_batch
torch.cuda.set_device(0)
g = torch.Generator(device='cuda').manual_seed(0)
w16 = (torch.randn(16, 4096, 4096, generator=g, device='cuda') * 0.02).half()
wbf = w16.bfloat16()
stats = dict(deq=False, codes=False, scale=True, seg_sse=True, totals=True)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for w in (w16, wbf):
    for fmt, dt, gs in [(_lib.INT_SYM, None, None), (_lib.INT_SYM, None, 128), (_lib.INT_SYM, None, 32),
                        (_lib.INT_ASYM, None, 128), (_lib.LUT_FP, "fp8_e4m3", None), (_lib.LUT_FP, "fp8_e4m3", 128)]:
        res = None
        for _ in range(reps):
            res = quantize_batch(w, fmt, 8, gs, dt, out=res, **stats)
torch.cuda.synchronize()
