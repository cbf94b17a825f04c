import sys, torch
sys.path.insert(0, '.')
from quantization_amd import _lib
from quantization_amd.fused import quantize_batch
torch.cuda.set_device(0)
g = torch.Generator(device='cuda').manual_seed(0)
w = (torch.randn(8, 4096, 4096, generator=g, device='cuda') * 0.02).half()
res = None
for _ in range(4):
    res = quantize_batch(w, _lib.LUT_FP, 8, None, "fp8_e4m3", deq=False, codes=False, abs_err=torch.float32, p99=99, out=res)
torch.cuda.synchronize()
