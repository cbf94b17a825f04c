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
from quantization_amd import report
from quantization_amd.fused import make_cfg, quantize
for gs in (None, 128):
    res = quantize(w[0], _lib.INT_SYM, 8, gs, deq=False, codes=False, scale=True, seg_mse=True)
    out = torch.zeros((), dtype=torch.float32, device="cuda")
    cfg = make_cfg(_lib.INT_SYM, 8, gs, None)
    f = lambda: report.tensor_mse_numpy(w[0], cfg, res.scale, None, out)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f"tensor_mse_numpy g={gs}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us per 4096^2", flush=True)
lists = [torch.rand(n, device="cuda") for n in (4096,) * 64 + (524288,) * 16]
f = lambda: report.numpy_list_means(lists)
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print(f"numpy_list_means, 64 lists of 4096 + 16 of 524288: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us", flush=True)
