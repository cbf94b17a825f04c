"""mq_error_stats throughput: two stored 8192 x 8192 tensors, totals + row sums (+ abs_err dump)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantization_amd import error_metrics as em
g = torch.Generator(device="cuda").manual_seed(0)
for dt in (torch.float16, torch.float32):
    a = torch.randn(8192, 8192, generator=g, device="cuda").to(dt)
    b = (a.float() + 0.01 * torch.randn(8192, 8192, generator=g, device="cuda")).to(dt)
    for kw in (dict(seg_len=8192), dict(seg_len=128), dict(seg_len=128, abs_err=torch.float32), dict(cols=True)):
        for _ in range(3): em.error_stats(a, b, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): em.error_stats(a, b, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        byts = 2 * a.numel() * a.element_size() + (4 * a.numel() if kw.get("abs_err") else 0)
        print(f"{dt} {kw}: {us:8.1f} us  {byts / us / 1e6:6.2f} TB/s")
