"""Throughput of the histogram scans on one MI355X (run by hand through gpurun)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quantization_amd import error_metrics as em

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for dt in (torch.float16, torch.float32):
    w = (torch.randn(16 * 4096 * 4096 // 4, device="cuda") * 0.02).repeat(4).to(dt)
    nbytes = w.numel() * w.element_size()
    t = timeit(lambda: em.minmax(w))
    print(f"minmax    {dt}: {t*1e3:8.1f} us  {nbytes/t/1e6:8.1f} GB/s")
    edges = np.linspace(-0.12, 0.12, 201)
    counts = torch.zeros(200, dtype=torch.int64, device="cuda")
    t = timeit(lambda: em.histogram(w, edges, counts=counts))
    print(f"histogram {dt}: {t*1e3:8.1f} us  {nbytes/t/1e6:8.1f} GB/s (normal weights, 200 bins)")
    a = w.abs().float() if dt == torch.float32 else None
    if a is not None:
        e2 = np.linspace(0, 0.12, 201)
        t = timeit(lambda: em.histogram(a, e2, counts=counts))
        print(f"histogram |x| f32: {t*1e3:8.1f} us  {nbytes/t/1e6:8.1f} GB/s (half-normal, 200 bins)")
