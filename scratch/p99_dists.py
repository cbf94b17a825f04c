import sys, torch, numpy as np
sys.path.insert(0, '.')
from quantization_amd import _lib
from quantization_amd.fused import quantize_batch
torch.cuda.set_device(0)
g = torch.Generator(device='cuda').manual_seed(0)
base = torch.randn(8, 4096, 4096, generator=g, device='cuda')
dists = {"normal": base * 0.02,
         "student_t3": torch.distributions.StudentT(3.0).sample((8, 4096, 4096)).cuda() * 0.02,
         "lognormal": torch.exp(base * 2) * 1e-3 * torch.sign(torch.randn(8, 4096, 4096, generator=g, device='cuda')),
         "outlier_cols": base * 0.02 * (1 + 30 * (torch.rand(1, 1, 4096, generator=g, device='cuda') < 0.01))}
for name, w in dists.items():
    w = w.half()
    for fmt, dt, gs in ((_lib.INT_ASYM, None, 128), (_lib.INT_SYM, None, None), (_lib.LUT_FP, "fp8_e4m3", None)):
        res = None
        for _ in range(2):
            res = quantize_batch(w, fmt, 8, gs, dt, deq=False, codes=False, abs_err=torch.float32, p99=99, out=res)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            res = quantize_batch(w, fmt, 8, gs, dt, deq=False, codes=False, abs_err=torch.float32, p99=99, out=res)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        want = np.float32(np.percentile(res.abs_err[0].cpu().numpy().ravel(), 99))
        ok = res.abs_p99[0].cpu().numpy().tobytes() == want.tobytes()
        print(f"{name:13s} fmt={fmt} gs={gs}: {us:7.1f} us per 8x4096^2 call, p99 exact={ok}")
