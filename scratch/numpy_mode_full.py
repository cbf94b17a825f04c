"""The whole Llama-3-8B sweep in numpy_order mode (every CSV number in numpy's bits) against the bandwidth mode:
time per step, and the two modes' records side by side (they must agree to 1e-6; mse_p99 to the list entries'
float32 rounding)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from quantization_amd.report import SweepPlan
torch.cuda.set_device(0)
wl = bench.ModelSweep("llama8b", torch.device("cuda", 0), 0, 1, "normal")
fast = SweepPlan(wl.weights, wl.passes)
exact = SweepPlan(wl.weights, wl.passes, numpy_order=True)
for name, plan in (("bandwidth mode", fast), ("numpy_order", exact)):
    plan.run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        plan.run()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e3:.1f} ms per step", flush=True)
a, b = fast.run().clone(), exact.run().clone()
mt = exact.mse_tensor
torch.cuda.synchronize()
for col, what in ((1, "sse"), (3, "mse_rep"), (4, "mse_p99")):
    rel = ((a[..., col] - b[..., col]).abs() / b[..., col].abs()).max().item()
    print(f"{what}: largest relative difference between the modes {rel:.3g}", flush=True)
rel = ((a[..., 1] / a[..., 0] - mt).abs() / mt.abs()).max().item()
print(f"mse_tensor: sse / numel against numpy's float32 mean, largest relative difference {rel:.3g}", flush=True)
numel = a[0, :, 0]
for n in sorted(set(numel.tolist())):
    m = numel == n
    r = ((a[:, m, 1] / a[:, m, 0] - mt[:, m]).abs() / mt[:, m].abs()).max().item()
    print(f"  tensors of {int(n)} elements: {r:.3g}", flush=True)
