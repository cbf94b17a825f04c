"""Does it pay to run the sweep's main stream at high priority (the selections' second stream stays normal)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
for W in (1, 8):
    wl = bench.ModelSweep("llama8b", torch.device("cuda", 0), 0, W, "normal")
    wl.step()
    hi = torch.cuda.Stream(priority=-1)
    def step_hi():
        with torch.cuda.stream(hi):
            return wl.step()
    res = {"default stream": [], "high-priority stream": []}
    for rep in range(4):
        for name, fn in (("default stream", wl.step), ("high-priority stream", step_hi)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"world {W}: " + "; ".join(f"{k} {min(v):.3f} ms ({' '.join(f'{x:.3f}' for x in v)})" for k, v in res.items()), flush=True)
    del wl
    torch.cuda.empty_cache()
