"""Does it pay to give the selections' second stream high priority?  (A/B in one process: the plan picks the
stream up from report._side_streams at every run.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from quantization_amd import report
torch.cuda.set_device(0)
normal, high = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
for W in (1, 2, 8):
    wl = bench.ModelSweep("llama8b", torch.device("cuda", 0), 0, W, "normal")
    wl.step()
    res = {"normal": [], "high": []}
    for rep in range(4):
        for name, st in (("normal", normal), ("high", high)):
            report._side_streams[0] = st
            for _ in range(3):
                wl.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                wl.step()
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"world {W}: second stream " + "; ".join(f"{k} {min(v):.3f} ms ({' '.join(f'{x:.3f}' for x in v)})" for k, v in res.items()), flush=True)
    del wl
    torch.cuda.empty_cache()
