"""The step of one rank's share of the Llama-3-8B sweep at world sizes 1 .. 8, on one GPU (no collectives)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
for world in (1, 2, 4, 8):
    wl = bench.ModelSweep("llama8b", dev, 0, world, "normal")
    for _ in range(3): wl.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n): wl.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"world {world}: rank 0 holds {len(wl.mine)} layers, {wl.local_weight_bytes / len(wl.passes) / 1e9:.2f} GB; step {dt * 1e3:.3f} ms "
          f"-> {wl.local_weight_bytes / dt / 1e9:.0f} GB/s per rank, x{world} = {world * wl.local_weight_bytes / dt / 1e9:.0f}", flush=True)
    del wl
    torch.cuda.empty_cache()
