"""Per-rank step time of the sharded sweep without the collectives: ModelSweep(rank 0 of W) on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
for W in (1, 2, 4, 8):
    wl = bench.ModelSweep(sys.argv[1] if len(sys.argv) > 1 else "llama8b", torch.device("cuda", 0), 0, W, "normal")
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        wl.step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"world {W}: rank 0 holds {len(wl.mine)} layers, {wl.local_weight_bytes / 1e9:.1f} GB per step: {t / n * 1e3:.2f} ms per step "
          f"(host enqueue {t_host / n * 1e3:.2f} ms) -> {wl.total_weight_bytes / (t / n) / 1e9:.0f} GB/s if all ranks take this long", flush=True)
    del wl
    torch.cuda.empty_cache()
