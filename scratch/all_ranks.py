"""Step time of EVERY rank's share of the Llama-3-8B sweep at one world size, one after the other on one GPU (no
collectives): the job's step is the slowest rank's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for rank in range(W):
    wl = bench.ModelSweep("llama8b", torch.device("cuda", 0), rank, W, "normal")
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        wl.step()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / n
    kinds = sorted({wl.layers[i][1] for i in wl.mine})
    print(f"world {W} rank {rank}: {len(wl.mine)} layers {wl.local_weight_bytes / len(wl.passes) / 1e9:.3f} GB, "
          f"{len([la for ps in wl.plan.plan for la in ps['launches']])} launches: {t * 1e3:.3f} ms  shapes {kinds}", flush=True)
    del wl
    torch.cuda.empty_cache()
