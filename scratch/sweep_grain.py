"""Per-rank step time of the sharded sweep for one setting of the environment (MQ_SWEEP_SINGLE, MQ_SWEEP_WG_KB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
name = sys.argv[1] if len(sys.argv) > 1 else "llama8b"
for W in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,8").split(",")]:
    wl = bench.ModelSweep(name, torch.device("cuda", 0), 0, W, "normal")
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            wl.step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    nl = sum(len(ps["launches"]) for ps in wl.plan.plan)
    print(f"single={os.environ.get('MQ_SWEEP_SINGLE', '1')} wg_kb={os.environ.get('MQ_SWEEP_WG_KB', '64')} world {W}: "
          f"{len(wl.mine)} layers, {nl} launches per step, {best * 1e3:.3f} ms per step", flush=True)
    del wl
    torch.cuda.empty_cache()
