"""Host-side profile of one model-sweep step (argv: workload)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
wl = bench.ModelSweep(sys.argv[1], torch.device("cuda", 0), 0, 1, "normal")
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    wl.step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
