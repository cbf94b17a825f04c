"""rank 0 of 8 of the sharded sweep, a few steps: for a kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
wl = bench.ModelSweep("llama8b", torch.device("cuda", 0), 0, int(sys.argv[1]), "normal")
for _ in range(12):
    wl.step()
torch.cuda.synchronize()
