"""A few steps of the numpy-order Llama-3-8B sweep for a kernel trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
wl = bench.ModelSweep("llama8b", torch.device("cuda", 0), 0, 1, "normal", numpy_order=True)
for _ in range(5): wl.step()
torch.cuda.synchronize()
