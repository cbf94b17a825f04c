"""Two steps of the Llama-3-8B sweep in numpy_order mode, for a kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from quantization_amd.report import SweepPlan
torch.cuda.set_device(0)
wl = bench.ModelSweep("llama8b", torch.device("cuda", 0), 0, 1, "normal")
plan = SweepPlan(wl.weights, wl.passes, numpy_order=True)
for _ in range(2):
    plan.run()
torch.cuda.synchronize()
