"""A few c2 steps for a kernel trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
wl = bench.Workload("c2", torch.device("cuda", 0), 0)
for _ in range(6): wl.step()
torch.cuda.synchronize()
