"""A few steps of rank 0's share of the Llama-3-8B sweep at one world size, for a kernel trace:
rocprofv3 --kernel-trace -f csv ... -- python3 scratch/share_trace.py WORLD"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
world = int(sys.argv[1])
wl = bench.ModelSweep("llama8b", torch.device("cuda", 0), 0, world, "normal")
for _ in range(6): wl.step()
torch.cuda.synchronize()
