"""Timing of the imported reference itself (torch CPU + numpy) on the c2 unit of work, in the BUILD container
(/root/reference is not present on the GPU box): one 4096x4096 fp16 matrix, asymmetric int8, g = 32 / 64 / 128,
with the analysis scripts' numpy error arithmetic.  Run by hand: python scratch/ref_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quant_utils import quant_weight as qw
from oracle import oracle as orc   # the numpy restatement of the analysis scripts

torch.manual_seed(0)
w = (torch.randn(4096, 4096) * 0.02).half()
qw._int_quant_meta(w[:64], 8, True, 128)
t0 = time.perf_counter(); passes = 0
while time.perf_counter() - t0 < 20:
    for gs in (32, 64, 128):
        d, q, s, zp = qw._int_quant_meta(w, wq_bits=8, asym=True, group_size=gs)
        a, b = w.numpy(), d.numpy()
        a32, b32 = orc.proto_to_array(a), orc.proto_to_array(b)
        orc.compute_mse(a32, b32)
        orc.elementwise_f32(a, b)
        dd = (a32 - b32).reshape(-1, gs)
        vec = np.mean(dd * dd, axis=1, dtype=np.float32).astype(np.float64)
        vec.mean(), np.percentile(vec, 99)
        passes += 1
dt = time.perf_counter() - t0
print(f"{passes} passes in {dt:.1f} s: {passes * w.numel() * 2 / dt / 1e9:.4f} GB/s of weights, torch threads {torch.get_num_threads()}")
