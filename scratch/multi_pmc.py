"""A few several-groupings launches (int8, 32 x 4096 x 4096 bfloat16) per mask, for a counter run:
rocprofv3 --pmc ... -- python3 scratch/multi_pmc.py [mask ...]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantization_amd import report, _lib
from quantization_amd.fused import workspace
masks = [int(a) for a in sys.argv[1:]] or [15, 1]
g = torch.Generator(device="cuda").manual_seed(1)
ws = [(torch.randn(4096, 4096, generator=g, device="cuda") * 0.02).bfloat16() for _ in range(32)]
plan = report.SweepPlan(ws, [("int8", 8, None), ("int8", 8, 32), ("int8", 8, 64), ("int8", 8, 128)])
la = [l for ps in plan.plan for l in ps["launches"] if l[0] == "multisweep"]
L = plan.L
ws_, st = workspace(ws[0].device, plan.ws_bytes)
cfg = plan.plan[0]["cfg"]
for mask in masks:
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for l in la:
            _lib.check(L.mq_sweep_multi_run(l[1].data_ptr(), l[2], l[3], l[4], l[5], ctypes.byref(cfg), mask,
                                            ws_.data_ptr(), ws_.numel(), st.cuda_stream))
        e1.record(); torch.cuda.synchronize()
        print(f"mask {mask:2d}: {e0.elapsed_time(e1) * 1000:8.1f} us, {len(la)} launches, {sum(w.numel() for w in ws) * 2 / 1e9:.2f} GB", flush=True)
