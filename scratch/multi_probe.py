"""Time of the several-groupings launch per grouping mask (16 x 4096 x 4096 bfloat16, int8 and fp8_e4m3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantization_amd import report
g = torch.Generator(device="cuda").manual_seed(1)
ws = [(torch.randn(4096, 4096, generator=g, device="cuda") * 0.02).bfloat16() for _ in range(32)]
GS = {1: None, 2: 32, 4: 64, 8: 128}
for fmt, bits in (("int8", 8), ("int8_asym", 8), ("fp8_e4m3", 8)):
    for mask in (1, 2, 4, 8, 3, 5, 9, 6, 15):
        passes = [(fmt, bits, GS[b]) for b in (1, 2, 4, 8) if mask & b]
        if len(passes) < 2:
            passes.append((fmt, bits, 256))      # (a family needs two members; the g = 256 pass is not part of it)
            continue
        plan = report.SweepPlan(ws, passes)
        la = [l for ps in plan.plan for l in ps["launches"] if l[0] == "multisweep"]
        import ctypes
        from quantization_amd import _lib
        from quantization_amd.fused import workspace
        L = plan.L
        ws_, st = workspace(ws[0].device, plan.ws_bytes)
        cfg = plan.plan[0]["cfg"]
        def run():
            for l in la:
                _lib.check(L.mq_sweep_multi_run(l[1].data_ptr(), l[2], l[3], l[4], l[5], ctypes.byref(cfg), l[6], ws_.data_ptr(), ws_.numel(), st.cuda_stream))
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        print(f"{fmt:10s} mask {mask:2d}  {e0.elapsed_time(e1) / 5 * 1000:8.1f} us for {sum(w.numel() for w in ws) * 2 / 1e9:.2f} GB", flush=True)
