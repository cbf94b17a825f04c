"""Time of the several-groupings launch (int8, per row + g = 32 / 64 / 128) per column count, about 1.9 GB of bfloat16
weights each; MQ_MULTI_NOSKIP / MQ_MULTI_NOROT select the forms of rows that do not fill the kernel's grid."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantization_amd import _lib
if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
    print("library", _lib.LIB_PATH)
from quantization_amd import report
from quantization_amd.fused import workspace
g = torch.Generator(device="cuda").manual_seed(1)
fmts = sys.argv[1].split(",") if len(sys.argv) > 1 else ["int8"]
for cols in (4096, 8192, 5632, 11008, 14336, 28672):
    n = max(1, int(1.9e9 / (4096 * cols * 2)))
    ws = [(torch.randn(4096, cols, generator=g, device="cuda") * 0.02).bfloat16() for _ in range(n)]
    for fmt in fmts:
        plan = report.SweepPlan(ws, [(fmt, 8, gs) for gs in (32, 64, 128, None)])
        la = [l for ps in plan.plan for l in ps["launches"] if l[0] == "multisweep"]
        L = plan.L
        ws_, st = workspace(ws[0].device, plan.ws_bytes)
        cfg = plan.plan[0]["cfg"]
        def run():
            for l in la:
                _lib.check(L.mq_sweep_multi_run(l[1].data_ptr(), l[2], l[3], l[4], l[5], ctypes.byref(cfg), l[6],
                                                ws_.data_ptr(), ws_.numel(), st.cuda_stream))
        run(); torch.cuda.synchronize()
        us = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            us = min(us, e0.elapsed_time(e1) / 20 * 1000)
        gb = sum(w.numel() for w in ws) * 2 / 1e9
        print(f"{fmt:10s} cols {cols:6d}  {us:8.1f} us for {gb:.2f} GB  {gb / us * 1e3:6.2f} TB/s of weights, {len(la)} launches", flush=True)
    del ws, plan
