"""Host-side timeline of SweepPlan.run() for one rank's share: how long each enqueue call takes on the host."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from quantization_amd import _lib, report
from quantization_amd.fused import workspace
torch.cuda.set_device(0)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = bench.ModelSweep("llama8b", torch.device("cuda", 0), 0, W, "normal")
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
plan = wl.plan
dev, L = plan.dev, plan.L
nl = len(plan.order)
main = torch.cuda.current_stream(dev)
side = report._side_streams[dev.index]
log = []
T0 = time.perf_counter()
def mark(what):
    log.append((what, (time.perf_counter() - T0) * 1e6))
side.wait_stream(main)
ws, _ = workspace(dev, plan.ws_bytes)
mark("start")
for k, ps in enumerate(plan.plan):
    cfg = ps["cfg"]
    for items, B, dt, K, C, _ in ps["launches"]:
        _lib.check(L.mq_quantize_batch(items, B, dt, K, C, ctypes.byref(cfg), ws.data_ptr(), ws.numel(), main.cuda_stream))
    mark(f"pass {k}: {len(ps['launches'])} launches enqueued")
    ready = torch.cuda.Event()
    ready.record(main)
    mark("  event recorded")
    with torch.cuda.stream(side):
        side.wait_event(ready)
        mark("  side waits")
        sws, _ = workspace(dev, plan.sel_ws_bytes)
        _lib.check(L.mq_percentile_multi(ps["ptrs"], ps["ns"], nl, _lib.F32_AS_F64, plan.pct, ps["p99"].data_ptr(), None,
                                         sws.data_ptr(), sws.numel(), side.cuda_stream))
        mark("  mq_percentile_multi returned")
        tot = ps["totals"]
        mse = tot[:, 0] / plan.numel
        p99 = torch.where(torch.isnan(mse), mse, ps["p99"])
        rec = torch.stack([plan.numel, tot[:, 0], tot[:, 1], mse, p99, torch.zeros_like(mse)], dim=1)
        plan.records[k].index_copy_(0, plan.perm, rec)
        mark("  torch ops enqueued")
main.wait_stream(side)
mark("joined")
torch.cuda.synchronize()
mark("GPU done")
for what, t in log:
    print(f"{t:9.1f} us  {what}")
