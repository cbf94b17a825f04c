"""Per-pass kernel time of the llama8b sweep (HIP events around each pass, selections on the same stream skipped):
argv: world size.  Environment: MQ_SWEEP_SINGLE, MQ_SWEEP_WG_KB."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from quantization_amd import _lib
from quantization_amd.fused import workspace
torch.cuda.set_device(0)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1
wl = bench.ModelSweep("llama8b", torch.device("cuda", 0), 0, W, "normal")
for _ in range(2):
    wl.step()
torch.cuda.synchronize()
plan = wl.plan
L, dev = plan.L, plan.dev
st = torch.cuda.current_stream(dev)
ws, _ = workspace(dev, plan.ws_bytes)
out = []
for k, ps in enumerate(plan.plan):
    cfg = ps["cfg"]
    def go():
        for la in ps["launches"]:
            if la[0] == "sweep":
                _, tab, n, seg_len, blocks, dt, flag = la
                _lib.check(L.mq_sweep_run(tab.data_ptr(), n, seg_len, blocks, dt, ctypes.byref(cfg), flag, ws.data_ptr(), ws.numel(), st.cuda_stream))
            else:
                _, items, B, dt, K, C = la
                _lib.check(L.mq_quantize_batch(items, B, dt, K, C, ctypes.byref(cfg), ws.data_ptr(), ws.numel(), st.cuda_stream))
    go(); go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5):
        go()
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gb = wl.local_elems * 2 / 1e9
    out.append(f"g={wl.passes[k][2]}: {len(ps['launches'])} launches {ms:.3f} ms {gb / ms:.0f} GB/s")
print(f"single={os.environ.get('MQ_SWEEP_SINGLE', '1')} wg_kb={os.environ.get('MQ_SWEEP_WG_KB', '64')} world {W}: " + "; ".join(out), flush=True)
