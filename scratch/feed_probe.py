"""Timing of one per-group pass over Llama-3-8B-shaped weights in three forms: lists written, no lists (totals
only: the floor for anything that avoids the stores), lists fed.  Quantize launches only (library event hooks)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantization_amd import _lib, report, shapes

L = _lib.lib()
dev = torch.device("cuda", 0)
layers = shapes.linear_shapes("llama-3-8b")
ws = [shapes.synthetic_weight(i, s, torch.bfloat16, dev) for i, (_, s) in enumerate(layers)]

def run(plan, steps=6):
    _lib.check(L.mq_profile_reserve(64))
    for _ in range(2):
        plan.run(overlap=False)
    torch.cuda.synchronize()
    _lib.check(L.mq_profile_begin())
    for _ in range(steps):
        plan.run(overlap=False)
    torch.cuda.synchronize()
    ms, n = ctypes.c_double(0), ctypes.c_int64(0)
    _lib.check(L.mq_profile_end(ctypes.byref(ms), ctypes.byref(n)))
    return ms.value / n.value * 1e3

for gs in (32, 64, 128):
    out = []
    for mode in ("lists", "nolist", "feed"):
        os.environ["MQ_SWEEP_NOLIST"] = "1" if mode == "nolist" else "0"
        plan = report.SweepPlan(ws, [("int8", 8, gs)], feed_lists=(mode == "feed"))
        out.append((mode, round(run(plan), 1)))
        del plan
    print("g =", gs, out, flush=True)
