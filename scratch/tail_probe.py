"""Where a rank's step ends: time from the step's start to the end of its last statistics launch (main stream) and to
the end of the step, for rank 0's share of the Llama-3-8B sweep at several world sizes (one GPU, no collectives)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from quantization_amd import report
dev = torch.device("cuda", 0)
NP = "--numpy" in sys.argv
worlds = [int(a) for a in sys.argv[1:] if a != "--numpy"] or [1, 8]
for world in worlds:
    wl = bench.ModelSweep("llama8b", dev, 0, world, "normal", numpy_order=NP)
    marks = []
    orig = wl.plan.launch
    def launch(overlap=True, _o=orig):
        _o(overlap)
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e)
    wl.plan.launch = launch
    for _ in range(3): wl.step()
    torch.cuda.synchronize()
    marks.clear()
    n = 20
    ev = []
    t0 = time.perf_counter()
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); a.record()
        wl.step()
        b = torch.cuda.Event(enable_timing=True); b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    main = sum(a.elapsed_time(m) for (a, b), m in zip(ev, marks)) / n
    full = sum(a.elapsed_time(b) for a, b in ev) / n
    print(f"world {world}: {len(wl.mine)} layers; wall {wall * 1e3:.3f} ms/step; GPU start -> last statistics launch done "
          f"{main:.3f} ms, -> step end {full:.3f} ms (tail {full - main:.3f})", flush=True)
    del wl
    torch.cuda.empty_cache()
