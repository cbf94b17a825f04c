"""A/B of the step's tail in one process: the model-wide percentile after the sweep's streams are joined (A) or beside
the last pass's selections (B: SweepPlan.launch / finish, what bench.py does)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from quantization_amd.distributed import gather_records, joint_percentile
from quantization_amd.report import RECORD_FIELDS
torch.cuda.set_device(0)
for W in (1, 8):
    wl = bench.ModelSweep("llama8b", torch.device("cuda", 0), 0, W, "normal")
    wl.step()
    def step_a():
        local = wl.plan.run().reshape(-1, len(RECORD_FIELDS))
        p = joint_percentile(wl.plan.lists(wl.joint_pass), 99.0, dtype=torch.float32, total=wl.total_vectors, wide=True)
        return gather_records(wl.index, local, wl.n_layers * len(wl.passes), len(RECORD_FIELDS), wl.device), p
    res = {"A": [], "B": []}
    for rep in range(4):
        for name, fn in (("A", step_a), ("B", wl.step)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"world {W}: A {min(res['A']):.3f} ms (runs {' '.join(f'{x:.3f}' for x in res['A'])}); "
          f"B {min(res['B']):.3f} ms (runs {' '.join(f'{x:.3f}' for x in res['B'])})", flush=True)
    del wl
    torch.cuda.empty_cache()
