"""Does a collective right after SweepPlan.run() (two streams, pending work) complete?  One rank, backend from argv."""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(50, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
backend, overlap = sys.argv[1], sys.argv[2] == "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
if backend == "nccl":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
else:
    dist.init_process_group(backend, rank=0, world_size=1)
import bench
from quantization_amd.report import RECORD_FIELDS, SweepPlan
dev = torch.device("cuda", 0)
wl = bench.ModelSweep("llama8b", dev, 0, 1, "normal")
plan = SweepPlan(wl.weights, wl.passes)
for it in range(4):
    t0 = time.perf_counter()
    local = plan.run(overlap=overlap).reshape(-1, len(RECORD_FIELDS))
    table = torch.zeros(wl.n_layers * len(wl.passes), len(RECORD_FIELDS), dtype=torch.float64, device=dev)
    table[wl.index] = local
    dist.all_reduce(table)
    h = torch.zeros(32768, dtype=torch.int32, device=dev)
    dist.all_reduce(h)
    torch.cuda.synchronize()
    print(backend, "overlap", overlap, "step", it, f"{(time.perf_counter() - t0) * 1e3:.1f} ms", float(table.sum()), flush=True)
dist.destroy_process_group()
print("ok", flush=True)
