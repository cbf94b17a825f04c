"""Per-channel asymmetric rows through the one-launch-per-pass row kernels against the general quantize kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantization_amd import report, _lib
from quantization_amd.fused import quantize
g = torch.Generator().manual_seed(5)
for shape in [(6, 16384), (6, 8192), (8, 14336), (16, 4096), (5, 32768), (6, 16384)]:
    for dt in (torch.bfloat16, torch.float16):
        w = (torch.randn(*shape, generator=g) * 0.02).to(dt).cuda()
        for name, fmt in (("int8_asym", _lib.INT_ASYM), ("int8", _lib.INT_SYM)):
            plan = report.SweepPlan([w], [(name, 8, None)], fuse_groupings=False)
            plan.run(); torch.cuda.synchronize()
            got = plan.lists(0)[0].cpu()
            ref = quantize(w, fmt, 8, None, None, deq=False, codes=False, scale=False, seg_mse=True, totals=True).seg_mse.cpu()
            kinds = [la[0] for la in plan.plan[0]["launches"]]
            print(shape, dt, name, kinds, "max rel diff", float(((got - ref).abs() / ref).max()), got[:3].tolist(), ref[:3].tolist())
