import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from quantization_amd import _lib
from quantization_amd.fused import quantize
from oracle import oracle
import test_gpu_stats_kernels as T
rows, cols, gs = 37, 1024, 32
w = T._inputs(rows, cols, 500)
wt = w.half()
res = quantize(wt.cuda(), _lib.INT_SYM, 8, gs, None, deq=False, codes=False, scale=False, seg_mse="numpy", totals=True)
raw = wt.view(torch.int16).numpy().view(np.uint16)
exp = oracle.quantize(raw, oracle.F16, group_size=gs, fmt=oracle.INT_SYM, bits=8)
a = wt.float().numpy(); b = exp["deq"].view(np.float16).astype(np.float32).reshape(rows, cols)
with np.errstate(all="ignore"):
    want = np.array([np.mean((a[r, g:g + gs] - b[r, g:g + gs]) ** 2) for r in range(rows) for g in range(0, cols, gs)], dtype=np.float32)
got = res.seg_mse.cpu().numpy()
bad = np.where((got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want)))[0]
print(len(bad), "bad; rows:", sorted(set(bad // (cols // gs))))
for i in bad[:6]:
    print(i, i // (cols // gs), got[i], want[i], got[i].view(np.uint32) - want[i].view(np.uint32))
