import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from quantization_amd import _lib
from quantization_amd.fused import quantize
from oracle import oracle
from test_gpu_stats_kernels import _inputs
def pw(a):
    n = len(a)
    if n <= 128:
        r = [np.float32(a[i]) for i in range(8)]
        for i in range(8, n, 8):
            for j in range(8): r[j] = np.float32(r[j] + a[i + j])
        return np.float32(np.float32(np.float32(r[0]+r[1])+np.float32(r[2]+r[3]))+np.float32(np.float32(r[4]+r[5])+np.float32(r[6]+r[7])))
    n2 = n // 2; n2 -= n2 % 8
    return np.float32(pw(a[:n2]) + pw(a[n2:]))
for rows, cols, seed in ((5, 14336, 901), (4, 11008, 902)):
    w = _inputs(rows, cols, seed).half()
    res = quantize(w.cuda(), _lib.INT_SYM, 8, None, None, deq=False, codes=False, scale=True, seg_mse="numpy", totals=True)
    exp = oracle.quantize(w.view(torch.int16).numpy().view(np.uint16), oracle.F16, group_size=None, fmt=oracle.INT_SYM, bits=8)
    a = w.float().numpy(); b = exp["deq"].view(np.float16).astype(np.float32).reshape(rows, cols)
    got = res.seg_mse.cpu().numpy()
    for r in range(rows):
        d = (a[r] - b[r]); sq = d * d
        s_np = np.add.reduce(sq); s_pw = pw(sq)
        m_np = np.mean(sq)
        print(rows, cols, r, "got", got[r].view(np.uint32), "np.mean", np.float32(m_np).view(np.uint32), "pw/n f32", np.float32(s_pw / np.float32(cols)).view(np.uint32),
              "pw sum", s_pw.view(np.uint32), "np sum", s_np.view(np.uint32), "f64 path", np.float32(np.float64(s_np) / cols).view(np.uint32), "got*n", np.float32(got[r] * np.float32(cols)))
