import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import oracle as orc
from quantization_amd import quant_weight as qw
orc.lib()
rng = np.random.default_rng(5)
found = 0
for it in range(200):
    rows, cols = 17, 1816
    w = (torch.from_numpy(rng.standard_t(3, (rows, cols)).astype(np.float32)) * 1e-3).half()
    d = qw.quant_datatype(w.cuda(), wq_bits=8, datatype="fp8_e4m3", group_size=-1)
    exp = orc.quantize(w.numpy(), orc.F16, fmt=orc.LUT, bits=8, datatype="fp8_e4m3", group_size=-1, lut_f32_path=True, want=("deq", "q", "scale"))
    got = d.cpu().view(torch.int16).numpy().view(np.uint16).reshape(-1)
    want = exp["deq"].reshape(-1)
    bad = np.nonzero(got != want)[0]
    if bad.size:
        w32 = w.float().numpy().reshape(-1)
        rmax = np.float32(np.abs(w32).max())
        s = np.float32(rmax / np.float32(480.0))
        for i in bad[:4]:
            x = np.float32(w32[i] / s)
            print(f"it {it} elem {i}: w {w32[i]!r} bits {w.view(torch.int16).numpy().view(np.uint16).reshape(-1)[i]:#06x} s {s!r} x {x!r} got {got[i]:#06x} want {want[i]:#06x}  want_f {want[i:i+1].view(np.float16)[0]!r}")
        found += 1
        if found >= 3: break
print("done", found)
