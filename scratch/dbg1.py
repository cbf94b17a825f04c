import sys; sys.path.insert(0,'.')
import numpy as np, torch
import quantization_amd as qa
from quantization_amd import _lib
from oracle import oracle as orc
def bits(t): return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)
g = torch.Generator().manual_seed(0)
for scale_in, name in ((0.02,'normal'),(1e-6,'tiny')):
  for C in (512, 1024, 4096, 16384):
    w = (torch.randn(8, C, generator=g) * scale_in).half(); wd = w.cuda()
    for fmt, code, ofmt in (("int8", _lib.INT_SYM, orc.INT_SYM), ("int8_asym", _lib.INT_ASYM, orc.INT_ASYM), ("fp8_e4m3", _lib.LUT_FP, orc.LUT)):
      for gs in (None, 128):
        dt = fmt if code == _lib.LUT_FP else None
        res = qa.quantize(wd, code, 8, gs, dt)
        exp = orc.quantize(w.numpy(), orc.F16, ofmt, 8, gs, datatype=dt)
        d = bits(res.deq); s = bits(res.scale)
        q = bits(res.codes) if code == _lib.LUT_FP else res.codes.cpu().numpy()
        bd = np.argwhere(d != exp['deq']); bq = np.argwhere(q != exp['q']); bs = np.flatnonzero(s != exp['scale'])
        msg = f"{name} C={C} {fmt} gs={gs}: deq_bad={len(bd)} q_bad={len(bq)} scale_bad={len(bs)}"
        if res.zero_point is not None:
            z = bits(res.zero_point); bz = np.flatnonzero(z != exp['zp']); msg += f" zp_bad={len(bz)}"
            if len(bz): msg += f" zp got {z[bz[:3]]} exp {exp['zp'][bz[:3]]}"
        if len(bs): msg += f" scale got {s[bs[:3]]} exp {exp['scale'][bs[:3]]}"
        if len(bd):
            i = tuple(bd[0]); vi = w.numpy().view(np.uint16)[i]
            msg += f" first deq at {i}: v={vi:#06x} got={d[i]:#06x} exp={exp['deq'][i]:#06x} q got={q[i]} exp={exp['q'][i]}"
        print(msg)
