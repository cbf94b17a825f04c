import sys; sys.path.insert(0,'.')
import torch, quantization_amd as qa
from quantization_amd import _lib
dev='cuda'
ws=[(torch.randn(4096,4096,device=dev)*0.02).half() for _ in range(16)]
wb=[w.bfloat16() for w in ws]
def timeit(f, n=3):
    f(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
variants = {
 'all':   dict(deq=True,codes=True,scale=True,abs_err=torch.float32,seg_sse=True),
 'noabs': dict(deq=True,codes=True,scale=True,abs_err=None,seg_sse=True),
 'deq':   dict(deq=True,codes=False,scale=False,abs_err=None,seg_sse=False),
 'stats': dict(deq=False,codes=False,scale=True,abs_err=None,seg_sse=True),
 'none':  dict(deq=False,codes=False,scale=False,abs_err=None,seg_sse=False),
}
for fmt,code,dt in (('sym',_lib.INT_SYM,None),('asym',_lib.INT_ASYM,None),('e4m3',_lib.LUT_FP,'fp8_e4m3')):
  for gs in (32,128,None):
    for vn,v in variants.items():
      res=[None]*16
      def run(W=ws):
        for i,w in enumerate(W): res[i]=qa.quantize(w,code,8,gs,dt,out=res[i],**v)
      t=timeit(run)/16*1e3
      line=f"{fmt} gs={gs} {vn}: {t:.1f} us"
      if vn in ('stats','none'):
        res=[None]*16
        t2=timeit(lambda: run(wb))/16*1e3
        line+=f"   bf16: {t2:.1f} us"
      print(line, flush=True)
