import sys, os
sys.path.insert(0,'.')
import torch, quantization_amd as qa
from quantization_amd import _lib
r,c,fmt,gs,vn = int(sys.argv[1]),int(sys.argv[2]),sys.argv[3],(None if sys.argv[4]=='None' else int(sys.argv[4])),sys.argv[5]
dt = sys.argv[6] if len(sys.argv)>6 else 'f16'
VARS={'all':dict(deq=True,codes=True,scale=True,abs_err=torch.float32,seg_sse=True),'api':dict(deq=True,codes=True,scale=True),'stats':dict(deq=False,codes=False,scale=True,abs_err=None,seg_sse=True),'none':dict(deq=False,codes=False,scale=False,abs_err=None,seg_sse=False)}
codes={'sym':(_lib.INT_SYM,None),'asym':(_lib.INT_ASYM,None),'e4m3':(_lib.LUT_FP,'fp8_e4m3')}
w=(torch.randn(r,c,device='cuda')*0.02).to(torch.float16 if dt=='f16' else torch.bfloat16)
for _ in range(3): qa.quantize(w,codes[fmt][0],8,gs,codes[fmt][1],**VARS[vn])
torch.cuda.synchronize()
