# Kernel-time table from a rocprofv3 kernel trace: run as
#   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/var2 -- python3 scratch/var2.py run
#   python3 scratch/var2.py report gpurun_out/var2
import sys, glob, csv, json
sys.path.insert(0,'.')
REPS=4
def configs():
    variants = {
     'all':   dict(deq=True,codes=True,scale=True,abs_err='f32',seg_sse=True),
     'api':   dict(deq=True,codes=True,scale=True,abs_err=None,seg_sse=False),
     'stats': dict(deq=False,codes=False,scale=True,abs_err=None,seg_sse=True),
     'none':  dict(deq=False,codes=False,scale=False,abs_err=None,seg_sse=False),
    }
    out=[]
    for fmt in ('sym','asym','e4m3'):
        for gs in (32,128,None):
            for dt in ('f16','bf16'):
                for vn,v in variants.items():
                    if dt=='bf16' and vn in ('all','api'): continue
                    out.append((fmt,gs,dt,vn,v))
    return out
if sys.argv[1]=='run':
    import torch, quantization_amd as qa
    from quantization_amd import _lib
    ws=[(torch.randn(4096,4096,device='cuda')*0.02).half() for _ in range(REPS)]
    wb=[w.bfloat16() for w in ws]
    codes={'sym':(_lib.INT_SYM,None),'asym':(_lib.INT_ASYM,None),'e4m3':(_lib.LUT_FP,'fp8_e4m3')}
    torch.cuda.synchronize()
    for fmt,gs,dt,vn,v in configs():
        v=dict(v); v['abs_err']=torch.float32 if v['abs_err'] else None
        code,d=codes[fmt]
        for w in (ws if dt=='f16' else wb):
            qa.quantize(w,code,8,gs,d,**v)
        torch.cuda.synchronize()
else:
    f=glob.glob(sys.argv[2]+'/**/*kernel_trace.csv', recursive=True)[0]
    rows=[r for r in csv.DictReader(open(f)) if 'k_quant' in r['Kernel_Name']]
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    cfgs=configs()
    assert len(rows)==len(cfgs)*REPS, (len(rows), len(cfgs)*REPS)
    for i,(fmt,gs,dt,vn,v) in enumerate(cfgs):
        ds=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows[i*REPS:(i+1)*REPS]]
        name=rows[i*REPS]['Kernel_Name'].split('(')[0].replace('void ','')
        print(f"{fmt:5s} gs={str(gs):5s} {dt:4s} {vn:6s} {min(ds):7.1f} us (median {sorted(ds)[len(ds)//2]:7.1f})  vgpr={rows[i*REPS].get('VGPR_Count','?')} {name}")
