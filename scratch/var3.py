import sys, glob, csv
sys.path.insert(0,'.')
REPS=3
SIZES=[(1024,4096),(4096,4096),(8192,8192),(16384,16384)]
VARS={'all':dict(deq=True,codes=True,scale=True,abs_err='f32',seg_sse=True),'stats':dict(deq=False,codes=False,scale=True,abs_err=None,seg_sse=True),'none':dict(deq=False,codes=False,scale=False,abs_err=None,seg_sse=False)}
def configs():
    return [(r,c,fmt,gs,vn) for (r,c) in SIZES for fmt in ('sym','asym') for gs in (128,None) for vn in VARS]
if sys.argv[1]=='run':
    import torch, quantization_amd as qa
    from quantization_amd import _lib
    codes={'sym':(_lib.INT_SYM,None),'asym':(_lib.INT_ASYM,None)}
    for (r,c) in SIZES:
        w=(torch.randn(r,c,device='cuda')*0.02).half()
        for fmt in ('sym','asym'):
            for gs in (128,None):
                for vn,v in VARS.items():
                    v=dict(v); v['abs_err']=torch.float32 if v['abs_err'] else None
                    for _ in range(REPS): qa.quantize(w,codes[fmt][0],8,gs,None,**v)
                    torch.cuda.synchronize()
        del w
else:
    f=glob.glob(sys.argv[2]+'/**/*kernel_trace.csv', recursive=True)[0]
    rows=[r for r in csv.DictReader(open(f)) if 'k_quant' in r['Kernel_Name']]
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    cfgs=configs(); assert len(rows)==len(cfgs)*REPS,(len(rows),len(cfgs)*REPS)
    for i,(r,c,fmt,gs,vn) in enumerate(cfgs):
        ds=[(int(x['End_Timestamp'])-int(x['Start_Timestamp']))/1e3 for x in rows[i*REPS:(i+1)*REPS]]
        n=r*c; by={'all':9.2,'stats':2.0,'none':2.0}[vn]
        print(f"{r}x{c} {fmt:4s} gs={str(gs):4s} {vn:5s} {min(ds):8.1f} us  {n*by/min(ds)/1e3:7.0f} GB/s  grid={rows[i*REPS]['Grid_Size_X'] if 'Grid_Size_X' in rows[i*REPS] else rows[i*REPS].get('Grid_Size','?')} {rows[i*REPS]['Kernel_Name'].split('(')[0][5:]}")
