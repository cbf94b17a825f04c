import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
    print('library', _lib.LIB_PATH)
from quantization_amd.fused import quantize_batch
torch.cuda.set_device(0)
g = torch.Generator(device='cuda').manual_seed(0)
stats = dict(deq=False, codes=False, scale=True, seg_sse=True, totals=True)
def t(w, fmt, dt, bits):
    res = None
    for _ in range(2):
        res = quantize_batch(w, fmt, bits, None, dt, out=res, **stats)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            res = quantize_batch(w, fmt, bits, None, dt, out=res, **stats)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    return best
for cols in (4096, 8192, 9216, 10240, 11008, 12288, 14336, 28672):
    rows = (1 << 28) // cols
    w = (torch.randn(1, rows, cols, generator=g, device='cuda') * 0.02).to(torch.bfloat16)
    line = f"cols={cols:6d} rows={rows:6d}:"
    for name, fmt, dt, bits in [("int8", _lib.INT_SYM, None, 8), ("asym", _lib.INT_ASYM, None, 8), ("e4m3", _lib.LUT_FP, "fp8_e4m3", 8)]:
        a = t(w, fmt, dt, bits)
        line += f"  {name} {a:7.1f} us {w.numel()*2/a/1e6:5.2f} TB/s"
    print(line, flush=True)
    del w
