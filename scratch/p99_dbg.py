import sys, torch, numpy as np
sys.path.insert(0, '.')
from quantization_amd import _lib, fused
from quantization_amd.fused import quantize_batch
torch.cuda.set_device(0)
g = torch.Generator(device='cuda').manual_seed(0)
w = (torch.randn(8, 4096, 4096, generator=g, device='cuda') * 0.02).half()
for fmt, gs in ((_lib.INT_SYM, None), (_lib.INT_ASYM, 128)):
    res = quantize_batch(w, fmt, 8, gs, None, deq=False, codes=False, abs_err=torch.float32, p99=99)
    torch.cuda.synchronize()
    ws = list(fused._workspaces.values())[0]
    off = (8 * (256 + 16384 * 16) + 255) // 256 * 256
    L = _lib.lib(); L.mq_percentile_workspace_bytes.restype = __import__("ctypes").c_size_t
    stride = L.mq_percentile_workspace_bytes()
    for i in range(8):
        h = ws[off + i * stride: off + i * stride + 64].cpu().numpy()
        print("   tensor", i, "br_lo, br_nb, ok =", h[40:52].view(np.uint32), "below", h[56:64].view(np.uint64)[0])
    hdr = ws[off:off + 64].cpu().numpy()
    br_lo, br_nb, ok, _ = hdr[40:56].view(np.uint32)
    below = hdr[56:64].view(np.uint64)[0]
    e = res.abs_err[0].cpu().numpy().ravel()
    bits = e.view(np.uint32) >> 16
    n = e.size
    k0 = int(np.float32(n - 1) * (np.float32(99) / np.float32(100)))
    print("fmt", fmt, "gs", gs, "br_lo", br_lo, "br_nb", br_nb, "ok", ok, "below", below,
          "true below", int((bits < br_lo).sum()), "inside", int(((bits >= br_lo) & (bits < br_lo + br_nb)).sum()), "k0", k0,
          "bin of p99", int(np.sort(bits)[k0]))
    # sample histogram check: samp area
    import ctypes
