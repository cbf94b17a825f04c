import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantization_amd import error_metrics as em
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(8192, 8192, generator=g, device="cuda").half()
b = (a.float() + 0.01 * torch.randn(8192, 8192, generator=g, device="cuda")).half()
for _ in range(5): em.error_stats(a, b, cols=True)
torch.cuda.synchronize()
