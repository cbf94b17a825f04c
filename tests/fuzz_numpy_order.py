"""Randomised differential run of the numpy-order modes against numpy itself (on the CPU oracle's dequantized values):
per-segment list entries (seg_mse="numpy": groups of 32 / 64 / 128 in the bandwidth kernels, every other length in
the wave-per-segment kernel), the list's float64 mean (report.numpy_list_means) and the tensor mean
(report.tensor_mse_numpy) -- bits, a NaN matching a NaN.  run(seconds, seed); as a script: argv = seconds, seed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import oracle as orc
from quantization_amd import _lib, report
from quantization_amd.fused import make_cfg, quantize


def run(budget, seed):
    orc.lib()
    rng = np.random.default_rng(seed)
    FORMATS = [("int8", 8), ("int8_asym", 8), ("int4", 4), ("int4_asym", 4), ("fp8_e4m3", 8), ("fp8_e3m4", 8), ("fp6_e3m2", 6), ("fp4", 4)]
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        name, bits = FORMATS[rng.integers(len(FORMATS))]
        gs = [None, None, None, 32, 64, 128, 256, 512, 1024, 96, 40, 8][rng.integers(12)]
        unit = gs or 8
        if gs is None:
            cols = int(rng.choice([8 * rng.integers(1, 600), 1024, 2048, 4096, 5632, 8192, 14336, 11008, 8 * rng.integers(600, 4096)]))
        else:
            cols = unit * int(rng.integers(1, max(2, 6144 // unit)))
        rows = int(rng.integers(1, 40))
        dt = ["f16", "bf16"][rng.integers(2)]
        scale = float(rng.choice([0.02, 1.0, 30.0, 1e-3, 1e-5]))
        w = torch.from_numpy(rng.standard_normal((rows, cols)).astype(np.float32)) * scale
        kind = rng.integers(8)
        if kind == 0: w[rng.integers(rows)] = 0
        if kind == 1: w[rng.integers(rows)] = w[rng.integers(rows)].abs() + scale * 0.1
        if kind == 2: w[rng.integers(rows), rng.integers(cols)] = float("inf")
        if kind == 3: w[rng.integers(rows), rng.integers(cols)] = float("nan")
        if kind == 4: w[:, rng.integers(cols)] = scale * 40
        if kind == 5: w = torch.from_numpy(rng.standard_t(3, (rows, cols)).astype(np.float32)) * scale
        wt = w.half() if dt == "f16" else w.bfloat16()
        if name.startswith("int"):
            asym = "asym" in name
            code, table, oargs = (_lib.INT_ASYM if asym else _lib.INT_SYM), None, dict(fmt=orc.INT_ASYM if asym else orc.INT_SYM, bits=bits)
        else:
            code, table, oargs = _lib.LUT_FP, name, dict(fmt=orc.LUT, bits=bits, datatype=name)
        what = f"[{n}] {name}/{dt}/{rows}x{cols}/g={gs}/scale={scale}/kind={kind}"
        try:
            res = quantize(wt.cuda(), code, bits, gs, table, deq=False, codes=False, scale=True, seg_mse="numpy", totals=True)
            rep = report.numpy_list_means([res.seg_mse.reshape(-1)])
            mt = torch.zeros((), dtype=torch.float32, device="cuda")
            report.tensor_mse_numpy(wt.cuda(), make_cfg(code, bits, gs, table), res.scale, res.zero_point, mt)
        except Exception as e:
            print("EXC", what, e, flush=True); raise
        L = cols if gs is None else gs
        raw = wt.contiguous().view(torch.int16).numpy().view(np.uint16)
        exp = orc.quantize(raw, orc.F16 if dt == "f16" else orc.BF16, group_size=gs, **oargs)
        a = wt.half().float().numpy()
        b = exp["deq"].view(np.float16).astype(np.float32).reshape(rows, cols)
        with np.errstate(invalid="ignore", over="ignore"):
            vec = np.array([float(np.mean((a[r, g:g + L] - b[r, g:g + L]) ** 2)) for r in range(rows) for g in range(0, cols, L)])
            want_rep, want_mt = float(vec.mean()), np.float32(np.mean((a - b) ** 2))
        got = res.seg_mse.cpu().numpy().reshape(-1)
        v32 = vec.astype(np.float32)
        same = (got.view(np.uint32) == v32.view(np.uint32)) | (np.isnan(got) & np.isnan(v32))
        assert same.all(), f"{what}: {int((~same).sum())} list entries differ"
        g_rep, g_mt = float(rep[0].item()), mt.cpu().numpy()
        assert g_rep == want_rep or (np.isnan(g_rep) and np.isnan(want_rep)), f"{what}: mse_rep {g_rep} vs {want_rep}"
        assert g_mt.tobytes() == want_mt.tobytes() or (np.isnan(g_mt) and np.isnan(want_mt)), f"{what}: mse_tensor {g_mt} vs {want_mt}"
        n += 1
    return n


if __name__ == "__main__":
    t0 = time.time()
    n = run(float(sys.argv[1]), int(sys.argv[2]))
    print("ok:", n, "random cases in", round(time.time() - t0, 1), "s, seed", sys.argv[2], flush=True)
