"""Randomised differential run of the statistics kernels (batched, every grouping, random shapes and contents)
against the CPU oracle: scales / zero-points bit for bit, per-segment float32 MSE entries and totals to 1e-6.
run(seconds, seed); as a script: argv = seconds, seed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import golden_util as gu
from oracle import oracle as orc
from quantization_amd import _lib
from quantization_amd.fused import quantize_batch


def run(budget, seed):
    orc.lib()
    rng = np.random.default_rng(seed)
    FORMATS = [("int8", 8), ("int8_asym", 8), ("int4", 4), ("int4_asym", 4), ("fp8_e4m3", 8), ("fp8_e3m4", 8), ("fp8_e2m5", 8), ("fp6_e3m2", 6), ("fp4", 4)]
    def close(got, want, what):
        got, want = np.asarray(got, np.float64).reshape(-1), np.asarray(want, np.float64).reshape(-1)
        fin = np.isfinite(want)
        assert np.array_equal(np.isnan(got), np.isnan(want)), what + ": NaN pattern"
        assert np.array_equal(np.isinf(got), np.isinf(want)), what + ": inf pattern"
        err = np.abs(got[fin] - want[fin])
        bad = err > 1e-6 * np.abs(want[fin]) + 1e-300
        assert not bad.any(), f"{what}: {bad.sum()} beyond 1e-6, worst {np.max(err / np.maximum(np.abs(want[fin]), 1e-300))}"
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        name, bits = FORMATS[rng.integers(len(FORMATS))]
        gs = [None, None, 32, 64, 128, 256, 512, 96, 40][rng.integers(9)]
        unit = gs or 8
        if gs is None:
            cols = int(rng.choice([8 * rng.integers(1, 600), 1024, 2048, 4096, 5632, 8192, 14336, 11008, 8 * rng.integers(600, 4200)]))
        else:
            cols = unit * int(rng.integers(1, max(2, 4096 // unit)))
        rows = int(rng.integers(1, 70))
        B = int(rng.choice([1, 1, 2, 3, 5]))
        dt = ["f16", "bf16"][rng.integers(2)]
        scale = float(rng.choice([0.02, 1.0, 30.0, 1e-3, 1e-5]))
        w = torch.from_numpy(rng.standard_normal((B, rows, cols)).astype(np.float32)) * scale
        kind = rng.integers(8)
        if kind == 0: w[:, rng.integers(rows)] = 0
        if kind == 1: w[:, rng.integers(rows)] = w[:, rng.integers(rows)].abs() + scale * 0.1
        if kind == 2: w[rng.integers(B), rng.integers(rows), rng.integers(cols)] = float("inf")
        if kind == 3: w[rng.integers(B), rng.integers(rows), rng.integers(cols)] = float("nan")
        if kind == 4: w[:, :, rng.integers(cols)] = scale * 40
        if kind == 5: w = torch.from_numpy(rng.standard_t(3, (B, rows, cols)).astype(np.float32)) * scale
        wt = w.half() if dt == "f16" else w.bfloat16()
        if name.startswith("int"):
            asym = "asym" in name
            code, table, oargs = (_lib.INT_ASYM if asym else _lib.INT_SYM), None, dict(fmt=orc.INT_ASYM if asym else orc.INT_SYM, bits=bits)
        else:
            code, table, oargs = _lib.LUT_FP, name, dict(fmt=orc.LUT, bits=bits, datatype=name)
        what = f"[{n}] {name}/{dt}/B{B}x{rows}x{cols}/g={gs}/scale={scale}/kind={kind}"
        try:
            res = quantize_batch(wt.cuda(), code, bits, gs, table, deq=False, codes=False, scale=True, seg_mse=True, totals=True)
        except Exception as e:
            print("EXC", what, e, flush=True); raise
        L = cols if gs is None else gs
        for b in range(B):
            raw = wt[b].contiguous().view(torch.int16).numpy().view(np.uint16)
            exp = orc.quantize(raw, orc.F16 if dt == "f16" else orc.BF16, group_size=gs, **oargs)
            assert gu.same_f16_bits(res.scale[b].view(torch.int16).cpu().numpy().view(np.uint16), exp["scale"]), what + " scale"
            if "asym" in name:
                assert gu.same_f16_bits(res.zero_point[b].view(torch.int16).cpu().numpy().view(np.uint16), exp["zp"]), what + " zp"
            v = wt[b].half().float().numpy()
            with np.errstate(invalid="ignore", over="ignore"):
                e = v - exp["deq"].view(np.float16).astype(np.float32).reshape(rows, cols)
                sq = (e * e).astype(np.float32).astype(np.float64)
                close(res.seg_mse[b].cpu().numpy(), sq.reshape(-1, L).sum(axis=1) / L, what + " seg_mse")
                tot = res.totals[b].cpu().numpy()
                close(tot[0], sq.sum(), what + " sse")
                close(tot[1], np.abs(e).astype(np.float64).sum(), what + " sum|err|")
        n += 1
    return n


if __name__ == "__main__":
    t0 = time.time()
    n = run(float(sys.argv[1]), int(sys.argv[2]))
    print("ok:", n, "random cases in", round(time.time() - t0, 1), "s, seed", sys.argv[2], flush=True)
