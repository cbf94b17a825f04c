"""Randomised differential run of the exact percentile selection against numpy: lists of random lengths and
contents (ties, zeros, infinities, NaN, tiny and huge ranges), float32 / float64 / float32-keys-with-float64-result,
single, multi and row forms, and the selection fed from the quantize pass.  run(seconds, seed); as a script: argv = seconds, seed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from quantization_amd import _lib
from quantization_amd.error_metrics import percentile, percentile_multi, percentile_rows
from quantization_amd.fused import quantize_batch


def run(budget, seed):
    rng = np.random.default_rng(seed)
    def draw(n, dtype):
        kind = rng.integers(9)
        if kind == 0: x = np.abs(rng.standard_normal(n))
        elif kind == 1: x = np.abs(rng.standard_t(2, n)) * 1e-3
        elif kind == 2: x = rng.integers(0, 5, n).astype(np.float64)             # heavy ties
        elif kind == 3: x = np.zeros(n)
        elif kind == 4: x = np.abs(rng.standard_normal(n)) * 10.0 ** rng.integers(-30, 30, n)   # every exponent
        elif kind == 5: x = np.full(n, 0.125)
        elif kind == 6:
            x = np.abs(rng.standard_normal(n)); x[rng.integers(n)] = np.inf
        elif kind == 7:
            x = np.abs(rng.standard_normal(n)); x[rng.integers(n)] = np.nan
        else: x = np.sort(np.abs(rng.standard_normal(n)))
        with np.errstate(over="ignore"):
            return x.astype(dtype)
    def same(got, want, what):
        got, want = np.asarray(got), np.asarray(want, dtype=got.dtype)
        assert got.tobytes() == want.tobytes() or (np.isnan(got).all() and np.isnan(want).all() and got.shape == want.shape) or \
            np.array_equal(got, want, equal_nan=True), f"{what}: got {got} want {want}"
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        pct = float(rng.choice([99, 99, 50, 0, 100, 99.9, 1, 75.5]))
        form = rng.integers(4)
        if form == 0:      # one list
            dtype = [np.float32, np.float64][rng.integers(2)]
            wide = dtype == np.float32 and rng.integers(2) == 1
            m = int(rng.choice([1, 2, 3, 17, 1000, 32768, 32769, 100003, int(rng.integers(1, 3_000_000))]))
            x = draw(m, dtype)
            got = percentile(torch.from_numpy(x).cuda(), pct, wide=wide).cpu().numpy()
            with np.errstate(invalid="ignore"):
                want = np.percentile(x.astype(np.float64) if wide else x, pct)
            same(got, want, f"[{n}] single {dtype.__name__} wide={wide} n={m} pct={pct}")
        elif form == 1:    # several lists of different lengths
            dtype = [np.float32, np.float64][rng.integers(2)]
            wide = dtype == np.float32 and rng.integers(2) == 1
            k = int(rng.integers(1, 80))
            xs = [draw(int(rng.choice([1, 5, 4096, 40000, int(rng.integers(1, 300_000))])), dtype) for _ in range(k)]
            got = percentile_multi([torch.from_numpy(x).cuda() for x in xs], pct, wide=wide).cpu().numpy()
            with np.errstate(invalid="ignore"):
                want = np.array([np.percentile(x.astype(np.float64) if wide else x, pct) for x in xs])
            same(got, want, f"[{n}] multi {dtype.__name__} wide={wide} k={k} pct={pct}")
        elif form == 2:    # rows of padded matrices (no NaN check in this form: leave NaN out)
            mats, want = [], []
            for _ in range(int(rng.integers(1, 6))):
                r, c = int(rng.integers(1, 40)), int(rng.choice([8, 24, 4096, 14336, int(rng.integers(1, 5000)) * 8]))
                a = np.stack([np.nan_to_num(draw(c, np.float32), nan=1.0) for _ in range(r)])
                pad = torch.zeros(r, c + 8, dtype=torch.float32)
                pad[:, :c] = torch.from_numpy(a)
                mats.append(pad.cuda()[:, :c])
                want += [np.percentile(a[i].astype(np.float64), pct) for i in range(r)]
            got = percentile_rows(mats, pct, wide=True).cpu().numpy()
            same(got, np.array(want), f"[{n}] rows pct={pct}")
        else:              # the selection fed from the quantize pass
            gs = [32, 64, 128, 256, None][rng.integers(5)]
            rows, cols = int(rng.choice([512, 1024, 2048, 1536])), int(rng.choice([1024, 2048, 4096, 1536]))
            B = int(rng.integers(1, 4))
            scale = float(rng.choice([0.02, 1.0, 1e-4]))
            if rng.integers(2):
                w = torch.from_numpy(rng.standard_t(3, (B, rows, cols)).astype(np.float32)) * scale
            else:
                w = torch.from_numpy(rng.standard_normal((B, rows, cols)).astype(np.float32)) * scale
            if rng.integers(4) == 0:
                w[:, ::2] = 0
            fmt, table = [(_lib.INT_ASYM, None), (_lib.INT_SYM, None), (_lib.LUT_FP, "fp8_e4m3")][rng.integers(3)]
            a = quantize_batch(w.half().cuda(), fmt, 8, gs, table, deq=False, codes=False, scale=False, abs_err=torch.float32, p99=pct)
            err = a.abs_err.cpu().numpy()
            want = np.array([np.percentile(err[i].ravel(), pct) for i in range(B)], np.float32)
            same(a.abs_p99.cpu().numpy(), want, f"[{n}] fed fmt={fmt} {B}x{rows}x{cols} g={gs} scale={scale} pct={pct}")
        n += 1
    return n


if __name__ == "__main__":
    t0 = time.time()
    n = run(float(sys.argv[1]), int(sys.argv[2]))
    print("ok:", n, "random cases in", round(time.time() - t0, 1), "s, seed", sys.argv[2], flush=True)
