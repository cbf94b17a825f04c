"""Randomised differential run of the error half (compute_errors both forms, QuantizationErrorAnalyzer, bucketize,
compare_formats / to_ranges, minmax, histogram) against the numpy restatement of the analysis scripts
(oracle/oracle.py) and numpy itself: percentiles, masks, bucket indices, histograms and ranges bit for bit; sums
and means to 1e-6.  run(seconds, seed); as a script: argv = seconds, seed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import oracle as orc
from quantization_amd import error_metrics as em


def _close(got, want, what, tol=1e-6):
    got, want = float(got), float(want)
    if np.isnan(want) or np.isinf(want):
        assert (np.isnan(got) and np.isnan(want)) or got == want, f"{what}: got {got} want {want}"
        return
    assert abs(got - want) <= tol * abs(want) + 1e-300, f"{what}: got {got!r} want {want!r}"


def run(budget, seed):
    rng = np.random.default_rng(seed)
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        fam = ["multi", "single", "analyzer", "bucket", "compare", "hist"][rng.integers(6)]
        K = int(rng.integers(1, 60))
        C = int(rng.choice([int(rng.integers(1, 500)), 8 * int(rng.integers(1, 200)), 1001, 1024, 4096]))
        scale = float(rng.choice([0.02, 1.0, 1e-3, 50.0]))
        dt = [np.float16, np.float32][rng.integers(2)]
        a = (rng.standard_normal((K, C)) * scale).astype(dt)
        noise = (rng.standard_normal((K, C)) * scale * float(rng.choice([1e-2, 1e-3, 0.3]))).astype(np.float32)
        b = (a.astype(np.float32) + noise).astype(dt)
        kind = rng.integers(6)
        if kind == 0: b[rng.integers(K)] = a[rng.integers(K)]            # exact rows
        if kind == 1: b = a.copy()                                        # zero error everywhere
        if kind == 2 and fam in ("multi", "single", "analyzer"): b[rng.integers(K), rng.integers(C)] = np.nan
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        what = f"[{n}] {fam}/{np.dtype(dt).name}/{K}x{C}/scale={scale}/kind={kind}"
        with np.errstate(all="ignore"):
            if fam in ("multi", "single"):
                divs = [d for d in (1, 2, 4, 8, 16, 32, 64, 128) if C % d == 0]
                grouping, gs = ("per-channel", None) if rng.integers(2) else ("per-group", int(rng.choice(divs)))
                what += f"/{grouping}/{gs}"
                want = (orc.compute_errors_multi if fam == "multi" else orc.compute_errors_single)(a, b, grouping, gs)
                got = em.compute_errors(ta, tb, grouping, gs, form=fam)
                for k in ("mse_tensor", "mse_rep", "mse_p99"):
                    _close(got[k], want[k], what + " " + k)
            elif fam == "analyzer":
                rngs = None
                if rng.integers(2):
                    s0 = int(rng.integers(0, K * C)); rngs = (s0, s0 + int(rng.integers(1, K * C + 10)))
                an = em.QuantizationErrorAnalyzer(ta, tb, rngs)
                want = orc.elementwise_f32(a, b)
                got_abs = an.abs_elementwise_error.cpu().numpy()
                assert np.array_equal(got_abs, want["abs"], equal_nan=True), what + " abs"
                gp = an.get_p99_error().cpu().numpy()
                assert gp.dtype == np.float32 and (gp.tobytes() == np.float32(want["p99"]).tobytes() or (np.isnan(gp) and np.isnan(want["p99"]))), what + f" p99 {gp} {want['p99']}"
                _close(an.get_error_sum(), np.sum(want["abs"].astype(np.float64)), what + " sum")
                if rngs:
                    sl = orc.sliced_stats(want["err"], *rngs)
                    e64 = want["err"][rngs[0]:min(rngs[1], K * C)].astype(np.float64)
                    _close(an.get_sliced_mse(), np.mean(np.square(e64.astype(np.float32)).astype(np.float64)), what + " sliced mse")
                    _close(an.get_sliced_abs_error_sum(), np.abs(e64).sum(), what + " sliced sum")
            elif fam == "bucket":
                a16, b16 = a.astype(np.float16), b.astype(np.float16)
                step = float(rng.choice([1e-4, 1e-3, 0.01])); mx = step * int(rng.integers(1, 40))
                want = orc.bucketize(a16, b16, step, mx)
                abs_err, idx, values = em.bucketize(torch.from_numpy(a16).cuda(), torch.from_numpy(b16).cuda(), step, mx)
                assert np.array_equal(abs_err.cpu().numpy().view(np.uint16), want["abs_err"].view(np.uint16)), what + " abs"
                assert np.array_equal(idx.cpu().numpy(), want["idx"].astype(np.uint8)), what + " idx"
                assert np.array_equal(values[want["idx"]], want["values"]), what + " values"
            elif fam == "compare":
                a16 = a.astype(np.float16)
                F = int(rng.integers(1, 6))
                qs = [(a16.astype(np.float32) + rng.standard_normal((K, C)).astype(np.float32) * scale * 0.01 * (f + 1)).astype(np.float16) for f in range(F)]
                if kind == 0 and F > 1: qs[1] = qs[0].copy()            # ties between formats
                thr = float(rng.choice([0.02, 1e-4, scale * 0.02]))
                want = orc.compare_formats(a16, qs, thr)
                got = em.compare_formats(torch.from_numpy(a16).cuda(), [torch.from_numpy(q).cuda() for q in qs], thr)
                assert np.array_equal(got["min_err"].cpu().numpy().view(np.uint16), want["min_err"].view(np.uint16)), what + " min_err"
                assert np.array_equal(got["mask_out"].cpu().numpy(), want["mask_out"]), what + " mask_out"
                for f in range(F):
                    assert np.array_equal(got["best"][f].cpu().numpy(), want["best"][f]), what + f" best {f}"
                assert em.to_ranges(got["mask_out"]) == orc.to_ranges(np.where(want["mask_out"])[0]), what + " ranges"
            else:
                x = a.reshape(-1).astype(np.float32)   # (the scripts widen float16 initializers first: proto_to_array)
                bins = int(rng.choice([10, 50, 200]))
                r = None if rng.integers(2) else (float(x.min()) - 0.1 * scale, float(x.max()) * 0.5)
                cw, ew = np.histogram(x, bins=bins, range=r)
                cg, eg = em.histogram(ta, bins=bins, range=r)
                assert np.array_equal(np.asarray(eg), ew), what + " edges"
                assert np.array_equal(cg.cpu().numpy(), cw), what + " counts"
                mm = em.minmax(ta).cpu().numpy()
                assert mm[0] == x.min() and mm[1] == x.max(), what + " minmax"
        n += 1
    return n


if __name__ == "__main__":
    t0 = time.time()
    n = run(float(sys.argv[1]), int(sys.argv[2]))
    print("ok:", n, "random cases in", round(time.time() - t0, 1), "s, seed", sys.argv[2], flush=True)
