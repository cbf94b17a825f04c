"""GPU: the numpy-order sweep launches (SweepPlan(numpy_order=True): k_np_group / k_np_row + the finishing kernels)
against numpy itself on the oracle's dequantized values -- the three numbers of a CSV row of the reference's reports:

    mse_tensor = float(np.mean((a32 - b32) ** 2))            MSE&P99/quant_error_multi_weight.py:55-58
    mse_rep    = float(vec.mean()),  vec = np.array(per_vec)   MSE&P99/quant_error_single_weight.py:74-78
    mse_p99    = float(np.percentile(vec, 99))                 :79
    per_vec[i] = compute_mse(row / (row, group) slice)          :56-71

Everything is compared bit for bit (a NaN matches a NaN).  The first test walks the kernel variants on small
shapes with rows off the fast path; the second one runs the sizes the benchmark runs (4096 x 4096 and the
128256 x 4096 lm_head of Llama-3-8B), where the reference's float32 sums have drifted from float64 sums by more
than 1e-6 -- the default mode's distance from numpy's value is recorded there with an upper bound."""
import numpy as np
import pytest
import torch

from test_gpu_stats_kernels import _fmt, _inputs

pytestmark = pytest.mark.gpu


def _same(got, want):
    got, want = np.asarray(got), np.asarray(want)
    assert got.dtype == want.dtype and got.shape == want.shape, (got.dtype, want.dtype, got.shape, want.shape)
    u = {4: np.uint32, 8: np.uint64}[got.dtype.itemsize]
    return (got.view(u) == want.view(u)) | (np.isnan(got) & np.isnan(want))


def _numpy_row(a, b, L):
    """(mse_tensor, per_vec float32 array, mse_rep, mse_p99) as the scripts compute them, slice by slice."""
    with np.errstate(invalid="ignore", over="ignore"):
        mse_tensor = np.float64(float(np.mean((a - b) ** 2)))
        a1, b1 = a.reshape(-1, L), b.reshape(-1, L)
        per = np.array([np.mean((a1[i] - b1[i]) ** 2) for i in range(a1.shape[0])], dtype=np.float32)
        vec = np.array([float(x) for x in per])
        return mse_tensor, per, np.float64(float(vec.mean())), np.float64(float(np.percentile(vec, 99)))


SMALL = [(64, 1024), (16, 4096), (8, 14336), (16, 5632), (32, 11008), (4, 28672), (4, 32768), (64, 2048), (128, 256),
         (64, 512), (32, 1536)]
PASSES = [("int8", 8, 32), ("int8", 8, 64), ("int8", 8, 128), ("int8", 8, None), ("int8_asym", 8, 32), ("int8_asym", 8, None),
          ("int4_asym", 4, 128), ("fp8_e4m3", 8, 64), ("fp8_e4m3", 8, None), ("fp8_e3m4", 8, 128), ("fp4", 4, 32),
          ("fp4", 4, None), ("int2", 2, 64), ("int8", 8, 256), ("fp8_e3m4", 8, 256),
          # (second and third group sizes of a format: those passes share one read, k_np_group3)
          ("int8_asym", 8, 128), ("fp8_e4m3", 8, 32), ("fp4", 4, 64)]


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_numpy_sweep_small_shapes(oracle, dt):
    from quantization_amd import _lib, report
    ws16 = []
    gen = torch.Generator().manual_seed(31)
    for k, (rows, cols) in enumerate(SMALL):          # finite weights: every sum is a number whose bits must match
        w = torch.randn(rows, cols, generator=gen) * 0.02
        w[0, 3] = 0.5
        w[1] = 0
        ws16.append(w.half() if dt == "f16" else w.bfloat16())
    for k, (rows, cols) in enumerate(SMALL[:5] + [(64, 2048)]):   # rows off the fast path: inf, NaN, one-sided, clamped scales
        w = _inputs(rows, cols, 4000 + k)
        ws16.append(w.half() if dt == "f16" else w.bfloat16())
    dev = [w.cuda() for w in ws16]
    plan = report.SweepPlan(dev, PASSES, numpy_order=True)
    rec = plan.run().cpu().numpy()
    mt = plan.mse_tensor.cpu().numpy()
    fast = 0
    for k, (name, bits, gs) in enumerate(PASSES):
        kinds = [la[0] for la in plan.plan[k]["launches"] if la[0] != "npgroups"]
        fast += kinds.count("npsweep") + len(plan.plan[k]["np_chunks"])
        if gs in (32, 64, 128):   # one numpy-order launch takes every layer of a per-group pass
            shared = sum(1 for n2, b2, g2 in PASSES if (n2, b2) == (name, bits) and g2 in (32, 64, 128)) > 1
            # (the passes of a format that share a read have handed their layers to the family's chunk launches)
            assert kinds == ([] if shared else ["npsweep"]) and bool(plan.plan[k]["np_chunks"]) == shared, (name, gs, kinds)
        (_, _), oargs = _fmt(oracle, _lib, name, bits)
        for i, wt in enumerate(ws16):
            rows, cols = wt.shape
            L = gs or cols
            raw = wt.view(torch.int16).numpy().view(np.uint16)
            exp = oracle.quantize(raw, oracle.F16 if dt == "f16" else oracle.BF16, group_size=gs, **oargs)
            a = wt.half().float().numpy()
            b = exp["deq"].view(np.float16).astype(np.float32).reshape(rows, cols)
            want_mt, want_vec, want_rep, want_p99 = _numpy_row(a, b, L)
            got_vec = plan.lists(k)[i].cpu().numpy()
            ok = _same(got_vec, want_vec)
            assert ok.all(), (name, dt, gs, wt.shape, int((~ok).sum()), got_vec[~ok][:4], want_vec[~ok][:4])
            assert _same(mt[k, i], want_mt).all(), (name, dt, gs, wt.shape, "mse_tensor", mt[k, i], want_mt)
            assert _same(rec[k, i, 3], want_rep).all(), (name, dt, gs, wt.shape, "mse_rep", rec[k, i, 3], want_rep)
            assert _same(rec[k, i, 4], want_p99).all(), (name, dt, gs, wt.shape, "mse_p99", rec[k, i, 4], want_p99)
            # the float64 tensor sums of the record stay the 1e-6 kind
            e = (a - b).astype(np.float64)
            with np.errstate(invalid="ignore", over="ignore"):
                sse, sab = (e * e).sum(), np.abs(e).sum()
            if np.isfinite(sse):
                assert abs(rec[k, i, 1] - sse) <= 1e-6 * sse and abs(rec[k, i, 2] - sab) <= 1e-6 * sab, (name, gs, wt.shape)
    assert fast >= len(PASSES)   # (the per-channel passes add one launch per column count that the kernels take)


def _full_size_case(oracle, shape, name, bits, groupings, seed):
    from quantization_amd import _lib, report
    g = torch.Generator(device="cuda").manual_seed(seed)
    w = (torch.randn(*shape, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    passes = [(name, bits, gs) for gs in groupings]
    plan = report.SweepPlan([w], passes, numpy_order=True)
    rec = plan.run()[:, 0].cpu().numpy()
    mt = plan.mse_tensor[:, 0].cpu().numpy()
    fast_rec = report.SweepPlan([w], passes).run()[:, 0].cpu().numpy()
    raw = w.cpu().view(torch.int16).numpy().view(np.uint16)
    a = w.cpu().half().float().numpy()
    (_, _), oargs = _fmt(oracle, _lib, name, bits)
    out = []
    for k, gs in enumerate(groupings):
        shared = gs is not None and sum(1 for g2 in groupings if g2 is not None) > 1
        assert [la[0] for la in plan.plan[k]["launches"] if la[0] != "npgroups"] == ([] if shared else ["npsweep"])
        assert bool(plan.plan[k]["np_chunks"]) == shared
        exp = oracle.quantize(raw, oracle.BF16, group_size=gs, nthreads=16, **oargs)
        b = exp["deq"].view(np.float16).astype(np.float32).reshape(shape)
        d = a - b
        d *= d                                              # (a32 - b32) ** 2, float32
        want_mt = np.float64(float(np.mean(d)))
        L = gs or shape[1]
        if gs is None:   # rows: compute_mse per row slice, as the scripts' loop (quant_error_single_weight.py:56-60)
            per = np.array([np.mean(d[r]) for r in range(shape[0])], dtype=np.float32)
        else:            # groups: one leaf of numpy's summation each; the vectorised reduction has the loop's bits
            per = np.mean(d.reshape(-1, L), axis=1, dtype=np.float32)
            probe = np.array([np.mean(d.reshape(-1, L)[i]) for i in range(0, per.size, max(1, per.size // 4096))], np.float32)
            assert np.array_equal(probe.view(np.uint32), per[::max(1, per.size // 4096)].view(np.uint32))
        vec = per.astype(np.float64)
        want_rep, want_p99 = np.float64(float(vec.mean())), np.float64(float(np.percentile(vec, 99)))
        got_vec = plan.lists(k)[0].cpu().numpy()
        ok = _same(got_vec, per)
        assert ok.all(), (shape, name, gs, int((~ok).sum()))
        assert _same(mt[k], want_mt).all(), (shape, name, gs, "mse_tensor", mt[k], want_mt)
        assert _same(rec[k, 3], want_rep).all(), (shape, name, gs, "mse_rep", rec[k, 3], want_rep)
        assert _same(rec[k, 4], want_p99).all(), (shape, name, gs, "mse_p99", rec[k, 4], want_p99)
        # the default (float64-sum) mode against the same numpy values: relative distances
        dist = (abs(fast_rec[k, 1] / fast_rec[k, 0] - want_mt) / want_mt, abs(fast_rec[k, 3] - want_rep) / want_rep,
                abs(fast_rec[k, 4] - want_p99) / want_p99)
        out.append((gs, dist))
    return out


def test_numpy_sweep_4096_square(oracle):
    """16.8 M elements.  The default mode's mse_tensor is the float64 sum: numpy's float32 sum over 2048 chunks is up to
    a few 1e-6 away from it (measured 2.1e-6 on these sizes) -- bounded here at 1e-5; mse_rep / mse_p99 stay inside
    1e-6."""
    for name, bits in (("int8", 8), ("int8_asym", 8), ("fp8_e4m3", 8)):
        for gs, (d_mt, d_rep, d_p99) in _full_size_case(oracle, (4096, 4096), name, bits, [32, 128, None], 11):
            assert d_mt < 1e-5 and d_rep < 1e-6 and d_p99 < 1e-6, (name, gs, d_mt, d_rep, d_p99)


def test_numpy_sweep_lm_head(oracle):
    """The 128256 x 4096 lm_head of Llama-3-8B (525 M elements, 64 128 chunks in numpy's float32 tensor sum)."""
    for gs, (d_mt, d_rep, d_p99) in _full_size_case(oracle, (128256, 4096), "int8", 8, [32, None], 12):
        assert d_mt < 2e-5 and d_rep < 1e-6 and d_p99 < 1e-6, (gs, d_mt, d_rep, d_p99)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_group_sizes_sharing_one_read_equal_separate_launches(dt):
    """numpy order: the passes of a format with groups of 32 / 64 / 128 from one read (mq_sweep_numpy_run_groups) against
    one launch per pass: lists, records and tensor means bit for bit -- every format, two and three sizes, weights
    with segments off the fast path."""
    from quantization_amd import report
    g = torch.Generator(device="cuda").manual_seed(77)
    ws = []
    for shape in [(1024, 4096), (512, 14336), (256, 2048), (4096, 1024)]:
        w = torch.randn(*shape, generator=g, device="cuda") * 0.02
        ws.append(w)
    ws[0][3, 64:96] = 0.0                # an all-zero group of 32
    ws[1][7, 128:256] = 1.0              # one-sided groups (asymmetric: off the fast path)
    ws[2][5, 17] = float("inf")
    ws[2][9, 300] = float("nan")
    ws[3][100, :64] = -0.0
    ws[3][100, 1] = 0.0
    ws = [w.to(dt) for w in ws]
    for passes in ([("int8", 8, 32), ("int8", 8, 64), ("int8", 8, 128), ("int8", 8, None)],
                   [("int8_asym", 8, 128), ("int8_asym", 8, 32), ("fp8_e4m3", 8, 64), ("fp8_e4m3", 8, 128)],
                   [("fp8_e3m4", 8, 32), ("fp8_e3m4", 8, 64), ("fp4", 4, 32), ("fp4", 4, 128), ("int4_asym", 4, 64),
                    ("int4_asym", 4, 128)]):
        a = report.SweepPlan(ws, passes, numpy_order=True, fuse_groupings=False)
        c = report.SweepPlan(ws, passes, numpy_order=True)
        assert a.np_families == 0 and c.np_families == len({(n, b) for n, b, gs in passes if gs})
        ra, rc = a.run().cpu().numpy(), c.run().cpu().numpy()
        torch.cuda.synchronize()
        assert ra.tobytes() == rc.tobytes(), passes
        assert a.mse_tensor.cpu().numpy().tobytes() == c.mse_tensor.cpu().numpy().tobytes(), passes
        for k in range(len(passes)):
            for i in range(len(ws)):
                assert a.lists(k)[i].cpu().numpy().tobytes() == c.lists(k)[i].cpu().numpy().tobytes(), (passes[k], i)
        assert c.read_bytes < a.read_bytes
