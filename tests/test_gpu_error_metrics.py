"""GPU: percentile (radix select), error statistics, bucketing and format comparison against numpy /
the numpy restatement of the reference's analysis scripts and the stored known answers.
Bit-exact: percentiles of a given array, element-wise errors, bucket indices, masks.
Within 1e-6 relative (float64 vs numpy's float32 pairwise sums): MSE values and sums."""
import hashlib

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def em():
    from quantization_amd import _lib, error_metrics
    _lib.lib()
    assert torch.cuda.is_available()
    return error_metrics


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _arrays(dtype):
    rng = np.random.default_rng(5)
    yield "one", np.array([0.25], dtype)
    yield "two", np.array([3.0, 1.0], dtype)
    yield "seven", np.abs(rng.standard_normal(7)).astype(dtype)
    yield "1000", np.abs(rng.standard_normal(1000)).astype(dtype)
    yield "100003", (np.abs(rng.standard_normal(100003)) * 1e-4).astype(dtype)
    yield "all_equal", np.full(300001, 0.0123, dtype)
    yield "mostly_zero", np.where(rng.random(400000) < 0.995, 0, rng.random(400000)).astype(dtype)
    yield "few_levels", (rng.integers(0, 5, 1 << 20) * np.float64(2.0 ** -12)).astype(dtype)
    yield "wide_range", np.exp(rng.standard_normal(1 << 21) * 8).astype(dtype)
    yield "5M", np.abs(rng.standard_normal(5_000_001) * 0.003).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_percentile_matches_numpy_bits(em, dtype):
    for name, arr in _arrays(dtype):
        x = torch.from_numpy(arr).cuda()
        for pct in (99, 50, 0, 100, 99.9, 12.5):
            want = np.percentile(arr, pct)
            got = em.percentile(x, pct).cpu().numpy()
            assert got.dtype == arr.dtype
            assert got.tobytes() == np.asarray(want, dtype).tobytes(), (name, pct, got, want)


def test_percentile_large_index_rounding(em):
    """n - 1 beyond 2^24: numpy's float32 virtual index is a rounded integer (gamma == 0)."""
    rng = np.random.default_rng(9)
    arr = np.abs(rng.standard_normal(4096 * 4096 + 3)).astype(np.float32)
    got = em.percentile(torch.from_numpy(arr).cuda(), 99).cpu().numpy()
    assert got.tobytes() == np.float32(np.percentile(arr, 99)).tobytes()


def _gold_inputs():
    gold = gu.load_json("error_stats.json")
    K, C = gold["K"], gold["C"]
    rng = np.random.default_rng(gold["seed"])
    inputs = {"normal": (rng.standard_normal((K, C)) * 0.02).astype(np.float16),
              "student_t": (rng.standard_t(3, (K, C)) * 0.02).astype(np.float16)}
    return gold, inputs


def _quant_on_gpu(w16, fmt, gs):
    import quantization_amd as qa
    w = torch.from_numpy(w16).cuda()
    qw = qa.quant_weight
    if fmt == "int8":
        return qw.quant_int(w, 8, gs)
    if fmt == "int8_asym":
        return qw.quant_int_asym(w, 8, gs)
    return qw.quant_datatype(w, 8, fmt, gs)


def test_known_answers(em, oracle):
    gold, inputs = _gold_inputs()
    for cname, w16 in inputs.items():
        g = gold["cases"][cname]
        assert _digest(w16) == g["input_sha256"]
        w = torch.from_numpy(w16).cuda()
        for key, e in g["formats"].items():
            fmt, gk = key.split("/")
            gs = None if gk == "none" else int(gk)
            deq = _quant_on_gpu(w16, fmt, gs)
            assert _digest(deq.view(torch.int16).cpu().numpy().view(np.uint16)) == e["deq_sha256"]
            grouping = "per-channel" if gs is None else "per-group"
            a32, b32 = w.float(), deq.float()
            multi = em.compute_errors(a32, b32, grouping, gs, form="multi")
            single = em.compute_errors(a32, b32, grouping, gs, form="single")
            for got, want in ((multi["mse_tensor"], e["mse_tensor"]), (single["mse_tensor"], e["mse_tensor"]),
                              (multi["mse_rep"], e["multi"]["mse_rep"]), (multi["mse_p99"], e["multi"]["mse_p99"]),
                              (single["mse_rep"], e["single"]["mse_rep"]), (single["mse_p99"], e["single"]["mse_p99"])):
                assert abs(got - want) <= 1e-6 * abs(want), (cname, key, got, want)
            # numpy_order: the stored answers (numpy's own float32 / float64 sums) to the last bit, both forms
            mn = em.compute_errors(a32, b32, grouping, gs, form="multi", numpy_order=True)
            sn = em.compute_errors(a32, b32, grouping, gs, form="single", numpy_order=True)
            for got, want in ((mn["mse_tensor"], e["mse_tensor"]), (sn["mse_tensor"], e["mse_tensor"]),
                              (mn["mse_rep"], e["multi"]["mse_rep"]), (mn["mse_p99"], e["multi"]["mse_p99"]),
                              (sn["mse_rep"], e["single"]["mse_rep"]), (sn["mse_p99"], e["single"]["mse_p99"])):
                assert got == want, (cname, key, "numpy_order", got, want)
            an = em.QuantizationErrorAnalyzer(w, deq, analysis_range=(1000, 200000))
            assert _digest(an.abs_elementwise_error.cpu().numpy()) == e["abs_sha256"]
            p99 = an.get_p99_error().cpu().numpy()
            assert int(p99.view(np.uint32)) == e["abs_p99_f32_bits"], (cname, key)
            want_sum = float(np.uint32(e["abs_sum_f32_bits"]).view(np.float32))
            assert abs(an.get_error_sum() - want_sum) <= 1e-6 * want_sum
            want_smse = float(np.uint32(e["sliced_mse_f32_bits"]).view(np.float32))
            want_ssum = float(np.uint32(e["sliced_abs_sum_f32_bits"]).view(np.float32))
            assert abs(an.get_sliced_mse() - want_smse) <= 1e-6 * want_smse
            assert abs(an.get_sliced_abs_error_sum() - want_ssum) <= 1e-6 * want_ssum
            # numpy_order: the script's float32 sums and sliced mean, to the last bit
            ann = em.QuantizationErrorAnalyzer(w, deq, analysis_range=(1000, 200000), numpy_order=True)
            assert ann.get_error_sum() == want_sum and ann.get_sliced_mse() == want_smse
            assert ann.get_sliced_abs_error_sum() == want_ssum
            abs16, idx, values = em.bucketize(w, deq)
            assert _digest(abs16.view(torch.int16).cpu().numpy().view(np.uint16)) == e["abs_f16_sha256"]
            assert _digest(idx.cpu().numpy().astype(np.int64)) == e["bucket_idx_sha256"]
        # compare.py on the three 8-bit formats, per-channel
        c = g["compare"]
        qs = [_quant_on_gpu(w16, f, None) for f in c["formats"]]
        r = em.compare_formats(w, qs, threshold=0.02)
        assert _digest(r["min_err"].view(torch.int16).cpu().numpy().view(np.uint16)) == c["min_err_sha256"]
        assert int(r["mask_out"].sum()) == c["thr_0.02"]["n_out"]
        for i, b in enumerate(r["best"]):
            assert int(b.sum()) == c["thr_0.02"]["n_best"][i]
            assert _digest(np.packbits(b.cpu().numpy())) == c["thr_0.02"]["best_sha256"][i]
        r2 = em.compare_formats(w, qs, threshold=1e-4)
        assert int(r2["mask_out"].sum()) == c["thr_1e-4"]["n_out"]
        assert _digest(np.packbits(r2["mask_out"].cpu().numpy())) == c["thr_1e-4"]["out_sha256"]
        assert len(em.to_ranges(r2["mask_out"])) == c["thr_1e-4"]["n_ranges_out"]


def test_error_stats_shapes_vs_numpy(em, oracle):
    """Row / column forms on ragged shapes (vector path, short teams, scalar fallback) against the numpy
    restatement evaluated with float64 accumulation."""
    rng = np.random.default_rng(3)
    for (K, C, gs) in [(33, 256, 32), (7, 1024, 1024), (5, 1536, 96), (64, 40, 8), (6, 100, 20), (3, 8192, 128)]:
        a = (rng.standard_normal((K, C)) * 0.02).astype(np.float16)
        b = (a.astype(np.float32) + rng.standard_normal((K, C)).astype(np.float32) * 1e-4).astype(np.float16)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        d = a.astype(np.float32) - b.astype(np.float32)
        sq = (d * d).astype(np.float64)
        for dtype in (torch.float16, torch.float32):
            st = em.error_stats(ta.to(dtype), tb.to(dtype), seg_len=gs, cols=(C % 8 == 0), abs_err=torch.float32)
            np.testing.assert_allclose(st["totals"].cpu().numpy(), [sq.sum(), np.abs(d).astype(np.float64).sum()], rtol=2e-7)
            np.testing.assert_allclose(st["seg_sse"].cpu().numpy(), sq.reshape(-1, gs).sum(1), rtol=2e-7)
            if C % 8 == 0:
                np.testing.assert_allclose(st["col_sse"].cpu().numpy(), sq.sum(0), rtol=1e-12)
            assert np.array_equal(st["abs_err"].cpu().numpy(), np.abs(d))
        m = em.compute_errors(ta.float(), tb.float(), "per-group", gs, form="single")
        want = oracle.compute_errors_single(a.astype(np.float32), b.astype(np.float32), "per-group", gs)
        assert abs(m["mse_p99"] - want["mse_p99"]) <= 1e-6 * want["mse_p99"]
        assert abs(m["mse_rep"] - want["mse_rep"]) <= 1e-6 * want["mse_rep"]
    # to_ranges against the oracle's run-length helper
    mask = torch.from_numpy(rng.random(5000) < 0.3).cuda()
    assert em.to_ranges(mask) == oracle.to_ranges(np.where(mask.cpu().numpy())[0])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_percentile_batch(em, dtype):
    rng = np.random.default_rng(21)
    for B, n in ((1, 1000), (5, 4096), (16, 100000), (40, 2048)):
        arr = (np.abs(rng.standard_normal((B, n))) * rng.random((B, 1))).astype(dtype)
        arr[0, : n // 2] = 0
        got = em.percentile_batch(torch.from_numpy(arr).cuda(), 99).cpu().numpy()
        want = np.array([np.percentile(arr[i], 99) for i in range(B)], dtype)
        assert got.tobytes() == want.tobytes(), (B, n)


def test_percentile_sample_route_falls_back(em):
    """Large float32 arrays are located through a 1/32 sample first (one 256-byte line of every 8 KiB, line
    index = top 5 bits of chunk * 0x9E3779B1).  An array whose sampled lines are all zero makes the sample
    point at the wrong place; the exact route behind it must still give numpy's bits."""
    rng = np.random.default_rng(77)
    n = (1 << 22) + 12345
    arr = np.abs(rng.standard_normal(n)).astype(np.float32)
    chunks = np.arange(n // 2048, dtype=np.uint64)
    line = ((chunks * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) >> np.uint64(27)
    start = (chunks * np.uint64(2048) + line * np.uint64(64)).astype(np.int64)
    idx = (start[:, None] + np.arange(64)[None, :]).ravel()
    fooled = arr.copy()
    fooled[idx] = 0
    x = torch.from_numpy(np.stack([arr[: 1 << 22], fooled[: 1 << 22]])).cuda()
    got = em.percentile_batch(x, 99).cpu().numpy()
    want = np.array([np.percentile(arr[: 1 << 22], 99), np.percentile(fooled[: 1 << 22], 99)], np.float32)
    assert got.tobytes() == want.tobytes()
    for a in (arr, fooled):
        for pct in (99, 50, 1, 99.99):
            got = em.percentile(torch.from_numpy(a).cuda(), pct).cpu().numpy()
            assert got.tobytes() == np.float32(np.percentile(a, pct)).tobytes(), pct


def test_vector_mse_stats(em):
    rng = np.random.default_rng(4)
    sse = (rng.random((3, 4096)) ** 4 * 1e-6).astype(np.float64)
    for L in (32, 128, 96, 4096):
        mean, p99 = em.vector_mse_stats(torch.from_numpy(sse).cuda(), L)
        vec = sse / L
        assert p99.cpu().numpy().tobytes() == np.array([np.percentile(v, 99) for v in vec]).tobytes(), L
        np.testing.assert_allclose(mean.cpu().numpy(), vec.mean(1), rtol=1e-12)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_percentile_multi_mixed_lengths(em, dtype):
    """Arrays of different lengths in one call: small ones (exact route), large float32 ones (sample +
    bracket route), more than 64 arrays (several groups)."""
    rng = np.random.default_rng(31)
    lens = [1, 7, 1000, 4096, (1 << 21) + 16, 100003, (1 << 22), 5, 3_000_000] + [2048 + 4 * i for i in range(70)]
    arrs = [(np.abs(rng.standard_normal(n)) * rng.random()).astype(dtype) for n in lens]
    got = em.percentile_multi([torch.from_numpy(a).cuda() for a in arrs], 99).cpu().numpy()
    want = np.array([np.percentile(a, 99) for a in arrs], dtype)
    assert got.tobytes() == want.tobytes()


def test_compare_formats_vector_and_scalar_paths(em, oracle):
    """compare.py masks on ragged lengths, unaligned views, inf / NaN elements and odd thresholds: the 8-per-lane
    kernel, its scalar tail and the all-scalar route against the numpy restatement."""
    rng = np.random.default_rng(11)
    n = 8 * 1000 + 5
    w = (rng.standard_normal(n) * 0.02).astype(np.float16)
    qs = [(w.astype(np.float32) + rng.standard_normal(n).astype(np.float32) * s).astype(np.float16)
          for s in (1e-3, 2e-3, 1e-3, 5e-4, 3e-3)]
    qs[2][::3] = qs[0][::3]                      # ties between formats
    w[17], qs[1][40], qs[3][99] = np.nan, np.inf, -np.inf
    qs[4][123] = np.nan
    w[200:208] = qs[0][200:208]                  # exact zeros
    tw, tq = torch.from_numpy(w).cuda(), [torch.from_numpy(q).cuda() for q in qs]
    for nf in (1, 3, 5):
        for lo in (0, 1):                        # lo = 1: views that are not 16-byte aligned
            for thr in (0.02, 1e-3, 0.0, -0.0, -1.0, float("nan")):
                with np.errstate(invalid="ignore"):
                    exp = oracle.compare_formats(w[lo:], [q[lo:] for q in qs[:nf]], threshold=np.float16(thr))
                got = em.compare_formats(tw[lo:], [q[lo:] for q in tq[:nf]], threshold=thr)
                gm, wm = got["min_err"].cpu().numpy(), exp["min_err"]
                assert np.array_equal(np.isnan(gm), np.isnan(wm)) and np.array_equal(gm[~np.isnan(gm)], wm[~np.isnan(wm)])
                assert np.array_equal(got["mask_out"].cpu().numpy(), exp["mask_out"]), (nf, lo, thr)
                for f in range(nf):
                    assert np.array_equal(got["best"][f].cpu().numpy(), exp["best"][f]), (nf, lo, thr, f)
    flags = torch.empty(n, dtype=torch.uint8, device="cuda")
    r = em.compare_formats(tw, tq[:3], 0.02, out=flags)
    assert r["flags"] is flags
    assert torch.equal(flags, em.compare_formats(tw, tq[:3], 0.02)["flags"])


def test_short_array_single_launch_route(em):
    """Arrays of up to 32768 values are selected by one workgroup in one launch: lengths around the limit, tiny
    arrays, ties, zeros and infinities, several quantiles, both dtypes, mixed into one percentile_multi call --
    bit for bit against numpy."""
    rng = np.random.default_rng(21)
    for dt in (np.float32, np.float64):
        arrays = []
        for n in (1, 2, 3, 64, 1000, 2047, 32767, 32768, 32769, 70000):
            arrays.append(np.abs(rng.standard_normal(n)).astype(dt) * dt(1e-7))
        arrays.append(np.full(500, 0.25, dt))                                   # all equal
        arrays.append(np.zeros(77, dt))
        arrays.append(np.repeat(np.abs(rng.standard_normal(40)).astype(dt), 250))   # heavy ties
        a = np.abs(rng.standard_normal(3000)).astype(dt)
        a[:40] = np.inf                                                         # the top 1.3 % is infinite
        arrays.append(a)
        arrays.append((rng.random(9999) ** 8).astype(dt) * dt(1e-30))           # tiny magnitudes, wide exponent range
        dev = [torch.from_numpy(x).cuda() for x in arrays]
        for pct in (99, 50, 0, 100, 99.9, 12.5):
            got = em.percentile_multi(dev, pct).cpu().numpy()
            for i, x in enumerate(arrays):
                with np.errstate(invalid="ignore"):
                    want = np.percentile(x, pct)
                if np.isnan(want):
                    assert np.isnan(got[i]), (dt, pct, i)
                else:
                    assert got[i].tobytes() == np.asarray(want, dt).tobytes(), (dt.__name__, pct, i, x.size)
        v, pair = em.percentile(dev[4], 99, return_pair=True)
        srt = np.sort(arrays[4])
        k = (arrays[4].size - 1) * 0.99
        assert pair.cpu().numpy().tolist() == [srt[int(np.floor(k))], srt[int(np.ceil(k))]]


def test_error_stats_any_column_count_and_offset_views(em, oracle):
    """compute_errors(form="multi") -- the per-column loop of quant_error_multi_weight.py:67-78 -- accepts any
    shape: a column count that is not a multiple of 8, storage that starts off a 16-byte boundary, an index
    range with an odd start (the scalar column kernel and the flat totals sweep of mq_error_stats)."""
    rng = np.random.default_rng(5)
    for K, C, gs in [(37, 1001, None), (24, 1001, 143), (16, 1000, 125)]:
        a = (rng.standard_normal((K, C)) * 0.02).astype(np.float16)
        b = (a.astype(np.float32) + rng.standard_normal((K, C)).astype(np.float32) * 1e-3).astype(np.float16)
        # device views whose first element is not 16-byte aligned
        ta = torch.from_numpy(np.concatenate([np.zeros(3, np.float16), a.reshape(-1)])).cuda()[3:].view(K, C)
        tb = torch.from_numpy(np.concatenate([np.zeros(5, np.float16), b.reshape(-1)])).cuda()[5:].view(K, C)
        assert ta.data_ptr() % 16 and tb.data_ptr() % 16
        grouping = "per-channel" if gs is None else "per-group"
        got = em.compute_errors(ta, tb, grouping, gs, form="multi")
        want = oracle.compute_errors_multi(oracle.proto_to_array(a), oracle.proto_to_array(b), grouping, gs)
        for k in ("mse_tensor", "mse_rep", "mse_p99"):
            assert abs(got[k] - want[k]) <= 1e-6 * abs(want[k]), (K, C, gs, k)
    # a flat index range with an odd start and length (QuantizationErrorAnalyzer's sliced statistics)
    x = torch.randn(300001, device="cuda").half()
    y = (x.float() * 1.001).half()
    st = em.error_stats(x[7:300000], y[7:300000])
    e = (x[7:300000].float() - y[7:300000].float()).double()
    torch.testing.assert_close(st["totals"], torch.stack([(e * e).sum(), e.abs().sum()]), rtol=1e-6, atol=0)


def test_percentile_unaligned_views_and_nan(em):
    """A sliced view (storage off a 16-byte boundary) is accepted, and an array holding a NaN gives NaN, as
    np.percentile does."""
    rng = np.random.default_rng(9)
    x = np.abs(rng.standard_normal(100003)).astype(np.float32)
    t = torch.from_numpy(x).cuda()
    got = em.percentile(t[1:], 99)
    assert got.cpu().numpy().tobytes() == np.percentile(x[1:], 99).astype(np.float32).tobytes()
    got = em.percentile_multi([t[1:], t[3:50000]], 99).cpu().numpy()
    assert got[0].tobytes() == np.percentile(x[1:], 99).astype(np.float32).tobytes()
    assert got[1].tobytes() == np.percentile(x[3:50000], 99).astype(np.float32).tobytes()
    x[777] = np.nan
    t = torch.from_numpy(x).cuda()
    assert torch.isnan(em.percentile(t, 99)) and np.isnan(np.percentile(x, 99))
    m = em.percentile_multi([t, t[1000:]], 50).cpu().numpy()
    assert np.isnan(m[0]) and m[1].tobytes() == np.percentile(x[1000:], 50).astype(np.float32).tobytes()
    b = em.percentile_batch(torch.from_numpy(x[:100000].reshape(4, 25000)).cuda(), 99).cpu().numpy()
    want = np.percentile(x[:100000].reshape(4, 25000), 99, axis=1).astype(np.float32)
    assert np.isnan(b[0]) and np.array_equal(b[1:], want[1:])


def test_compute_errors_numpy_order_has_numpys_bits(em, oracle):
    """compute_errors(numpy_order=True) against the scripts' numpy arithmetic (oracle.compute_errors_multi /
    _single restate quant_error_multi_weight.py:61-87 and quant_error_single_weight.py:50-79) on arbitrary (a, b)
    pairs: mse_tensor, mse_rep and mse_p99 equal bit for bit -- shapes whose slices are shorter than a leaf, longer
    than numpy's 8192-element chunks (columns of 9000 rows, blocks of 64 x 300), odd row and column counts."""
    rng = np.random.default_rng(31)
    for K, C, gs in ((300, 64, 32), (9000, 24, 8), (77, 1000, 40), (5, 8, None), (1030, 96, None), (64, 8200, 8200)):
        a = (rng.standard_normal((K, C)) * 0.02).astype(np.float16).astype(np.float32)
        b = (a + rng.standard_normal((K, C)).astype(np.float32) * 1e-4).astype(np.float16).astype(np.float32)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        for grouping in (("per-channel", "per-group") if gs else ("per-channel",)):
            wm = oracle.compute_errors_multi(a, b, grouping, gs)
            ws = oracle.compute_errors_single(a, b, grouping, gs)
            gm = em.compute_errors(ta, tb, grouping, gs, form="multi", numpy_order=True)
            gsn = em.compute_errors(ta, tb, grouping, gs, form="single", numpy_order=True)
            for k in ("mse_tensor", "mse_rep", "mse_p99"):
                assert gm[k] == wm[k], (K, C, gs, grouping, "multi", k, gm[k], wm[k])
                assert gsn[k] == ws[k], (K, C, gs, grouping, "single", k, gsn[k], ws[k])
    one = em.compute_errors(ta.reshape(-1), tb.reshape(-1), "per-tensor", None, numpy_order=True)
    assert one == {"mse_tensor": float(np.mean((a.reshape(-1) - b.reshape(-1)) ** 2))}
