"""GPU: the statistics-only kernels of the model sweeps (k_stats_group / k_stats_row: no element-sized output)
against the CPU oracle, on every kernel variant and on rows the fast path must hand to the per-element path.

Scales and zero-points: the oracle's bits (NaN == NaN).  Per-segment and per-tensor sums of squared errors and the
sum of absolute errors: within 1e-6 relative of the float64 sums of the oracle's float32 errors
(quant_error_multi_weight.py:55-58 squares in float32; the kernels add a lane's exact float32 squares with
v_dot2_f32_f16 and carry everything above two vectors in float64).  A non-finite expected sum must be non-finite
of the same kind."""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu

# (rows, cols, group size): every template instance the launcher can pick
#   groups of 32 * T elements, T = 1 .. 16 lanes per group, with a last partial wave step
#   rows held by one wave, 2 / 4 / 8 / 12 / 16 vectors per lane, whole and ragged
#   rows held by a 256-thread workgroup, 7 / 8 / 14 / 16 vectors per lane, whole and ragged
SHAPES = [
    (37, 1024, 32), (19, 1024, 64), (21, 2048, 128), (13, 2048, 256), (9, 2048, 512), (3, 96, 32),
    (11, 1024, None), (9, 1000, None), (7, 2048, None), (5, 4096, None), (5, 3000, None), (3, 5632, None),
    (3, 8192, None), (3, 7000, None),
    (3, 14336, None), (2, 16384, None), (2, 15000, None), (2, 28672, None), (2, 32768, None), (2, 30000, None),
]
FORMATS = [("int8", 8), ("int8_asym", 8), ("int4_asym", 4), ("int2", 2), ("fp8_e4m3", 8), ("fp8_e3m4", 8), ("fp4", 4)]


def _inputs(rows, cols, seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(rows, cols, generator=g) * 0.02
    w[0, 3] = 0.5                                   # an outlier
    if rows > 2:
        w[2] = 0                                    # a zero row
    if rows > 4:
        w[3] = w[3].abs() + 0.003                   # a row on one side of zero (asymmetric: per-element path)
        w[4, : cols // 2] = -w[4, : cols // 2].abs()
    if rows > 6:
        w[5, 7] = float("inf")
        w[6, cols - 1] = float("nan")
    if rows > 8:
        w[7] = w[7] * 1e-4                          # float16 subnormals
        w[8, 1] = 65504.0                           # the scale of tiny bit widths reaches the upper clamp
    return w


def _fmt(orc, lib, name, bits):
    if name.startswith("int"):
        asym = "asym" in name
        return (lib.INT_ASYM if asym else lib.INT_SYM, None), dict(fmt=orc.INT_ASYM if asym else orc.INT_SYM, bits=bits)
    return (lib.LUT_FP, name), dict(fmt=orc.LUT, bits=bits, datatype=name)


def _close(got, want, what):
    got, want = np.asarray(got, dtype=np.float64).reshape(-1), np.asarray(want, dtype=np.float64).reshape(-1)
    fin = np.isfinite(want)
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{what}: NaN pattern"
    assert np.array_equal(np.isinf(got), np.isinf(want)), f"{what}: inf pattern"
    err = np.abs(got[fin] - want[fin])
    bad = err > 1e-6 * np.abs(want[fin])
    assert not bad.any(), f"{what}: {bad.sum()} of {fin.sum()} beyond 1e-6, worst {np.max(err / np.maximum(np.abs(want[fin]), 1e-300))}"


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("fmt", FORMATS, ids=[f[0] for f in FORMATS])
def test_statistics_kernels_vs_oracle(oracle, fmt, dt):
    from quantization_amd import _lib
    from quantization_amd.fused import quantize
    name, bits = fmt
    for k, (rows, cols, gs) in enumerate(SHAPES):
        w = _inputs(rows, cols, 100 + k)
        wt = w.half() if dt == "f16" else w.bfloat16()
        (code, table), oargs = _fmt(oracle, _lib, name, bits)
        res = quantize(wt.cuda(), code, bits, gs, table, deq=False, codes=False, scale=True, seg_sse=True, totals=True)
        raw = wt.view(torch.int16).numpy().view(np.uint16)
        exp = oracle.quantize(raw, oracle.F16 if dt == "f16" else oracle.BF16, group_size=gs, **oargs)
        what = f"{name}/{dt}/{rows}x{cols}/g={gs}"
        assert gu.same_f16_bits(res.scale.view(torch.int16).cpu().numpy().view(np.uint16), exp["scale"]), what + " scale"
        if "asym" in name:
            assert gu.same_f16_bits(res.zero_point.view(torch.int16).cpu().numpy().view(np.uint16), exp["zp"]), what + " zp"
        v = wt.half().float().numpy()                       # the reference measures against float16(w)
        with np.errstate(invalid="ignore", over="ignore"):
            e = v - exp["deq"].view(np.float16).astype(np.float32).reshape(rows, cols)
            sq = (e * e).astype(np.float32).astype(np.float64)
            L = cols if gs is None else gs
            _close(res.seg_sse.cpu().numpy(), sq.reshape(-1, L).sum(axis=1), what + " seg_sse")
            tot = res.totals.cpu().numpy()
            _close(tot[0], sq.sum(), what + " sse")
            _close(tot[1], np.abs(e).astype(np.float64).sum(), what + " sum|err|")


@pytest.mark.parametrize("fmt", gu.FORMATS)
@pytest.mark.parametrize("gs", [None, 64], ids=gu.gs_key)
def test_statistics_kernels_special_rows_golden(fmt, gs):
    """Scales / zero-points of the statistics kernels on the reference-generated rows with +-inf / NaN, rows on
    one side of zero, constant rows and float16 limits (tests/golden/quant_specials.npz)."""
    from quantization_amd import _lib
    from quantization_amd.fused import quantize
    w, exp = gu.specials_expected(fmt, gs)
    wt = torch.from_numpy(w.view(np.int16).copy()).view(torch.float16).cuda()
    code = {"int8": _lib.INT_SYM, "int8_asym": _lib.INT_ASYM}.get(fmt, _lib.LUT_FP)
    res = quantize(wt, code, 8, gs, fmt if code == _lib.LUT_FP else None, deq=False, codes=False, scale=True,
                   seg_sse=True, totals=True)
    assert gu.same_f16_bits(res.scale.view(torch.int16).cpu().numpy().view(np.uint16), exp["scale"]), "scale"
    if "zp" in exp:
        assert gu.same_f16_bits(res.zero_point.view(torch.int16).cpu().numpy().view(np.uint16), exp["zp"]), "zp"
    with np.errstate(invalid="ignore", over="ignore"):
        e = w.view(np.float16).astype(np.float32) - exp["deq"].view(np.float16).astype(np.float32).reshape(w.shape)
        sq = (e * e).astype(np.float32).astype(np.float64)
        _close(res.seg_sse.cpu().numpy(), sq.reshape(-1, w.shape[1] if gs is None else gs).sum(axis=1), "seg_sse")


def test_float16_error_is_exact_on_every_pattern():
    """The statistics kernels take v - d in float16 (exact by Sterbenz on the fast path: the CPU proof is
    tests/tools/sterbenz_exhaustive.c).  The general kernels expose both forms of the same error -- |f32(v) -
    f32(d)| and the float16-arithmetic |h(v - d)| -- so on the matrix of all float16 bit patterns the two dumps
    must agree bit for bit wherever the segment is finite."""
    from quantization_amd import _lib
    from quantization_amd.fused import quantize
    w, _ = gu.exhaustive_expected("int8", None)
    wt = torch.from_numpy(w.view(np.int16).copy()).view(torch.float16).cuda()
    for code, table, bits in [(_lib.INT_SYM, None, 8), (_lib.INT_ASYM, None, 8), (_lib.INT_SYM, None, 4),
                              (_lib.LUT_FP, "fp8_e3m4", 8), (_lib.LUT_FP, "fp4", 4)]:
        for gs in (None, 128, 32):
            a = quantize(wt, code, bits, gs, table, deq=False, codes=False, scale=False, abs_err=torch.float32)
            b = quantize(wt, code, bits, gs, table, deq=False, codes=False, scale=False, abs_err=torch.float16)
            if code == _lib.INT_ASYM:
                # rows of one sign are off the fast path (their clamped codes may lie far from v): leave them out
                L = wt.shape[1] if gs is None else gs
                seg = wt.float().reshape(-1, L)
                keep = ((seg.min(dim=1).values <= 0) & (seg.max(dim=1).values >= 0)).repeat_interleave(L).reshape(wt.shape)
            else:
                keep = torch.ones_like(wt, dtype=torch.bool)
            assert torch.equal(a.abs_err[keep], b.abs_err[keep].float()), (code, table, bits, gs)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_segment_mse_float32_form(dt):
    """seg_mse (mq_quant_out.seg_mse_f32 = 1): every kernel scope writes float32(sum * (1 / L)) of the float64 sum
    it writes as seg_sse -- statistics-only kernels, the kernels with element outputs, the per-element kernel
    (ragged L), per-tensor scope and the explicit-list table kernels."""
    from quantization_amd import _lib, fused
    from quantization_amd.fused import quantize, quantize_batch, quantize_table

    def check(a, b, L, what):
        want = (b.double() * (1.0 / L)).float()
        assert a.dtype == torch.float32 and a.shape == b.shape, what
        assert torch.equal(a.view(torch.int32), want.view(torch.int32)), what

    for k, (rows, cols, gs) in enumerate(SHAPES + [(5, 1001, None), (4, 512, -1)]):
        w = _inputs(rows, cols, 300 + k)
        wt = (w.half() if dt == "f16" else w.bfloat16()).cuda()
        L = rows * cols if gs == -1 else (cols if gs is None else gs)
        for code, table, bits in ((_lib.INT_SYM, None, 8), (_lib.INT_ASYM, None, 8), (_lib.LUT_FP, "fp8_e4m3", 8),
                                  (_lib.LUT_FP, "fp4", 4)):
            for outs in (dict(deq=False, codes=False, scale=False), dict(deq=True, codes=True, scale=True)):
                a = quantize(wt, code, bits, gs, table, seg_mse=True, totals=True, **outs)
                b = quantize(wt, code, bits, gs, table, seg_sse=True, totals=True, **outs)
                check(a.seg_mse, b.seg_sse, L, (rows, cols, gs, code, table, tuple(outs.values())))
                assert torch.equal(a.totals.view(torch.int64), b.totals.view(torch.int64))
    w = (torch.randn(3, 24, 256, generator=torch.Generator().manual_seed(5)) * 0.02).to(
        torch.float16 if dt == "f16" else torch.bfloat16).cuda()
    a = quantize_batch(w, _lib.INT_SYM, 8, 64, deq=False, codes=False, scale=False, seg_mse=True)
    b = quantize_batch(w, _lib.INT_SYM, 8, 64, deq=False, codes=False, scale=False, seg_sse=True)
    check(a.seg_mse, b.seg_sse, 64, "batch")
    for gs in (64, None, -1):
        a = quantize_table(w[0], fused.TABLE_LISTS["flint4"], _lib.TABLE_F16, gs, seg_mse=True)
        b = quantize_table(w[0], fused.TABLE_LISTS["flint4"], _lib.TABLE_F16, gs, seg_sse=True)
        check(a.seg_mse, b.seg_sse, 24 * 256 if gs == -1 else (gs or 256), f"table {gs}")
    with pytest.raises(ValueError):
        quantize(w[0], _lib.INT_SYM, 8, 64, seg_sse=True, seg_mse=True)


def test_packed_division_selfcheck_on_device():
    """mq_selfcheck_div16: the packed float16 division of the statistics kernels against the float32 division of
    the general kernels, every admitted (scale, value) pair, with this device's own reciprocal instruction (the CPU
    enumeration with emulated float16 arithmetic is tests/test_div16_exhaustive.py)."""
    import ctypes
    from quantization_amd import _lib
    L = _lib.lib()
    L.mq_selfcheck_div16.restype = ctypes.c_int
    L.mq_selfcheck_div16.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    counts = torch.full((3,), -1, dtype=torch.int64, device="cuda")
    _lib.check(L.mq_selfcheck_div16(counts.data_ptr(), torch.cuda.current_stream().cuda_stream))
    pairs, bad_rint, bad_bits = counts.cpu().tolist()
    assert pairs > 1_200_000_000 and bad_rint == 0 and bad_bits == 0, (pairs, bad_rint, bad_bits)


@pytest.mark.parametrize("j", [0, 1, 2])
def test_packed_division_with_a_shared_power_of_two_selfcheck_on_device(j):
    """mq_selfcheck_div16_shared: the same enumeration with the power of two of a scale up to 4 x larger in place of
    the scale's own (what the several-groupings kernel offers the groups of a row): rint never differs, the bits
    never differ from |v / s| = 2^(j - 4) on."""
    import ctypes
    from quantization_amd import _lib
    L = _lib.lib()
    L.mq_selfcheck_div16_shared.restype = ctypes.c_int
    L.mq_selfcheck_div16_shared.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    counts = torch.full((3,), -1, dtype=torch.int64, device="cuda")
    _lib.check(L.mq_selfcheck_div16_shared(counts.data_ptr(), j, torch.cuda.current_stream().cuda_stream))
    pairs, bad_rint, bad_bits = counts.cpu().tolist()
    assert pairs > 1_200_000_000 and bad_rint == 0 and bad_bits == 0, (j, pairs, bad_rint, bad_bits)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("fmt", FORMATS, ids=[f[0] for f in FORMATS])
def test_numpy_order_list_entries_are_bit_exact(oracle, fmt, dt):
    """seg_mse="numpy": every entry of the per-group MSE list has the BITS of compute_mse's
    float(np.mean(diff * diff)) on float32(weight) and float32(dequantized weight)
    (quant_error_single_weight.py:46-48, :64-71) -- numpy itself is the judge here, on the oracle's dequantized
    values; rows with inf / NaN, zero rows, one-sided rows and float16 subnormals included (a NaN matches a NaN)."""
    from quantization_amd import _lib
    from quantization_amd.fused import quantize
    name, bits = fmt
    for k, (rows, cols, gs) in enumerate([(37, 1024, 32), (19, 1024, 64), (21, 2048, 128), (3, 96, 32), (9, 4096, 128)]):
        w = _inputs(rows, cols, 500 + k)
        wt = w.half() if dt == "f16" else w.bfloat16()
        (code, table), oargs = _fmt(oracle, _lib, name, bits)
        res = quantize(wt.cuda(), code, bits, gs, table, deq=False, codes=False, scale=False, seg_mse="numpy", totals=True)
        raw = wt.view(torch.int16).numpy().view(np.uint16)
        exp = oracle.quantize(raw, oracle.F16 if dt == "f16" else oracle.BF16, group_size=gs, **oargs)
        a = wt.half().float().numpy()
        b = exp["deq"].view(np.float16).astype(np.float32).reshape(rows, cols)
        with np.errstate(invalid="ignore", over="ignore"):
            want = np.array([np.mean((a[r, g:g + gs] - b[r, g:g + gs]) * (a[r, g:g + gs] - b[r, g:g + gs]))
                             for r in range(rows) for g in range(0, cols, gs)], dtype=np.float32)
        got = res.seg_mse.cpu().numpy()
        assert got.dtype == np.float32 and got.shape == want.shape
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (name, dt, rows, cols, gs, int((~same).sum()), got[~same][:4], want[~same][:4])
    with pytest.raises(ValueError):   # calls with element outputs: not offered
        quantize(wt.cuda(), code, bits, 128, table, deq=True, codes=False, scale=False, seg_mse="numpy")


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("fmt", FORMATS, ids=[f[0] for f in FORMATS])
def test_numpy_order_row_entries_are_bit_exact(oracle, fmt, dt):
    """seg_mse="numpy" on rows (per-channel, and groups longer than 512 elements): every entry has the BITS of
    float(np.mean(diff * diff)) over the row slice (quant_error_single_weight.py:46-48, :56-60) -- numpy's pairwise
    summation with its split of runs longer than 128 elements, numpy itself as judge.  Row lengths: powers of two,
    the Llama widths (14336 = 128 leaves of 112; 11008 and 5632 split into unequal leaves), lengths that are only
    a multiple of 8, one past the inner-loop size of numpy's reduction (8192), and a long group."""
    from quantization_amd import _lib
    from quantization_amd.fused import quantize
    name, bits = fmt
    cases = [(9, 4096, None), (5, 14336, None), (4, 11008, None), (6, 5632, None), (7, 1000, None), (3, 28672, None),
             (5, 520, None), (2, 32768, None), (4, 2600, None), (6, 4096, 1024), (5, 2048, 256), (3, 1536, 512),
             (7, 960, 96), (4, 200, None), (9, 1024, 8), (3, 480, 40)]
    for k, (rows, cols, gs) in enumerate(cases):
        w = _inputs(rows, cols, 900 + k)
        wt = w.half() if dt == "f16" else w.bfloat16()
        (code, table), oargs = _fmt(oracle, _lib, name, bits)
        res = quantize(wt.cuda(), code, bits, gs, table, deq=False, codes=False, scale=True, seg_mse="numpy", totals=True)
        ref = quantize(wt.cuda(), code, bits, gs, table, deq=False, codes=False, scale=True, seg_mse=True, totals=True)
        raw = wt.view(torch.int16).numpy().view(np.uint16)
        exp = oracle.quantize(raw, oracle.F16 if dt == "f16" else oracle.BF16, group_size=gs, **oargs)
        a = wt.half().float().numpy()
        b = exp["deq"].view(np.float16).astype(np.float32).reshape(rows, cols)
        L = gs or cols
        with np.errstate(invalid="ignore", over="ignore"):
            want = np.array([np.mean((a[r, g:g + L] - b[r, g:g + L]) * (a[r, g:g + L] - b[r, g:g + L]))
                             for r in range(rows) for g in range(0, cols, L)], dtype=np.float32)
        got = res.seg_mse.cpu().numpy()
        assert got.dtype == np.float32 and got.shape == want.shape
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (name, dt, rows, cols, gs, int((~same).sum()), got[~same][:4], want[~same][:4])
        # scales: those of the bandwidth kernels; totals: float32 partial sums over other runs of elements, the
        # 1e-6 bar of every sum (NaN alike)
        assert torch.equal(res.scale.view(torch.int16), ref.scale.view(torch.int16))
        x, y = res.totals, ref.totals
        assert bool(((x == y) | (x.isnan() & y.isnan()) | ((x - y).abs() <= 1e-6 * y.abs())).all()), (x, y)
    with pytest.raises(ValueError):   # past the longest row the split program holds
        quantize(torch.zeros(2, 65536, dtype=torch.float16, device="cuda"), code, bits, None, table, deq=False,
                 codes=False, scale=False, seg_mse="numpy")


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("fmt", FORMATS, ids=[f[0] for f in FORMATS])
def test_numpy_order_tensor_mse_is_bit_exact(oracle, fmt, dt):
    """report.tensor_mse_numpy: float(np.mean((a32 - b32) ** 2)) over the whole 2-D weight (compute_mse,
    quant_error_multi_weight.py:55-58) bit for bit -- numpy's float32 summation over the flattened tensor, whose
    8192-element chunks cross rows and groups -- from the scales / zero-points a statistics pass wrote.  numpy itself
    is the judge, on the oracle's dequantized values.  Shapes: whole chunks only, a last chunk shorter than 8192 (split
    into unequal leaves), rows that straddle chunks, one chunk, less than one leaf; per-channel, per-group, per-tensor;
    inf / NaN / zero / one-sided rows (a NaN matches a NaN)."""
    from quantization_amd import _lib, report
    from quantization_amd.fused import make_cfg, quantize
    name, bits = fmt
    cases = [(9, 4096, None), (5, 14336, 128), (4, 11008, None), (7, 1000, None), (9, 1000, 40), (3, 100 * 8, 32),
             (2, 64, None), (16, 512, -1), (11, 2600, 8), (33, 4096, 64), (1, 8, None)]
    for k, (rows, cols, gs) in enumerate(cases):
        w = _inputs(rows, cols, 1300 + k)
        wt = w.half() if dt == "f16" else w.bfloat16()
        (code, table), oargs = _fmt(oracle, _lib, name, bits)
        res = quantize(wt.cuda(), code, bits, gs, table, deq=False, codes=False, scale=True, seg_mse=True, totals=True)
        out = torch.zeros((), dtype=torch.float32, device="cuda")
        report.tensor_mse_numpy(wt.cuda(), make_cfg(code, bits, gs, table), res.scale, res.zero_point, out)
        raw = wt.view(torch.int16).numpy().view(np.uint16)
        exp = oracle.quantize(raw, oracle.F16 if dt == "f16" else oracle.BF16, group_size=gs, **oargs)
        a = wt.half().float().numpy()
        b = exp["deq"].view(np.float16).astype(np.float32).reshape(rows, cols)
        with np.errstate(invalid="ignore", over="ignore"):
            want = np.float32(np.mean((a - b) ** 2))
        got = out.cpu().numpy()
        assert got.tobytes() == want.tobytes() or (np.isnan(got) and np.isnan(want)), (name, dt, rows, cols, gs, got, want)
        # and against the float64 sum of the bandwidth kernels: the 1e-6 bar
        ref = float(res.totals[0].item()) / (rows * cols)
        assert np.isnan(ref) == bool(np.isnan(got)) and (np.isnan(ref) or abs(float(got) - ref) <= 1e-6 * ref)


def test_hardware_fp8_conversion_does_not_round_like_the_reference_table():
    """mq_selfcheck_fp8_cvt: the CDNA4 packed float32 -> fp8 e4m3 conversion against the reference's FP8_E4M3 table
    rounding (quant_weight.py:114-138, :577-601) on every float16 x with |x| <= 480.  They differ -- on ties (round
    to nearest even against ties toward -inf), above 448 (the hardware format ends there, the table goes to 480) and
    below 2^-6 (hardware subnormals against the table's e = 0 binade) -- which is why e4m3 is rounded in integer bit
    arithmetic.
    The counts are recorded here so that a toolchain or hardware change shows up."""
    import ctypes
    from quantization_amd import _lib
    L = _lib.lib()
    L.mq_selfcheck_fp8_cvt.restype = ctypes.c_int
    L.mq_selfcheck_fp8_cvt.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    counts = torch.full((5,), -1, dtype=torch.int64, device="cuda")
    _lib.check(L.mq_selfcheck_fp8_cvt(counts.data_ptr(), torch.cuda.current_stream().cuda_stream))
    checked, differ, ties, top, low = counts.cpu().tolist()
    print("fp8 e4m3 hardware conversion vs table:", checked, differ, ties, top, low)
    assert checked == 2 * (0x5f80 + 1)          # every float16 pattern of either sign up to 480
    assert differ > 0 and ties > 0 and top > 0  # the conversion is not the table's rounding
    assert ties + top + low >= differ           # and every difference is one of the three known kinds
