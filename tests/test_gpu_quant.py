"""GPU: the HIP quantize path (through the C ABI, via the reference-shaped Python API) against the
golden vectors and against the CPU oracle on seeded inputs.  Bit-exact for codes, scales, zero-points,
dequantized values and element-wise errors; sums of squares within 2e-7 relative of a float64 sum of
the same float32 squares (and within float64 rounding of the kernel's documented summation scheme)."""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import quantization_amd
    from quantization_amd import _lib
    _lib.lib()  # the native library must be there: no fallback
    assert torch.cuda.is_available()
    return quantization_amd


def to_dev(arr, dt):
    if dt == "f32":
        return torch.from_numpy(arr.copy()).cuda()
    t = torch.from_numpy(arr.view(np.int16).copy()).cuda()
    return t.view(torch.float16 if dt == "f16" else torch.bfloat16)


def bits16(t):
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


def fmt_args(orc, fmt):
    if fmt == "int8":
        return dict(fmt=orc.INT_SYM, bits=8)
    if fmt == "int8_asym":
        return dict(fmt=orc.INT_ASYM, bits=8)
    return dict(fmt=orc.LUT, bits=8, datatype=fmt)


def run_api(qa, w, fmt, gs):
    """The reference-shaped API -> dict of bit arrays like gen_golden.run_reference."""
    qw = qa.quant_weight
    out = {}
    if fmt.startswith("int"):
        asym = "asym" in fmt
        d, q, s, zp = qw._int_quant_meta(w, wq_bits=8, asym=asym, group_size=gs)
        d2 = (qw.quant_int_asym if asym else qw.quant_int)(w, wq_bits=8, group_size=gs)
        assert torch.equal(d.view(torch.int16), d2.view(torch.int16))
        out["q_is_int8"] = q.dtype == torch.int8
        out["q"] = q.contiguous().view(torch.uint8).cpu().numpy()
        if zp is not None:
            out["zp"] = bits16(zp.reshape(-1))
    else:
        d, q, s = qw._fp_quant_meta(w, wq_bits=8, datatype=fmt, group_size=gs)
        out["q"] = bits16(q)
        out["deq_quant_datatype"] = bits16(qw.quant_datatype(w, wq_bits=8, datatype=fmt, group_size=gs))
    assert d.dtype == torch.float16 and s.dtype == torch.float16 and d.shape == w.shape
    out["deq"] = bits16(d)
    out["scale"] = bits16(s.reshape(-1))
    out["shapes"] = (tuple(q.shape), tuple(s.shape))
    return out


def assert_same(got, exp, what):
    for k in ("deq", "q", "scale", "zp", "deq_quant_datatype"):
        if k in exp:
            assert k in got, f"{what}: missing {k}"
            assert got[k].shape == exp[k].shape, f"{what}: {k} shape {got[k].shape} != {exp[k].shape}"
            bad = np.flatnonzero(got[k].reshape(-1) != exp[k].reshape(-1))
            assert bad.size == 0, f"{what}: {k} differs at {bad.size} positions, first {bad[:5]}"
    if "q_is_int8" in exp:
        assert bool(exp["q_is_int8"]) == got["q_is_int8"], f"{what}: code dtype"


@pytest.mark.parametrize("fmt", gu.FORMATS)
@pytest.mark.parametrize("gs", gu.GROUPS, ids=gu.gs_key)
def test_golden_small(qa, fmt, gs):
    for key, dt, arr in gu.small_inputs():
        got = run_api(qa, to_dev(arr, dt), fmt, gs)
        assert_same(got, gu.small_expected(key, fmt, gs), f"{key}/{fmt}/{gs}")


@pytest.mark.parametrize("fmt", gu.FORMATS)
@pytest.mark.parametrize("gs", [None, 128], ids=gu.gs_key)
def test_golden_exhaustive(qa, fmt, gs):
    arr, exp = gu.exhaustive_expected(fmt, gs)
    assert_same(run_api(qa, to_dev(arr, "f16"), fmt, gs), exp, f"exhaustive/{fmt}/{gs}")


@pytest.mark.parametrize("fmt", gu.FORMATS)
@pytest.mark.parametrize("gs", [None, 64, -1], ids=gu.gs_key)
def test_golden_special_rows(qa, fmt, gs):
    """Rows with +-inf / NaN, rows on one side of zero, constant rows, float16 limits, as the reference leaves
    them (NaN == NaN; codes compared wherever the dequantized value is a number)."""
    arr, exp = gu.specials_expected(fmt, gs)
    got = run_api(qa, to_dev(arr, "f16"), fmt, gs)
    for k in ("deq", "scale", "zp", "deq_quant_datatype"):
        if k in exp:
            assert gu.same_f16_bits(got[k], exp[k]), f"{fmt}/{gs}: {k}"
    ok = (exp["deq"].reshape(-1) & 0x7fff) <= 0x7c00
    if fmt.startswith("int"):
        assert np.array_equal(got["q"].reshape(-1)[ok], exp["q"].reshape(-1)[ok]), f"{fmt}/{gs}: codes"
        assert bool(exp["q_is_int8"]) == got["q_is_int8"]
    else:
        assert gu.same_f16_bits(got["q"].reshape(-1)[ok], exp["q"].reshape(-1)[ok]), f"{fmt}/{gs}: codes"


@pytest.mark.parametrize("fmt", ["fp8_e4m3", "fp8_e3m4"])
def test_golden_lut_domain(qa, fmt):
    z = gu.npz("lut_domain.npz")
    w = to_dev(z[f"{fmt}/in"], "f16")
    d, q, s = qa.quant_weight._fp_quant_meta(w, wq_bits=8, datatype=fmt, group_size=None)
    assert torch.all(s == 1.0)
    assert np.array_equal(bits16(q), z[f"{fmt}/q"])
    assert np.array_equal(bits16(d), z[f"{fmt}/deq"])


# (rows, cols, group sizes): every kernel scope -- part-of-wave teams, wave per row, workgroup per row
# (256 and 1024 threads), the scalar fallback (odd lengths), and ragged tails
SHAPES = [
    (33, 256, [None, 8, 16, 32, 64, 128, 256, -1]),
    (7, 1024, [None, 512, 1024]),
    (5, 1536, [None, 96, 48, 768]),       # group sizes that are not powers of two
    (3, 4096, [None, 128]),
    (3, 8192, [None, 2048]),
    (2, 14336, [None, 128]),
    (2, 28672, [None, 64]),
    (1, 65536, [None, -1]),
    (2, 136000, [None]),                  # longer than a 1024-thread team holds -> fallback
    (6, 100, [None, 20, 10, -1]),         # no 8-element vectors -> fallback
    (1, 8, [None, -1]),
    (300, 40, [None, 8]),
]


def _rand(rows, cols, dt, seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(rows, cols, generator=g) * 0.02
    w[0, 0] = 0.7  # an outlier
    if rows > 2:
        w[2] = 0  # a zero row
    return {"f16": w.half(), "bf16": w.bfloat16(), "f32": w.float()}[dt]


def _np_in(wt, dt, orc):
    if dt == "f32":
        return wt.numpy(), orc.F32
    a = wt.view(torch.int16).numpy().view(np.uint16)
    return (a, orc.F16) if dt == "f16" else (a, orc.BF16)


@pytest.mark.parametrize("fmt", gu.FORMATS)
@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
def test_vs_oracle_shapes(qa, oracle, fmt, dt):
    from quantization_amd import _lib
    fmt_code = {"int8": _lib.INT_SYM, "int8_asym": _lib.INT_ASYM}.get(fmt, _lib.LUT_FP)
    for si, (rows, cols, groups) in enumerate(SHAPES):
        if dt != "f16" and cols > 30000:
            continue
        wt = _rand(rows, cols, dt, 100 + si)
        arr, code = _np_in(wt, dt, oracle)
        wd = wt.cuda()
        for gs in groups:
            exp = oracle.quantize(arr, code, group_size=gs, **fmt_args(oracle, fmt))
            res = qa.quantize(wd, fmt_code, 8, gs, fmt if fmt_code == _lib.LUT_FP else None,
                              abs_err=torch.float32, seg_sse=True, totals=True)
            what = f"{rows}x{cols} gs={gs} {fmt} {dt}"
            got = {"deq": bits16(res.deq), "scale": bits16(res.scale)}
            got["q"] = bits16(res.codes) if fmt_code == _lib.LUT_FP else res.codes.cpu().numpy()
            if res.zero_point is not None:
                got["zp"] = bits16(res.zero_point)
            assert_same(got, exp, what)
            # element-wise float32 error and its sums (onnx_error_cli.py:90-91, quant_error_multi_weight.py:55-58)
            v32 = exp["v16"].view(np.float16).astype(np.float32)
            d32 = exp["deq"].view(np.float16).astype(np.float32)
            err = v32 - d32
            assert np.array_equal(res.abs_err.cpu().numpy().view(np.uint32), np.abs(err).view(np.uint32)), what
            L = rows * cols if gs == -1 else (cols if (gs is None or gs <= 0) else gs)
            sq32, ab32 = err * err, np.abs(err)
            tot = res.totals.cpu().numpy()
            got_seg = res.seg_sse.cpu().numpy()
            # (a) against a float64 sum of the float32 squares: within the documented 2e-7
            np.testing.assert_allclose(got_seg, sq32.astype(np.float64).reshape(-1, L).sum(axis=1), rtol=2e-7,
                                       atol=0, err_msg=what)
            np.testing.assert_allclose(tot[0], sq32.astype(np.float64).sum(), rtol=2e-7, err_msg=what)
            np.testing.assert_allclose(tot[1], ab32.astype(np.float64).sum(), rtol=2e-7, err_msg=what)
            # (b) against the kernel's own summation scheme (a lane's 8 values pairwise in float32, float64
            #     above; the scalar fallback sums every element in float64): equal up to float64 rounding
            if L % 8 == 0 and L // 8 <= 16384:
                def lane_sums(a):
                    a = a.reshape(-1, 8)
                    return (((a[:, 0] + a[:, 1]) + (a[:, 2] + a[:, 3])) +
                            ((a[:, 4] + a[:, 5]) + (a[:, 6] + a[:, 7]))).astype(np.float64)
                psq, pab = lane_sums(sq32), lane_sums(ab32)
            else:
                psq, pab = sq32.astype(np.float64).reshape(-1), ab32.astype(np.float64).reshape(-1)
            np.testing.assert_allclose(got_seg, psq.reshape(got_seg.size, -1).sum(axis=1), rtol=1e-12, atol=0,
                                       err_msg=what)
            np.testing.assert_allclose(tot[0], psq.sum(), rtol=1e-12, err_msg=what)
            np.testing.assert_allclose(tot[1], pab.sum(), rtol=1e-12, err_msg=what)
            # float16 error variant (single_weight_quant_error.py:39-40)
            res16 = qa.quantize(wd, fmt_code, 8, gs, fmt if fmt_code == _lib.LUT_FP else None,
                                deq=False, codes=False, scale=False, abs_err=torch.float16)
            e16 = np.abs(exp["v16"].view(np.float16) - exp["deq"].view(np.float16))
            assert np.array_equal(bits16(res16.abs_err), e16.view(np.uint16)), what


def test_return_shapes_and_dtypes(qa):
    qw = qa.quant_weight
    w = (torch.randn(16, 256) * 0.02).half().cuda()
    d, q, s, zp = qw._int_quant_meta(w, 8, False, 64)
    assert q.shape == (16, 4, 64) and q.dtype == torch.int8 and s.shape == (16, 4) and zp is None
    d, q, s, zp = qw._int_quant_meta(w, 8, True, 64)
    assert q.shape == (16, 4, 64) and q.dtype == torch.uint8 and zp.shape == (16, 4) and zp.dtype == torch.float16
    d, q, s, zp = qw._int_quant_meta(w, 8, True, None)
    assert q.shape == (16, 256) and s.shape == (16,) and zp.shape == (16,)
    d, q, s, zp = qw._int_quant_meta(w, 8, False, -1)
    assert q.dtype == torch.uint8 and s.shape == () and zp is None
    d, q, s, zp = qw._int_quant_meta(w, 8, True, -1)
    assert q.dtype == torch.int8 and s.shape == () and zp.shape == ()
    d, q, s = qw._fp_quant_meta(w, 8, "fp8_e4m3", 32)
    assert d.shape == (16, 256) and q.shape == (16, 8, 32) and q.dtype == torch.float16 and s.shape == (16, 8)
    # a 3-D tensor with group_size=None reduces over the last axis (search_datatype's call shape, :640)
    w3 = w.reshape(16, 4, 64)
    d3 = qw.quant_datatype(w3, 8, "fp8_e3m4", None)
    d2 = qw.quant_datatype(w, 8, "fp8_e3m4", 64)
    assert d3.shape == (16, 4, 64) and torch.equal(d3.reshape(16, 256).view(torch.int16), d2.view(torch.int16))
    # host tensors go through the device and come back on the host
    dc = qw.quant_int(w.cpu(), 8, 64)
    assert dc.device.type == "cpu" and torch.equal(dc.view(torch.int16), qw.quant_int(w, 8, 64).cpu().view(torch.int16))


def test_errors_match_reference_types(qa):
    qw = qa.quant_weight
    w = (torch.randn(4, 100) * 0.02).half().cuda()
    with pytest.raises(RuntimeError):
        qw.quant_int(w, 8, 64)            # 100 % 64 != 0: torch reshape error in the reference
    with pytest.raises(ValueError):
        qw.quant_datatype(w, 7, "fp8_e4m3", None)   # unsupported bit width
    with pytest.raises(AssertionError):
        qw.quant_datatype(w, 8, "fp8_bogus", None)  # unknown table name
    with pytest.raises(ValueError):
        qw.quant_int(w.reshape(4, 2, 50), 8, 10)    # per-group needs a 2-D weight
    assert qw.quant_int(torch.empty(0, 64, dtype=torch.float16, device="cuda"), 8, None).shape == (0, 64)


@pytest.mark.parametrize("bits", [2, 3, 4, 5, 6, 7])
def test_low_bit_int(qa, oracle, bits):
    wt = _rand(16, 512, "f16", 7)
    arr, code = _np_in(wt, "f16", oracle)
    for asym in (False, True):
        for gs in (None, 128):
            exp = oracle.quantize(arr, code, oracle.INT_ASYM if asym else oracle.INT_SYM, bits, gs)
            d, q, s, zp = qa.quant_weight._int_quant_meta(wt.cuda(), bits, asym, gs)
            assert np.array_equal(bits16(d), exp["deq"]) and np.array_equal(bits16(s.reshape(-1)), exp["scale"])
            assert np.array_equal(q.contiguous().view(torch.uint8).cpu().numpy().reshape(16, 512), exp["q"])


def test_quant_model_walker(qa):
    import torch.nn as nn
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(256, 128), nn.ReLU(), nn.Linear(128, 64, bias=False)).half().cuda()
    w0 = model[0].weight.data.clone()
    qa.quant_weight.quant_model(model, 8, "int8_asym", 64)
    assert type(model[0]).__name__ == "_LinearDump" and model[0].weight_q.dtype == torch.float16
    assert model[0].weight_zp.shape == (128, 4) and model[2].weight_scale.shape == (64, 2)
    assert torch.equal(model[0].weight.data.view(torch.int16),
                       qa.quant_weight.quant_int_asym(w0, 8, 64).view(torch.int16))
    y = model(torch.randn(2, 256, device="cuda", dtype=torch.float16))
    assert y.shape == (2, 64)


def test_batched_launch_equals_single(qa):
    """quantize_batch (one launch, blockIdx.y = tensor) against per-tensor quantize: identical bits."""
    from quantization_amd import _lib
    from quantization_amd.fused import quantize_batch
    torch.manual_seed(3)
    for (B, rows, cols, gs) in [(5, 64, 512, 64), (3, 16, 2048, None), (2, 8, 16384, None), (4, 32, 256, -1), (35, 8, 128, 32)]:
        w = (torch.randn(B, rows, cols, device="cuda") * 0.02).half()
        for fmt, dt in ((_lib.INT_SYM, None), (_lib.INT_ASYM, None), (_lib.LUT_FP, "fp8_e3m4")):
            rb = quantize_batch(w, fmt, 8, gs, dt, abs_err=torch.float32, seg_sse=True, totals=True)
            for i in range(B):
                r1 = qa.quantize(w[i], fmt, 8, gs, dt, abs_err=torch.float32, seg_sse=True, totals=True)
                assert torch.equal(rb.deq[i].view(torch.int16), r1.deq.view(torch.int16))
                assert torch.equal(rb.codes[i].view(torch.uint8), r1.codes.view(torch.uint8))
                assert torch.equal(rb.scale[i].view(torch.int16), r1.scale.view(torch.int16))
                if r1.zero_point is not None:
                    assert torch.equal(rb.zero_point[i].view(torch.int16), r1.zero_point.view(torch.int16))
                assert torch.equal(rb.abs_err[i], r1.abs_err)
                assert torch.equal(rb.seg_sse[i], r1.seg_sse) and torch.equal(rb.totals[i], r1.totals)


def test_out_of_domain_arguments_behave_as_the_reference(qa):
    """tests/golden/api_errors.json (gen_golden.py api): what the imported reference raises -- type and message --
    or returns -- dtype and shape -- for bit widths, datatypes, group sizes and tensor ranks outside its domain."""
    import json
    import os
    import torch
    qw = qa.quant_weight
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "api_errors.json")) as f:
        cases = json.load(f)
    assert len(cases) >= 20
    for c in cases:
        torch.manual_seed(1)
        w = (torch.randn(*c["shape"]) * 0.02).half().cuda()
        fn, what = getattr(qw, c["fn"]), (c["fn"], c["kwargs"], c["shape"])
        if "raises" in c:
            exc = {"RuntimeError": RuntimeError, "ValueError": ValueError, "AssertionError": AssertionError}[c["raises"][0]]
            with pytest.raises(exc) as info:
                fn(w, **c["kwargs"])
            assert type(info.value) is exc, what
            assert str(info.value) == c["raises"][1], what
        else:
            r = fn(w, **c["kwargs"])
            r0 = r[0] if isinstance(r, tuple) else r
            assert str(r0.dtype) == c["ok"]["dtype"] and list(r0.shape) == c["ok"]["shape"], what


def test_fused_p99_matches_numpy_on_the_dump():
    """quantize_batch(p99=): the percentile selected inside the quantize pass equals np.percentile of the
    float32 |error| dump bit for bit, on shapes that take the fed route (team / wave / workgroup kernels,
    >= 2^21 elements), on shapes that fall back to selecting from the dump, and all other outputs are
    unchanged by it."""
    import numpy as np
    import torch
    from quantization_amd import _lib
    from quantization_amd.fused import quantize_batch
    g = torch.Generator(device="cuda").manual_seed(5)
    cases = [((2, 2048, 2048), _lib.INT_ASYM, None, 32), ((3, 1024, 4096), _lib.INT_SYM, None, 128),
             ((2, 2048, 2048), _lib.LUT_FP, "fp8_e4m3", None), ((1, 256, 16384), _lib.INT_ASYM, None, None),
             ((2, 2048, 1536), _lib.INT_ASYM, None, 96), ((2, 64, 1024), _lib.INT_SYM, None, 64),
             ((1, 1024, 4096), _lib.INT_ASYM, None, -1)]
    for shape, fmt, dt, gs in cases:
        w = (torch.randn(*shape, generator=g, device="cuda") * 0.02).half()
        for pct in (99, 50, 99.9):
            a = quantize_batch(w, fmt, 8, gs, dt, abs_err=torch.float32, seg_sse=True, totals=True, p99=pct)
            b = quantize_batch(w, fmt, 8, gs, dt, abs_err=torch.float32, seg_sse=True, totals=True)
            for f in ("deq", "codes", "scale", "zero_point", "abs_err", "seg_sse", "totals"):
                x, y = getattr(a, f), getattr(b, f)
                assert (x is None) == (y is None)
                if x is not None:
                    assert torch.equal(x.view(torch.uint8), y.view(torch.uint8)), (shape, gs, f)
            err = a.abs_err.cpu().numpy()
            want = np.array([np.percentile(err[i].ravel(), pct) for i in range(shape[0])], np.float32)
            assert a.abs_p99.cpu().numpy().tobytes() == want.tobytes(), (shape, fmt, gs, pct)


def test_fused_p99_survives_a_misleading_sample():
    """The sampling pass of the team kernels looks at one vector slot -- (64 / team) segments -- of one tile
    (4 slots) out of every 8 tiles: h = top 5 bits of t * 0x9E3779B1, tile t * 8 + (h & 7), slot h >> 3.  Zero
    weights in exactly those slots make the sample say 'all errors are 0'; the result must still be numpy's."""
    import numpy as np
    import torch
    from quantization_amd import _lib
    from quantization_amd.fused import quantize_batch
    g = torch.Generator(device="cuda").manual_seed(6)
    rows, cols, gs = 2048, 2048, 64          # team of 8 lanes: 8 segments per slot, 32 per tile
    w = (torch.randn(1, rows, cols, generator=g, device="cuda") * 0.02).half()
    seg_per_slot = 64 // (gs // 8)
    nseg = rows * cols // gs
    tiles = nseg // (seg_per_slot * 4)
    t = np.arange(tiles // 8, dtype=np.uint64)
    h = ((t * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) >> np.uint64(27)
    slot = (t * np.uint64(8) + (h & np.uint64(7))) * np.uint64(4) + (h >> np.uint64(3))
    flat = w.view(-1, seg_per_slot * gs)
    flat[torch.from_numpy(slot.astype(np.int64)).cuda()] = 0
    a = quantize_batch(w, _lib.INT_ASYM, 8, gs, None, abs_err=torch.float32, p99=99)
    want = np.float32(np.percentile(a.abs_err.cpu().numpy().ravel(), 99))
    assert a.abs_p99.cpu().numpy().tobytes() == np.array([want], np.float32).tobytes()
    assert want > 0


@pytest.mark.parametrize("bits,fmt", gu.MORE_TABLES)
def test_golden_more_table_formats(qa, bits, fmt):
    """Value tables with subnormals (fp4, fp5, fp6), fp8_e2m5 and the INTn lists as tables against the imported
    reference: small inputs, every float16 pattern, the rounding-domain sweep; quant_datatype incl. its
    float32 per-tensor branch."""
    qw = qa.quant_weight
    for name, arr, gs, exp in gu.more_cases(fmt):
        w = to_dev(arr, "f16")
        d, q, s = qw._fp_quant_meta(w, wq_bits=bits, datatype=fmt, group_size=gs)
        assert np.array_equal(bits16(d).reshape(-1), exp["deq"].reshape(-1)), (name, gs, "deq")
        assert np.array_equal(bits16(q).reshape(-1), exp["q"].reshape(-1)), (name, gs, "q")
        assert np.array_equal(bits16(s.reshape(-1)), exp["scale"].reshape(-1)), (name, gs, "scale")
        d2 = qw.quant_datatype(w, wq_bits=bits, datatype=fmt, group_size=gs)
        assert np.array_equal(bits16(d2).reshape(-1), exp["deq_quant_datatype"].reshape(-1)), (name, gs, "qd")
    z = gu.npz("quant_more_formats.npz")
    w = to_dev(z[f"domain/{fmt}/in"], "f16")
    d, q, s = qw._fp_quant_meta(w, wq_bits=bits, datatype=fmt, group_size=None)
    assert torch.all(s == 1.0)
    assert np.array_equal(bits16(q), z[f"domain/{fmt}/q"])


@pytest.mark.parametrize("asym", [False, True])
def test_golden_four_bit_int(qa, asym):
    for name, arr, gs, exp in gu.more_cases("int4_asym.int" if asym else "int4.int"):
        w = to_dev(arr, "f16")
        d, q, s, zp = qa.quant_weight._int_quant_meta(w, wq_bits=4, asym=asym, group_size=gs)
        assert np.array_equal(bits16(d).reshape(-1), exp["deq"].reshape(-1)), (name, gs)
        assert np.array_equal(q.contiguous().view(torch.uint8).cpu().numpy().reshape(-1), exp["q"].reshape(-1)), (name, gs)
        assert np.array_equal(bits16(s.reshape(-1)), exp["scale"].reshape(-1)), (name, gs)
        if asym:
            assert np.array_equal(bits16(zp.reshape(-1)), exp["zp"].reshape(-1)), (name, gs)


def test_quant_model_batched_equals_per_layer(qa):
    """quant_model groups the Linear layers by shape and runs one fused launch per group; weights and buffers
    must equal what the per-layer functions give (codes come from the already dequantized weight, as in the
    reference's _LinearDump)."""
    import torch.nn as nn
    qw = qa.quant_weight
    for datatype, bits, gs in (("int8", 8, None), ("int4_asym", 4, 64), ("fp8_e4m3", 8, 128), ("fp4", 4, -1),
                               ("int8_asym", 8, -1)):
        torch.manual_seed(1)
        layers = [nn.Linear(256, 128), nn.Linear(128, 256), nn.Linear(256, 128, bias=False), nn.Linear(256, 128),
                  nn.Linear(128, 256)]
        model = nn.Sequential(*layers).half().cuda()
        orig = [m.weight.data.clone() for m in model]
        qw.quant_model(model, bits, datatype, gs)
        for i, m in enumerate(model):
            if datatype.startswith("int"):
                asym = "asym" in datatype
                d = (qw.quant_int_asym if asym else qw.quant_int)(orig[i], bits, gs)
                _, q, s, zp = qw._int_quant_meta(d, bits, asym, gs)
            else:
                d = qw.quant_datatype(orig[i], bits, datatype, gs)
                _, q, s = qw._fp_quant_meta(d, bits, datatype, gs)
                zp = None
            assert torch.equal(m.weight.data.view(torch.int16), d.view(torch.int16)), (datatype, i, "weight")
            assert torch.equal(m.weight_q, q.to(torch.float16)) and m.weight_q.shape == q.shape, (datatype, i, "q")
            assert torch.equal(m.weight_scale.view(torch.int16), s.to(torch.float16).view(torch.int16)), (datatype, i)
            if zp is not None:
                assert torch.equal(m.weight_zp.view(torch.int16), zp.view(torch.int16)), (datatype, i, "zp")
            else:
                assert not hasattr(m, "weight_zp")


@pytest.mark.parametrize("fmt", ["int8", "int8_asym", "fp8_e4m3", "fp4"])
def test_statistics_only_mode_equals_general_mode(qa, fmt):
    """Without any element-sized output the kernels compiled without the stores run (the model sweeps): scales
    and zero-points must have the bits the general kernels produce; per-segment and per-tensor sums agree to the
    1e-6 bar (this mode adds a lane's 8 exact float32 squares with v_dot2_f32_f16, the general kernels pairwise;
    everything above a lane is float64 in both)."""
    from quantization_amd import _lib
    from quantization_amd.fused import quantize, quantize_batch
    code = {"int8": _lib.INT_SYM, "int8_asym": _lib.INT_ASYM}.get(fmt, _lib.LUT_FP)
    table = fmt if code == _lib.LUT_FP else None
    bits = 4 if fmt == "fp4" else 8
    torch.manual_seed(9)
    for (rows, cols, gs, dt) in [(64, 1024, 32, torch.float16), (48, 1024, 64, torch.bfloat16), (40, 2048, 128, torch.float16),
                                 (33, 2048, None, torch.float16), (9, 5632, None, torch.bfloat16), (5, 28672, None, torch.float16),
                                 (3, 70000, None, torch.float32), (16, 512, 512, torch.float16), (7, 96, 24, torch.float16), (11, 768, 256, torch.bfloat16),
                                 (4, 4096, -1, torch.float16)]:
        w = (torch.randn(rows, cols) * 0.02).to(dt).cuda()
        w[0, 3] = 0.5
        if rows > 2:
            w[2] = 0
        full = quantize(w, code, bits, gs, table, deq=True, codes=True, scale=True, seg_sse=True, totals=True)
        stats = quantize(w, code, bits, gs, table, deq=False, codes=False, scale=True, seg_sse=True, totals=True)
        what = (fmt, rows, cols, gs, dt)
        assert torch.equal(full.scale.view(torch.int16), stats.scale.view(torch.int16)), what
        if code == _lib.INT_ASYM:
            assert torch.equal(full.zero_point.view(torch.int16), stats.zero_point.view(torch.int16)), what
        # (same float32 squares, added in another order: at most 4 float32 roundings per lane)
        torch.testing.assert_close(stats.seg_sse, full.seg_sse, rtol=1e-6, atol=0)
        torch.testing.assert_close(stats.totals, full.totals, rtol=5e-7, atol=0)
        # and the error sums belong to the dequantized values the general kernel returned
        err = (w.to(torch.float16).float() - full.deq.float())
        want = (err * err).double().sum().item()
        assert abs(stats.totals[0].item() - want) <= 1e-6 * want, what
    w3 = (torch.randn(5, 256, 1024) * 0.02).half().cuda()
    b_full = quantize_batch(w3, code, bits, 128, table, deq=True, codes=False, scale=True, seg_sse=True, totals=True)
    b_stats = quantize_batch(w3, code, bits, 128, table, deq=False, codes=False, scale=False, seg_sse=True, totals=True)
    torch.testing.assert_close(b_stats.seg_sse, b_full.seg_sse, rtol=1e-6, atol=0)
    torch.testing.assert_close(b_stats.totals, b_full.totals, rtol=5e-7, atol=0)


QUANT_MODEL_CASES = [("int8", 8), ("int8_asym", 8), ("int4_asym", 4), ("fp8_e4m3", 8), ("fp4", 4)]


@pytest.mark.parametrize("case", QUANT_MODEL_CASES, ids=[c[0] for c in QUANT_MODEL_CASES])
def test_quant_model_reference_fixture(qa, case):
    """quant_model (weights rewritten in place, every nn.Linear replaced by a _LinearDump carrying weight_q /
    weight_scale / weight_zp) against what the reference's own quant_model produced for the same seeded model
    (tests/golden/quant_model.npz, written by gen_golden.py from the imported quant_weight.py:652-706, :876-936):
    module class names, weights, and every buffer's name, dtype, shape and bytes."""
    import contextlib
    import io
    import torch.nn as nn
    datatype, bits = case
    z = gu.npz("quant_model.npz")
    for gs in (None, 64, -1):
        layers = []
        for i, (fin, fout) in enumerate(((256, 128), (128, 256), (256, 128), (128, 64))):
            m = nn.Linear(fin, fout, bias=bool(z[f"in/{i}/has_bias"][0])).half()
            m.weight.data = torch.from_numpy(z[f"in/{i}/weight"].view(np.int16).copy()).view(torch.float16)
            layers.append(m)
        model = nn.Sequential(*layers).cuda()
        with contextlib.redirect_stdout(io.StringIO()):
            qa.quant_weight.quant_model(model, bits, datatype, gs)
        key = f"{datatype}/{gu.gs_key(gs)}"
        assert [type(m).__name__ for m in model] == list(z[f"{key}/classes"]), key
        for i, m in enumerate(model):
            what = (key, i)
            assert isinstance(m, nn.Linear)
            assert np.array_equal(bits16(m.weight.data), z[f"{key}/{i}/weight"]), (what, "weight")
            assert sorted(n for n, _ in m.named_buffers()) == list(z[f"{key}/{i}/buffers"]), what
            for n in z[f"{key}/{i}/buffers"]:
                b = getattr(m, str(n))
                assert str(b.dtype) == z[f"{key}/{i}/{n}/dtype"][0], (what, n, "dtype")
                assert list(b.shape) == list(z[f"{key}/{i}/{n}/shape"]), (what, n, "shape")
                got = b.contiguous().reshape(-1).view(torch.uint8).cpu().numpy()
                assert np.array_equal(got, z[f"{key}/{i}/{n}/bytes"]), (what, n, "bytes")


_SLOW_FEED_SCRIPT = r"""
import sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
from quantization_amd import _lib
from quantization_amd.fused import quantize_batch
g = torch.Generator(device="cuda").manual_seed(12)
rows, cols, gs = 2048, 4096, 32
x = torch.randn(1, rows, cols, generator=g, device="cuda")
chi = sum(torch.randn(1, rows, cols, generator=g, device="cuda") ** 2 for _ in range(3))
w = (x / torch.sqrt(chi / 3) * 0.02).half()
# one group in 16 lies on one side of zero with a range below half its offset: |v / s| >= 500, the per-element path;
# its errors have the size of every other group's, so they fall inside the percentile's bracket
grp = w.view(-1, gs)
pick = torch.rand(grp.shape[0], generator=g, device="cuda") < 1 / 16
pick[0::16] = False                                          # waves whose first lanes are on the fast path ...
pick[5::64] = True                                           # ... while other lanes of theirs are not
grp[pick] = grp[pick].abs().clamp(max=0.12) + 0.3
for pct in (99, 99.9):
    a = quantize_batch(w, _lib.INT_ASYM, 8, gs, None, abs_err=torch.float32, p99=pct)
    err = a.abs_err.cpu().numpy().ravel()
    want = np.float32(np.percentile(err, pct))
    got = a.abs_p99.cpu().numpy()
    assert got.tobytes() == np.array([want], np.float32).tobytes(), (pct, got, want)
print("ok")
"""


def test_fused_p99_wide_bracket_with_per_element_segments():
    """Heavy-tailed weights (a wide bracket: the values inside it are kept as a list, staged per wave in LDS) with
    segments on the per-element path scattered among the others, so that the stage is flushed by waves whose first
    lanes are not taking part.  Eight workgroups (MQ_MAX_BLOCKS, read once per process: hence a child process) make
    every wave stage more than its area holds.  The P99 must equal np.percentile of the dump."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _SLOW_FEED_SCRIPT, root], env=dict(os.environ, MQ_MAX_BLOCKS="8"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def _zero_sign_inputs(rows, cols, seed):
    """Non-negative rows with zeros of both signs in (nearly) every group of 4 elements; some rows with negatives."""
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(rows, cols, generator=g).abs() * 0.02 + 0.001
    zs = torch.rand(rows, cols, generator=g) < 0.35
    neg = torch.rand(rows, cols, generator=g) < 0.5
    w[zs & neg] = -0.0
    w[zs & ~neg] = 0.0
    if rows > 3:
        w[rows - 1, ::7] = -0.01          # the zeros are not the minimum there
        w[rows - 2] = -w[rows - 2]        # all <= 0: the zeros are the maximum
    return w


def _bits16(t):
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


def test_zero_point_sign_of_zero_golden():
    """The imported reference's outputs on segments whose minimum is a zero and that hold +0 and -0
    (tests/golden/quant_zero_sign.npz): below 64 elements torch.amin returns the first zero in memory and the
    zero-point's zero carries the opposite sign -- compared bit for bit, through the general kernels (every output)
    and through the statistics-only kernels; from 64 elements on the sign of a zero zero-point is left out (the
    reference's choice there depends on the vector lanes of the CPU that ran it)."""
    from quantization_amd import _lib
    from quantization_amd.fused import quantize
    z = gu.npz("quant_zero_sign.npz")
    w = z["in"]
    wt = torch.from_numpy(w.view(np.int16).copy()).view(torch.float16).cuda()
    for bits in (8, 4):
        for gs in (8, 16, 32, 64, 128, None, -1):
            pre = f"out/int{bits}_asym/{gu.gs_key(gs)}/"
            L = w.size if gs == -1 else (gs or w.shape[1])
            full = quantize(wt, _lib.INT_ASYM, bits, gs, None, deq=True, codes=True, scale=True)
            stat = quantize(wt, _lib.INT_ASYM, bits, gs, None, deq=False, codes=False, scale=True, totals=True)
            assert np.array_equal(_bits16(full.deq).reshape(-1), z[pre + "deq"].reshape(-1)), (bits, gs)
            assert np.array_equal(full.codes.cpu().numpy().reshape(-1), z[pre + "q"].reshape(-1)), (bits, gs)
            for res in (full, stat):
                assert np.array_equal(_bits16(res.scale), z[pre + "scale"]), (bits, gs)
                got, want = _bits16(res.zero_point).copy(), z[pre + "zp"].copy()
                if L >= 64:
                    got[got == 0x8000] = 0
                    want[want == 0x8000] = 0
                assert np.array_equal(got, want), (bits, gs, int((got != want).sum()))


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
def test_zero_point_sign_of_zero_every_kernel(oracle, dt):
    """The same rule against the oracle (which the CPU suite pins to the reference's fixture), bit for bit at EVERY
    segment length -- first zero in memory below 64 elements, -0 below +0 from there on -- through each kernel that
    stores zero-points: sub-wave teams (8 .. 56 elements), the generic kernel (lengths that are not a multiple of 8,
    unaligned rows), the four-vectors-per-lane statistics kernels (32 elements; float32 weights take the other one),
    rows, per-tensor."""
    from quantization_amd import _lib
    from quantization_amd.fused import quantize
    odt = {"f16": oracle.F16, "bf16": oracle.BF16, "f32": oracle.F32}[dt]
    cases = [(24, 1024, 8), (24, 1024, 16), (24, 960, 24), (24, 1024, 32), (24, 960, 40), (24, 960, 48), (24, 1008, 56),
             (24, 1024, 64), (24, 1024, 128), (16, 1000, 4), (16, 1000, 20), (16, 1000, 50), (16, 999, 27), (16, 999, 37),
             (24, 2048, None), (8, 333, None), (5, 40, None), (8, 512, -1), (37, 32, 32), (3, 24, None)]
    for k, (rows, cols, gs) in enumerate(cases):
        w = _zero_sign_inputs(rows, cols, 500 + k)
        wt = {"f16": w.half(), "bf16": w.bfloat16(), "f32": w}[dt]
        raw = wt.numpy() if dt == "f32" else wt.view(torch.int16).numpy().view(np.uint16)
        for bits in (8, 3):
            exp = oracle.quantize(raw, odt, group_size=gs, fmt=oracle.INT_ASYM, bits=bits)
            calls = [dict(deq=True, codes=True, scale=True), dict(deq=False, codes=False, scale=True, totals=True),
                     dict(deq=False, codes=False, scale=True, seg_sse=True)]
            for kw in calls:
                res = quantize(wt.cuda(), _lib.INT_ASYM, bits, gs, None, **kw)
                got = _bits16(res.zero_point)
                assert np.array_equal(got, exp["zp"]), (dt, rows, cols, gs, bits, sorted(kw), int((got != exp["zp"]).sum()))
                assert np.array_equal(_bits16(res.scale), exp["scale"])
                if kw.get("deq"):
                    assert np.array_equal(_bits16(res.deq).reshape(-1), exp["deq"].reshape(-1))


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_zero_point_sign_of_zero_numpy_order_launch(oracle, dt):
    """... and through the numpy-order sweep launch (32-element groups: a quad of lanes shares the group's words)."""
    import ctypes
    from quantization_amd import _lib
    from quantization_amd.fused import _DTYPES, _QuantItem, bind_batch, make_cfg, workspace
    L = bind_batch()
    shapes = [(16, 1024), (8, 2048), (4, 8192)]
    ws_ = [_zero_sign_inputs(r, c, 600 + i) for i, (r, c) in enumerate(shapes)]
    wts = [(w.half() if dt == "f16" else w.bfloat16()).cuda() for w in ws_]
    cfg = make_cfg(_lib.INT_ASYM, 8, 32, None)
    n = len(wts)
    items = (_QuantItem * n)()
    keep = []
    for j, wt in enumerate(wts):
        nseg = wt.numel() // 32
        ent = torch.empty(nseg, dtype=torch.float32, device="cuda")
        sc = torch.empty(nseg, dtype=torch.float16, device="cuda")
        zp = torch.empty(nseg, dtype=torch.float16, device="cuda")
        tot = torch.zeros(2, dtype=torch.float64, device="cuda")
        keep.append((ent, sc, zp, tot))
        items[j].w = wt.data_ptr()
        o = items[j].out
        o.abs_err_dtype = _lib.F32
        o.seg_mse_f32 = 2
        o.seg_sse, o.scale, o.zero_point, o.totals = ent.data_ptr(), sc.data_ptr(), zp.data_ptr(), tot.data_ptr()
    rows = (ctypes.c_int64 * n)(*[s[0] for s in shapes])
    cols = (ctypes.c_int64 * n)(*[s[1] for s in shapes])
    leaf = torch.empty(sum(w.numel() // 128 for w in wts), dtype=torch.float32, device="cuda")
    host = torch.empty(L.mq_sweep_numpy_table_bytes(n), dtype=torch.uint8)
    seg_len, blocks, tl, lc = (ctypes.c_int64(0) for _ in range(4))
    _lib.check(L.mq_sweep_numpy_prepare(items, rows, cols, n, _DTYPES[wts[0].dtype], ctypes.byref(cfg), leaf.data_ptr(),
                                        host.data_ptr(), host.numel(), ctypes.byref(seg_len), ctypes.byref(blocks),
                                        ctypes.byref(tl), ctypes.byref(lc)))
    tab = host.cuda()
    ws, stream = workspace(torch.device("cuda:0"), L.mq_sweep_workspace_bytes(blocks.value))
    _lib.check(L.mq_sweep_numpy_run(tab.data_ptr(), n, seg_len.value, blocks.value, _DTYPES[wts[0].dtype], ctypes.byref(cfg),
                                    ws.data_ptr(), ws.numel(), stream.cuda_stream))
    torch.cuda.synchronize()
    odt = oracle.F16 if dt == "f16" else oracle.BF16
    for wt, (ent, sc, zp, tot) in zip(wts, keep):
        raw = wt.cpu().view(torch.int16).numpy().view(np.uint16)
        exp = oracle.quantize(raw, odt, group_size=32, fmt=oracle.INT_ASYM, bits=8)
        assert (exp["zp"] == 0x8000).sum() > 10 and (exp["zp"] == 0).sum() > 10
        assert np.array_equal(_bits16(zp), exp["zp"]), int((_bits16(zp) != exp["zp"]).sum())
        assert np.array_equal(_bits16(sc), exp["scale"])


def test_passes_enqueued_together_equal_separate_calls():
    """quantize_batch_passes (mq_quantize_p99_passes: the passes of a sweep enqueued together, their selection chains
    on the library's two streams beside the neighbouring full passes) against one quantize_batch(p99=) call per pass
    plus percentile_rows over its lists: every output bit for bit.  Shapes on the fed route (groups of a large
    tensor), on the fallback (small tensors, per-channel rows: selection from the stored errors) and a stack of more
    than 32 tensors (two chunks per pass); twice, the second time reusing the results' buffers."""
    from quantization_amd import _lib
    from quantization_amd.error_metrics import percentile_rows
    from quantization_amd.fused import quantize_batch, quantize_batch_passes
    g = torch.Generator().manual_seed(41)
    outs = dict(deq=True, codes=True, scale=True, seg_mse=True, totals=True)
    cases = [((6, 1024, 2048), [(_lib.INT_ASYM, 8, 32, None), (_lib.INT_ASYM, 8, 64, None), (_lib.INT_ASYM, 8, 128, None),
                                (_lib.LUT_FP, 8, 64, "fp8_e4m3"), (_lib.INT_SYM, 4, None, None)]),
             ((3, 64, 256), [(_lib.INT_ASYM, 8, 32, None), (_lib.INT_SYM, 8, None, None), (_lib.INT_SYM, 8, -1, None)]),
             ((40, 128, 512), [(_lib.INT_SYM, 8, 32, None), (_lib.LUT_FP, 4, 128, "fp4")])]
    for shape, passes in cases:
        w = (torch.randn(*shape, generator=g) * 0.02).half().cuda()
        want = []
        for fmt, bits, gs, dt in passes:
            r = quantize_batch(w, fmt, bits, gs, dt, abs_err=torch.float32, p99=99, **outs)
            lp = percentile_rows([r.seg_mse], 97, wide=True)
            want.append((r, lp))
        torch.cuda.synchronize()
        got = None
        for rep in range(2):
            got = quantize_batch_passes(w, [dict(fmt=f, bits=b, group_size=gs, datatype=dt, **outs) for f, b, gs, dt in passes],
                                        p99=99, list_p99=97, out=got)
            torch.cuda.synchronize()
            for (r, lp), q, ps in zip(want, got, passes):
                for name in ("deq", "codes", "scale", "zero_point", "abs_err", "seg_mse", "totals", "abs_p99"):
                    a, b = getattr(r, name), getattr(q, name)
                    assert (a is None) == (b is None), (shape, ps, name)
                    if a is not None:
                        assert torch.equal(a.view(torch.uint8) if a.dtype != torch.uint8 else a,
                                           b.view(torch.uint8) if b.dtype != torch.uint8 else b), (shape, ps, name, rep)
                assert torch.equal(lp.view(torch.int64), q.list_p99.view(torch.int64)), (shape, ps, "list_p99", rep)
    with pytest.raises(ValueError):
        quantize_batch_passes(w, [dict(fmt=_lib.INT_SYM, seg_mse=False)], list_p99=99)
