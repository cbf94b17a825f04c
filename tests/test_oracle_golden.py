"""CPU: the C / numpy oracle against the golden vectors generated from the imported reference."""
import hashlib

import numpy as np
import pytest

import golden_util as gu


def _fmt_args(orc, fmt):
    if fmt == "int8":
        return dict(fmt=orc.INT_SYM, bits=8)
    if fmt == "int8_asym":
        return dict(fmt=orc.INT_ASYM, bits=8)
    return dict(fmt=orc.LUT, bits=8, datatype=fmt)


def _dtype_code(orc, name):
    return {"f16": orc.F16, "bf16": orc.BF16, "f32": orc.F32}[name]


def _check(orc, w, dt, fmt, gs, exp):
    K, C = w.shape
    out = orc.quantize(w, _dtype_code(orc, dt), group_size=gs, **_fmt_args(orc, fmt))
    assert np.array_equal(out["deq"], exp["deq"].reshape(K, C)), "deq bits"
    assert np.array_equal(out["q"], exp["q"].reshape(K, C)), "codes"
    assert np.array_equal(out["scale"], exp["scale"].reshape(-1)), "scale bits"
    if "zp" in exp:
        assert np.array_equal(out["zp"], exp["zp"].reshape(-1)), "zero-point bits"
    if "deq_quant_datatype" in exp:
        out2 = orc.quantize(w, _dtype_code(orc, dt), group_size=gs, lut_f32_path=(gs == -1),
                            want=("deq",), **_fmt_args(orc, fmt))
        assert np.array_equal(out2["deq"], exp["deq_quant_datatype"].reshape(K, C)), "quant_datatype deq"


@pytest.mark.parametrize("fmt", gu.FORMATS)
@pytest.mark.parametrize("gs", gu.GROUPS, ids=gu.gs_key)
def test_small_vectors(oracle, fmt, gs):
    for key, dt, w in gu.small_inputs():
        _check(oracle, w, dt, fmt, gs, gu.small_expected(key, fmt, gs))


@pytest.mark.parametrize("fmt", gu.FORMATS)
@pytest.mark.parametrize("gs", [None, 128], ids=gu.gs_key)
def test_exhaustive_f16_patterns(oracle, fmt, gs):
    w, exp = gu.exhaustive_expected(fmt, gs)
    _check(oracle, w, "f16", fmt, gs, exp)


@pytest.mark.parametrize("fmt", gu.FORMATS)
@pytest.mark.parametrize("gs", [None, 64, -1], ids=gu.gs_key)
def test_special_rows(oracle, fmt, gs):
    """+-inf / NaN rows, rows on one side of zero, constant rows, float16 limits: dequantized values, scales and
    zero-points as the reference leaves them (NaN == NaN); codes wherever the dequantized value is a number (the
    integer a NaN code becomes is the host's float -> int cast, not arithmetic)."""
    w, exp = gu.specials_expected(fmt, gs)
    K, C = w.shape
    out = oracle.quantize(w, oracle.F16, group_size=gs, **_fmt_args(oracle, fmt))
    assert gu.same_f16_bits(out["deq"], exp["deq"]), "deq"
    assert gu.same_f16_bits(out["scale"], exp["scale"]), "scale"
    if "zp" in exp:
        assert gu.same_f16_bits(out["zp"], exp["zp"]), "zero-point"
    ok = (exp["deq"].reshape(-1) & 0x7fff) <= 0x7c00
    if fmt.startswith("int"):
        assert np.array_equal(out["q"].reshape(-1)[ok], exp["q"].reshape(-1)[ok]), "codes"
    else:
        assert gu.same_f16_bits(out["q"].reshape(-1)[ok], exp["q"].reshape(-1)[ok]), "codes"


@pytest.mark.parametrize("bits", [8, 4])
@pytest.mark.parametrize("gs", [8, 16, 32, 64, 128, None, -1], ids=gu.gs_key)
def test_zero_point_sign_of_zero(oracle, bits, gs):
    """Segments whose minimum is a zero and that hold +0 and -0 (quant_zero_sign.npz, from the imported reference):
    everything bit for bit; the zero-point's zero carries the sign of -(first zero in memory) for segments of fewer
    than 64 elements.  From 64 elements on torch.amin's choice depends on the vector lanes of the CPU that ran it
    (the fixture shows about half of the segments following the first-in-memory rule): there the sign of a zero
    zero-point is left out of the comparison, and only there."""
    z = gu.npz("quant_zero_sign.npz")
    w = z["in"]
    pre = f"out/int{bits}_asym/{gu.gs_key(gs)}/"
    out = oracle.quantize(w, oracle.F16, group_size=gs, fmt=oracle.INT_ASYM, bits=bits)
    assert np.array_equal(out["deq"].reshape(-1), z[pre + "deq"].reshape(-1))
    assert np.array_equal(out["q"].reshape(-1), z[pre + "q"].reshape(-1))
    assert np.array_equal(out["scale"], z[pre + "scale"])
    assert np.array_equal(z[pre + "zp"], z[pre + "zp_again"])
    got, want = out["zp"].copy(), z[pre + "zp"].copy()
    L = w.size if gs == -1 else (gs or w.shape[1])
    if L < 64:
        assert (want == 0x8000).sum() > 50 and (want == 0).sum() > 50   # both signs occur
    else:
        got[got == 0x8000] = 0
        want[want == 0x8000] = 0
    assert np.array_equal(got, want)


@pytest.mark.parametrize("fmt", ["fp8_e4m3", "fp8_e3m4"])
def test_lut_domain_sweep(oracle, fmt):
    """Every float16 x with |x| <= LUTmax through the reference's LUT rounding (scale == 1)."""
    z = gu.npz("lut_domain.npz")
    w = z[f"{fmt}/in"]
    out = oracle.quantize(w, oracle.F16, oracle.LUT, 8, None, datatype=fmt)
    assert np.all(out["scale"] == 0x3C00)
    assert np.array_equal(out["q"], z[f"{fmt}/q"].reshape(w.shape))
    assert np.array_equal(out["deq"], z[f"{fmt}/deq"].reshape(w.shape))
    e, m, b = oracle.LUT_FORMATS[fmt]
    flat_in, flat_q = w.reshape(-1)[::37], out["q"].reshape(-1)[::37]
    for x, q in zip(flat_in, flat_q):
        assert oracle.lib().orc_lut_round_f16(int(x), e, m, b) == q


def test_lut_tables_closed_form(oracle):
    """Every table format the kernels support equals the reference's value list."""
    ref = gu.load_json("lut_tables.json")
    assert set(oracle.LUT_FORMATS) <= set(ref)
    for name, (e, m, b) in oracle.LUT_FORMATS.items():
        assert np.array_equal(oracle.build_lut(e, m, b, name in oracle.LUT_SUBNORMAL), np.array(ref[name])), name


@pytest.mark.parametrize("bits,fmt", gu.MORE_TABLES)
def test_more_table_formats(oracle, bits, fmt):
    """fp4 / fp5 / fp6 / fp8_e2m5 and the INTn lists as tables: small inputs, all float16 patterns, and the
    rounding-domain sweep (scale == 1)."""
    for name, w, gs, exp in gu.more_cases(fmt):
        K, C = w.shape
        out = oracle.quantize(w, oracle.F16, oracle.LUT, bits, gs, datatype=fmt)
        assert np.array_equal(out["deq"], exp["deq"].reshape(K, C)), (name, gs, "deq")
        assert np.array_equal(out["q"], exp["q"].reshape(K, C)), (name, gs, "q")
        assert np.array_equal(out["scale"], exp["scale"].reshape(-1)), (name, gs, "scale")
        out2 = oracle.quantize(w, oracle.F16, oracle.LUT, bits, gs, datatype=fmt, lut_f32_path=(gs == -1), want=("deq",))
        assert np.array_equal(out2["deq"], exp["deq_quant_datatype"].reshape(K, C)), (name, gs, "quant_datatype")
    z = gu.npz("quant_more_formats.npz")
    w = z[f"domain/{fmt}/in"]
    out = oracle.quantize(w, oracle.F16, oracle.LUT, bits, None, datatype=fmt)
    assert np.all(out["scale"] == 0x3C00)
    assert np.array_equal(out["q"], z[f"domain/{fmt}/q"].reshape(w.shape))


def test_explicit_lists_match_reference():
    """The value lists the product passes to mq_quantize_table equal the reference's lists."""
    from quantization_amd import fused
    ref = gu.load_json("lut_tables.json")
    for name, values in list(fused.TABLE_LISTS.items()) + list(fused.MX_TABLES.items()):
        assert [float(v) for v in values] == ref[name], name
    assert set(ref) == set(fused.LUT_FORMATS) | set(fused.TABLE_LISTS) | set(fused.MX_TABLES)


def _dtype_of(oracle, name):
    return {"f16": oracle.F16, "bf16": oracle.BF16, "f32": oracle.F32}[name.rsplit(".", 1)[1]]


@pytest.mark.parametrize("bits,fmt", gu.LIST_TABLES)
def test_list_table_formats(oracle, bits, fmt):
    """fp3 / flint4 / flint5 and the *_er_* / *_ea_* lists: _fp_quant_meta in float16 arithmetic for every scope,
    quant_datatype's float32 branch per tensor."""
    values = gu.load_json("lut_tables.json")[fmt]
    cases = gu.table_cases(fmt)
    assert cases
    for name, w, gs, exp in cases:
        out = oracle.quantize_table(w, _dtype_of(oracle, name), values, oracle.TABLE_F16, gs)
        for k in ("deq", "q", "scale"):
            assert gu.same_values(out[k], exp[k]), (name, gs, k)
        if gs == -1:
            out2 = oracle.quantize_table(w, _dtype_of(oracle, name), values, oracle.TABLE_F32, gs)
            assert gu.same_values(out2["deq"], exp["deq_quant_datatype"]), (name, gs, "quant_datatype")


def test_fp8_e5m2(oracle):
    """FP8_E5M2: float32 arithmetic, float32 deq / q / scale (quant_weight.py:505-540, :796-826)."""
    values = gu.load_json("lut_tables.json")["fp8_e5m2"]
    cases = gu.table_cases("fp8_e5m2")
    assert cases
    for name, w, gs, exp in cases:
        out = oracle.quantize_table(w, _dtype_of(oracle, name), values, oracle.TABLE_F32, gs, deq_f32=True,
                                    neg_inf_zero=True)
        for k in ("deq", "q", "scale"):
            assert gu.same_values(out[k], exp[k]), (name, gs, k)


@pytest.mark.parametrize("bits,fmt", gu.MX_TABLES)
def test_quant_mx(oracle, bits, fmt):
    values = gu.load_json("lut_tables.json")[fmt]
    cases = gu.table_cases(fmt)
    assert cases
    for name, w, gs, exp in cases:
        out = oracle.quantize_table(w, _dtype_of(oracle, name), values, oracle.TABLE_MX, gs)
        assert gu.same_values(out["deq"], exp["deq"]), (name, gs)


def test_search_datatype(oracle):
    """search_datatype: per row / group the first candidate list with the smallest float16 mean squared error."""
    from quantization_amd import fused
    cases = gu.search_cases()
    assert len(cases) == 128
    for name, w, bits, kind, gs, exp in cases:
        tables = [fused._SEARCH_VALUES[n] for n in fused.MIXED_LISTS[(bits, kind)]]
        out = oracle.search_tables(w, tables, gs)
        assert gu.same_values(out["deq"], exp), (name, bits, kind, gs)


@pytest.mark.parametrize("fmt", ["int4", "int4_asym"])
def test_four_bit_int(oracle, fmt):
    for name, w, gs, exp in gu.more_cases(fmt + ".int"):
        K, C = w.shape
        out = oracle.quantize(w, oracle.F16, oracle.INT_ASYM if "asym" in fmt else oracle.INT_SYM, 4, gs)
        assert np.array_equal(out["deq"], exp["deq"].reshape(K, C)), (name, gs)
        assert np.array_equal(out["q"], exp["q"].reshape(K, C)), (name, gs)
        assert np.array_equal(out["scale"], exp["scale"].reshape(-1)), (name, gs)
        if "zp" in exp:
            assert np.array_equal(out["zp"], exp["zp"].reshape(-1)), (name, gs)


def test_software_half_conversion(oracle):
    L = oracle.lib()
    bits = np.arange(65536, dtype=np.uint16)
    f = bits.view(np.float16).astype(np.float32)
    for b in range(0, 65536, 7):
        x = L.orc_h2f_soft(b)
        assert L.orc_h2f(b) == x or np.isnan(x)
        if np.isnan(f[b]):
            assert np.isnan(x)
        else:
            assert x == f[b] and np.signbit(x) == np.signbit(f[b])
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.standard_normal(4000).astype(np.float32) * s for s in (1e-7, 1e-4, 1, 300, 70000)])
    xs = np.concatenate([xs, np.array([65504, 65519.99, 65520, 2 ** -24, 2 ** -25, 2 ** -25 * 1.0001, 0.0, -0.0,
                                       np.inf, -np.inf], np.float32)])
    want = xs.astype(np.float16).view(np.uint16)
    for x, wv in zip(xs, want):
        assert L.orc_f2h_soft(float(x)) == wv and L.orc_f2h(float(x)) == wv


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_error_stats_known_answers(oracle):
    """Quantize with the C oracle, run the numpy error half, compare with the stored known answers."""
    gold = gu.load_json("error_stats.json")
    K, C = gold["K"], gold["C"]
    rng = np.random.default_rng(gold["seed"])
    inputs = {"normal": (rng.standard_normal((K, C)) * 0.02).astype(np.float16),
              "student_t": (rng.standard_t(3, (K, C)) * 0.02).astype(np.float16)}
    for cname, w16 in inputs.items():
        g = gold["cases"][cname]
        assert _digest(w16) == g["input_sha256"], "numpy RNG stream changed; regenerate the golden file"
        for key, e in g["formats"].items():
            fmt, gk = key.split("/")
            gs = None if gk == "none" else int(gk)
            out = oracle.quantize(w16, oracle.F16, group_size=gs, **_fmt_args(oracle, fmt))
            assert _digest(out["deq"]) == e["deq_sha256"]
            assert _digest(out["q"]) == e["q_sha256"]
            assert _digest(out["scale"]) == e["scale_sha256"]
            if "zp_sha256" in e:
                assert _digest(out["zp"]) == e["zp_sha256"]
            if fmt == "int8_asym" and gs in (None, 64):
                deq = out["deq"].view(np.float16)
                grouping = "per-channel" if gs is None else "per-group"
                m = oracle.compute_errors_multi(w16.astype(np.float32), deq.astype(np.float32), grouping, gs)
                assert m["mse_tensor"] == e["mse_tensor"] and m["mse_p99"] == e["multi"]["mse_p99"]
                el = oracle.elementwise_f32(w16, deq)
                assert int(np.float32(el["p99"]).view(np.uint32)) == e["abs_p99_f32_bits"]
