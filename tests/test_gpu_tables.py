"""GPU: the explicit-list formats (mq_quantize_table through the reference-shaped API) against vectors produced by
the imported reference: fp3 / flint4 / flint5 and the *_er_* / *_ea_* lists, fp8_e5m2 (float32 results) and
quant_mx.  Bit-exact (a NaN matches any NaN); error statistics against a float64 sum of the same float32 errors."""
import numpy as np
import os

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


def to_dev(arr, name):
    dt = name.rsplit(".", 1)[1]
    if dt == "f32":
        return torch.from_numpy(arr.copy()).cuda()
    t = torch.from_numpy(arr.view(np.int16).copy()).cuda()
    return t.view(torch.float16 if dt == "f16" else torch.bfloat16)


def bits16(t):
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("bits,fmt", gu.LIST_TABLES)
def test_golden_list_tables(qa, bits, fmt):
    qw = qa.quant_weight
    cases = gu.table_cases(fmt)
    assert cases
    for name, arr, gs, exp in cases:
        w = to_dev(arr, name)
        d, q, s = qw._fp_quant_meta(w, wq_bits=bits, datatype=fmt, group_size=gs)
        assert d.dtype == q.dtype == s.dtype == torch.float16 and d.shape == w.shape
        assert gu.same_values(bits16(d), exp["deq"]), (name, gs, "deq")
        assert gu.same_values(bits16(q), exp["q"]), (name, gs, "q")
        assert gu.same_values(bits16(s.reshape(-1)), exp["scale"]), (name, gs, "scale")
        if gs is not None and gs > 0:
            assert q.shape == (w.shape[0], w.shape[1] // gs, gs) and s.shape == (w.shape[0], w.shape[1] // gs)
        d2 = qw.quant_datatype(w, wq_bits=bits, datatype=fmt, group_size=gs)
        assert d2.dtype == torch.float16 and d2.shape == w.shape
        assert gu.same_values(bits16(d2), exp["deq_quant_datatype" if gs == -1 else "deq"]), (name, gs, "qd")


def test_golden_fp8_e5m2(qa):
    qw = qa.quant_weight
    cases = gu.table_cases("fp8_e5m2")
    assert cases
    for name, arr, gs, exp in cases:
        w = to_dev(arr, name)
        d, q, s = qw._fp_quant_meta(w, wq_bits=8, datatype="fp8_e5m2", group_size=gs)
        assert d.dtype == q.dtype == s.dtype == torch.float32
        assert d.shape == w.shape and q.shape == w.shape
        assert s.shape == (() if gs == -1 else ((w.shape[0],) if gs is None else (w.shape[0], w.shape[1] // gs)))
        assert gu.same_values(d.cpu().numpy(), exp["deq"]), (name, gs, "deq")
        assert gu.same_values(q.cpu().numpy(), exp["q"]), (name, gs, "q")
        assert gu.same_values(s.reshape(-1).cpu().numpy(), exp["scale"]), (name, gs, "scale")
        d2 = qw.quant_datatype(w, wq_bits=4, datatype="fp8_e5m2", group_size=gs)  # bit width is not looked at
        assert d2.dtype == torch.float32 and gu.same_values(d2.cpu().numpy(), exp["deq"]), (name, gs, "qd")


@pytest.mark.parametrize("bits,fmt", gu.MX_TABLES)
def test_golden_quant_mx(qa, bits, fmt):
    cases = gu.table_cases(fmt)
    assert cases
    for name, arr, gs, exp in cases:
        w = to_dev(arr, name)
        d = qa.quant_weight.quant_mx(w, wq_bits=bits, datatype=fmt, group_size=gs)
        assert d.dtype == torch.float16 and d.shape == w.shape
        assert gu.same_values(bits16(d), exp["deq"]), (name, gs)


@pytest.mark.parametrize("shape,gs", [((24, 8192), None), ((16, 4096), -1), ((3, 200), 40), ((5, 33), None)])
def test_table_vs_oracle_and_error_outputs(qa, oracle, shape, gs):
    """Shapes the fixtures do not hold: workgroup-per-segment rows (> 2048 elements), per-tensor scope, odd
    lengths; and the error outputs against a float64 sum of the float32 errors."""
    from quantization_amd import _lib, fused
    torch.manual_seed(3)
    w = (torch.randn(*shape) * 0.05).half()
    values = fused.TABLE_LISTS["flint4"]
    exp = oracle.quantize_table(w.numpy(), oracle.F16, values, oracle.TABLE_F16, gs)
    res = fused.quantize_table(w.cuda(), values, _lib.TABLE_F16, gs, abs_err=torch.float32, seg_sse=True, totals=True)
    assert np.array_equal(bits16(res.deq), exp["deq"])
    assert np.array_equal(bits16(res.codes), exp["q"])
    assert np.array_equal(bits16(res.scale), exp["scale"])
    err = w.float().numpy() - exp["deq"].view(np.float16).astype(np.float32)
    assert np.array_equal(res.abs_err.cpu().numpy(), np.abs(err))
    sq = (err * err).astype(np.float64)
    L = w.numel() if gs == -1 else (shape[1] if gs is None else gs)
    np.testing.assert_allclose(res.seg_sse.cpu().numpy(), sq.reshape(-1, L).sum(1), rtol=1e-12)
    np.testing.assert_allclose(res.totals.cpu().numpy(), [sq.sum(), np.abs(err).astype(np.float64).sum()], rtol=1e-12)


def test_golden_search_datatype(qa, oracle):
    from quantization_amd import fused
    qw = qa.quant_weight
    cases = gu.search_cases()
    assert len(cases) == 128
    for name, arr, bits, kind, gs, exp in cases:
        w = to_dev(arr, name + ".f16")
        d = qw.search_datatype(w, wq_bits=bits, datatype=kind, group_size=gs)
        assert d.dtype == torch.float16 and d.shape == w.shape
        assert gu.same_values(bits16(d), exp), (name, bits, kind, gs)
    # the chosen list per segment, and rows longer than a wave's share, against the oracle
    torch.manual_seed(5)
    w = (torch.randn(6, 8192) * 0.02).half()
    tables = [fused._SEARCH_VALUES[n] for n in fused.MIXED_LISTS[(4, "mixed_bitmod")]]
    for gs in (None, 128):
        exp = oracle.search_tables(w.numpy().view(np.uint16), tables, gs)
        d, choice = fused.search_tables(w.cuda(), tables, gs, want_choice=True)
        assert np.array_equal(bits16(d), exp["deq"]) and np.array_equal(choice.cpu().numpy(), exp["choice"])
        assert len(set(exp["choice"].tolist())) > 1


def test_table_errors_and_model_routes(qa):
    import torch.nn as nn
    from quantization_amd import _lib, fused
    qw = qa.quant_weight
    w = (torch.randn(4, 96) * 0.02).half().cuda()
    with pytest.raises(ValueError):
        qw.quant_mx(w, 5, "mx_fp4", 32)
    with pytest.raises(AssertionError):
        qw.quant_mx(w, 4, "mx_fp3", 32)
    with pytest.raises(RuntimeError):
        qw.quant_mx(w, 4, "mx_fp4", 64)        # 96 % 64
    with pytest.raises(AssertionError):
        qw.quant_datatype(w, 3, "flint4", None)   # a 4-bit name under 3 bits
    with pytest.raises(ValueError):
        fused.quantize_table(w, [1.0, 0.5], _lib.TABLE_F16)   # not ascending
    with pytest.raises(ValueError):
        fused.quantize_table(w, [0.0, 1.0 + 2.0 ** -12], _lib.TABLE_F16)   # not a float16 value
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 32, bias=False)).half().cuda()
    w0 = model[0].weight.data.clone()
    qw.quant_model(model, 4, "mx_fp4", None)
    assert type(model[0]) is nn.Linear   # MX layers are not wrapped (the reference's name test)
    assert torch.equal(model[0].weight.data.view(torch.int16), qw.quant_mx(w0, 4, "mx_fp4", 32).view(torch.int16))
    model = nn.Sequential(nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 32, bias=False)).half().cuda()
    w0 = model[0].weight.data.clone()
    qw.quant_model(model, 4, "fp4_er_pos", 64)
    assert type(model[0]).__name__ == "_LinearDump" and model[0].weight_scale.shape == (64, 2)
    assert torch.equal(model[0].weight.data.view(torch.int16),
                       qw.quant_datatype(w0, 4, "fp4_er_pos", 64).view(torch.int16))
    model = nn.Sequential(nn.Linear(128, 64)).half().cuda()
    qw.quant_model(model, 8, "fp8_e5m2", None)
    assert model[0].weight.dtype == torch.float32 and model[0].weight_q.dtype == torch.float32
    with pytest.raises(ValueError):
        qw.quant_model(nn.Sequential(nn.Linear(8, 8)).half().cuda(), 4, "flint4", None)   # as in the reference
    with pytest.raises(ValueError):
        qw.search_datatype(w, 5, "mixed_er", None)
    # weights that are not float16: the reference's own exception and message (tests/golden/search_errors.json)
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "search_errors.json")) as f:
        recorded = json.load(f)
    for key, (exc, msg) in recorded.items():
        dt = getattr(torch, key.split("/")[0])
        gs = None if key.endswith("none") else int(key.split("/")[1])
        assert exc == "RuntimeError"
        with pytest.raises(RuntimeError) as info:
            qw.search_datatype((torch.randn(8, 256) * 0.02).to(dt).cuda(), 4, "mixed_bitmod", gs)
        assert str(info.value) == msg
    model = nn.Sequential(nn.Linear(128, 64)).half().cuda()
    w0 = model[0].weight.data.clone()
    qw.quant_model(model, 3, "mixed_ant", 32)
    assert type(model[0]) is nn.Linear
    assert torch.equal(model[0].weight.data.view(torch.int16),
                       qw.search_datatype(w0, 3, "mixed_ant", 32).view(torch.int16))
