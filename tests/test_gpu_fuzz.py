"""GPU: a few seconds of the randomised differential runs (tests/fuzz_stats.py: statistics kernels against the CPU
oracle; tests/fuzz_quant.py: the general kernels, every output, against the CPU oracle; tests/fuzz_tables.py: the
explicit-list formats, MX and the datatype search against the CPU oracle; tests/fuzz_select.py:
percentile selection against numpy; tests/fuzz_errors.py: the error half against the scripts' numpy arithmetic;
tests/fuzz_sweep.py: SweepPlan against model_records against layer_record; tests/fuzz_numpy_order.py: the numpy-order
list entries, list means and tensor means against numpy) with fixed seeds, and the cases those runs have found.  Longer runs:
`python tests/fuzz_stats.py 90 1`, `python tests/fuzz_quant.py 100 1`, `python tests/fuzz_select.py 120 1`."""
import pytest

pytestmark = pytest.mark.gpu


def test_statistics_kernels_random_cases(oracle):
    import fuzz_stats
    assert fuzz_stats.run(6.0, 12345) > 50


def test_numpy_order_modes_random_cases(oracle):
    import fuzz_numpy_order
    assert fuzz_numpy_order.run(6.0, 2024) > 20


def test_percentile_selection_random_cases():
    import fuzz_select
    assert fuzz_select.run(8.0, 54321) > 20


def test_general_kernels_random_cases(oracle):
    import fuzz_quant
    assert fuzz_quant.run(8.0, 777) > 100


def test_scale_cut_to_the_upper_clamp_keeps_the_code_clamp(oracle):
    """Found by fuzz_quant: symmetric 2- / 3-bit formats on values above 1e4 * qmax.  The scale is cut down to the
    clamp's upper bound (float16 0x70E2), |v / s| then exceeds qmax and clamp(rint(x), -qmax, qmax)
    (quant_weight.py:262-263) acts -- on every grouping, float16 / bfloat16 / float32 weights."""
    import numpy as np
    import torch
    import golden_util as gu
    from quantization_amd import _lib
    from quantization_amd.fused import quantize
    g = torch.Generator().manual_seed(3)
    w = torch.randn(24, 2048, generator=g) * 9000.0
    w[3, 5], w[7, 9], w[11, 100] = 65504.0, -64000.0, 31000.0
    for dt, odt in ((torch.float16, oracle.F16), (torch.bfloat16, oracle.BF16), (torch.float32, oracle.F32)):
        wt = w.to(dt)
        raw = wt.numpy() if dt == torch.float32 else wt.view(torch.int16).numpy().view(np.uint16)
        for bits in (2, 3):
            for gs in (None, 64, 16, -1):
                exp = oracle.quantize(raw, odt, fmt=oracle.INT_SYM, bits=bits, group_size=gs)
                assert (exp["scale"] == 0x70E2).any()
                res = quantize(wt.cuda(), _lib.INT_SYM, bits, gs, None, deq=True, codes=True, scale=True)
                b16 = lambda t: t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16).reshape(-1)
                what = (str(dt), bits, gs)
                assert gu.same_f16_bits(b16(res.scale), exp["scale"].reshape(-1)), what
                assert gu.same_f16_bits(b16(res.deq), exp["deq"].reshape(-1)), what
                assert np.array_equal(res.codes.cpu().numpy().reshape(-1).view(np.uint8), exp["q"].reshape(-1)), what


def test_table_formats_random_cases(oracle):
    import fuzz_tables
    assert fuzz_tables.run(6.0, 4242) > 100


def test_float32_branch_rounds_the_product_twice(oracle):
    """Found by fuzz_tables: quant_datatype's per-tensor branch is float32 arithmetic whose product q * scale is a
    float32 tensor before .to(torch.float16) (quant_weight.py:575) -- two roundings.  A multiply folded into the
    conversion (v_fma_mixlo_f16) rounds once and lands one float16-subnormal step lower on ties such as
    0.234375 * 1.402537e-4 = 551.4999.. * 2^-24.  Student-t weights of scale 1e-3 hold a few such elements each."""
    import numpy as np
    import torch
    from quantization_amd import quant_weight as qw
    rng = np.random.default_rng(5)
    hit = 0
    for _ in range(6):
        w = (torch.from_numpy(rng.standard_t(3, (17, 1816)).astype(np.float32)) * 1e-3).half()
        for name in ("fp8_e4m3", "fp4", "flint4"):
            d = qw.quant_datatype(w.cuda(), wq_bits=8 if name == "fp8_e4m3" else 4, datatype=name, group_size=-1)
            got = d.cpu().view(torch.int16).numpy().view(np.uint16).reshape(-1)
            if name == "flint4":
                from quantization_amd import fused
                exp = oracle.quantize_table(w.numpy(), oracle.F16, fused.TABLE_LISTS[name], oracle.TABLE_F32, -1)
            else:
                exp = oracle.quantize(w.numpy(), oracle.F16, fmt=oracle.LUT, bits=8 if name == "fp8_e4m3" else 4,
                                      datatype=name, group_size=-1, lut_f32_path=True, want=("deq",))
            assert np.array_equal(got, exp["deq"].reshape(-1)), name
            # the case is present: some product lies within 2^-12 of a float16-subnormal tie
            hit += 1
    assert hit == 18


def test_error_half_random_cases(oracle):
    import fuzz_errors
    assert fuzz_errors.run(5.0, 99) > 100


def test_model_sweep_random_cases():
    import fuzz_sweep
    assert fuzz_sweep.run(5.0, 31) > 50
