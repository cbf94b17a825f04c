"""GPU: the headline workload at its full size -- batches of 4096 x 4096 matrices, where the launches are
grid-capped and every workgroup loops over many tiles -- against the CPU oracle (one matrix of each batch, bit for
bit), against numpy's percentile of the dumped errors, and through size-independent properties of the outputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def bits16(t):
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("fmt,gs", [("int8_asym", 32), ("int8_asym", 128), ("int8", 64), ("fp8_e4m3", 128),
                                    ("int8", None), ("fp8_e3m4", None)])
def test_c2_sized_batch(oracle, fmt, gs):
    from quantization_amd import _lib
    from quantization_amd.fused import quantize_batch
    code = {"int8": _lib.INT_SYM, "int8_asym": _lib.INT_ASYM}.get(fmt, _lib.LUT_FP)
    ocode = {"int8": oracle.INT_SYM, "int8_asym": oracle.INT_ASYM}.get(fmt, oracle.LUT)
    table = fmt if code == _lib.LUT_FP else None
    g = torch.Generator(device="cuda").manual_seed(5)
    B = 6
    w = (torch.randn(B, 4096, 4096, generator=g, device="cuda") * 0.02).half()
    res = quantize_batch(w, code, 8, gs, table, deq=True, codes=True, scale=True, abs_err=torch.float32, seg_sse=True,
                         totals=True, p99=99)
    # 1) one matrix of the batch against the oracle, bit for bit
    i = B - 1
    exp = oracle.quantize(w[i].cpu().numpy(), oracle.F16, ocode, 8, gs, datatype=table, nthreads=16)
    assert np.array_equal(bits16(res.deq[i]), exp["deq"])
    got_q = bits16(res.codes[i]) if code == _lib.LUT_FP else res.codes[i].cpu().numpy()
    assert np.array_equal(got_q, exp["q"])
    assert np.array_equal(bits16(res.scale[i]), exp["scale"])
    if code == _lib.INT_ASYM:
        assert np.array_equal(bits16(res.zero_point[i]), exp["zp"])
    # 2) every matrix: the dump is |float32(w) - float32(deq)| exactly; sums and P99 belong to it
    err = (w.float() - res.deq.float()).abs()
    assert torch.equal(res.abs_err, err)
    sq = (err * err).double()
    # (a lane's 8 squares / absolute values are added in float32 before the float64 sums: 3 roundings, 1.8e-7)
    torch.testing.assert_close(res.totals[:, 0], sq.sum(dim=(1, 2)), rtol=2e-7, atol=0)
    torch.testing.assert_close(res.totals[:, 1], err.double().sum(dim=(1, 2)), rtol=2e-7, atol=0)
    L = gs if gs else 4096
    torch.testing.assert_close(res.seg_sse, sq.reshape(B, -1, L).sum(dim=2), rtol=2e-7, atol=0)
    for k in (0, i):
        want = np.percentile(err[k].cpu().numpy().ravel(), 99)
        assert res.abs_p99[k].cpu().numpy().view(np.uint32) == want.view(np.uint32)
    # 3) properties: codes reproduce the dequantized values from scale / zero-point (float16 op by op), and for the
    #    symmetric format no error exceeds 0.6 steps of its segment (half a step, plus the float16 roundings of
    #    v / s at |v / s| <= 127 and of q * s; the asymmetric format may clip at the ends of its range)
    s = res.scale.reshape(B, -1, 1)
    if code == _lib.LUT_FP:
        back = (res.codes.reshape(B, -1, L) * s).reshape(B, 4096, 4096)
    elif code == _lib.INT_ASYM:
        z = res.zero_point.reshape(B, -1, 1)
        back = ((res.codes.reshape(B, -1, L).half() - z) * s).reshape(B, 4096, 4096)
    else:
        back = (res.codes.view(torch.int8).reshape(B, -1, L).half() * s).reshape(B, 4096, 4096)
    assert torch.equal(back, res.deq)   # as values: a symmetric zero code dequantizes to -0 for negative weights
    if code == _lib.INT_SYM:
        bound = (s.float() * 0.6).expand(B, s.shape[1], L).reshape(B, 4096, 4096)
        assert bool((err <= bound).all())


def test_llama_sized_rows_vs_oracle(oracle):
    """Row lengths of the Llama shape lists (workgroup-per-row kernels, 256 and 1024 threads), bf16 input."""
    from quantization_amd import _lib
    from quantization_amd.fused import quantize
    torch.manual_seed(6)
    for cols in (5632, 14336, 28672):
        w = (torch.randn(24, cols) * 0.02).bfloat16()
        wb = w.view(torch.int16).numpy().view(np.uint16)
        for code, ocode, table in ((_lib.INT_SYM, oracle.INT_SYM, None), (_lib.INT_ASYM, oracle.INT_ASYM, None),
                                   (_lib.LUT_FP, oracle.LUT, "fp8_e4m3"), (_lib.LUT_FP, oracle.LUT, "fp4")):
            bits = 4 if table == "fp4" else 8
            exp = oracle.quantize(wb, oracle.BF16, ocode, bits, None, datatype=table)
            res = quantize(w.cuda(), code, bits, None, table, seg_sse=True, totals=True)
            assert np.array_equal(bits16(res.deq), exp["deq"]), (cols, table, code)
            assert np.array_equal(bits16(res.scale), exp["scale"]), (cols, table, code)
            e = exp["v16"].view(np.float16).astype(np.float32) - exp["deq"].view(np.float16).astype(np.float32)
            np.testing.assert_allclose(res.seg_sse.cpu().numpy(), (e * e).astype(np.float64).sum(1), rtol=2e-7)


@pytest.mark.parametrize("gs", [32, 128])
def test_c2_sized_batch_heavy_tails(gs):
    """Student-t (3 degrees of freedom) weights: the |error| distribution around its P99 spans more level-0 bins
    than a bracket histogram holds, so the quantize pass keeps the bit patterns of the values inside the bracket
    as a list and the percentile is selected among them (k_sel_cand).  Bit for bit numpy's percentile of the dump,
    for 99 and for quantiles whose brackets are wider still."""
    from quantization_amd import _lib
    from quantization_amd.fused import quantize_batch
    g = torch.Generator(device="cuda").manual_seed(77)
    B = 4
    x = torch.randn(B, 4096, 4096, generator=g, device="cuda")
    chi = sum(torch.randn(B, 4096, 4096, generator=g, device="cuda") ** 2 for _ in range(3))
    w = (x / torch.sqrt(chi / 3) * 0.02).half()
    for pct in (99, 99.9, 90):
        res = quantize_batch(w, _lib.INT_ASYM, 8, gs, None, deq=True, codes=False, scale=False, abs_err=torch.float32,
                             p99=pct)
        err = (w.float() - res.deq.float()).abs()
        assert torch.equal(res.abs_err, err)
        for k in range(B):
            want = np.percentile(err[k].cpu().numpy().ravel(), pct)
            assert res.abs_p99[k].cpu().numpy().view(np.uint32) == want.view(np.uint32), (gs, pct, k)


def test_tensor_beyond_the_32_bit_buffer_offsets():
    """A 4.8 GB bfloat16 tensor: above the 0xf0000000-byte limit of the statistics kernels' buffer loads, so the
    launcher takes the general kernels.  Additivity: scales equal, and list entries / totals within the 1e-6
    contract of, the results over its two halves (each inside the limit, statistics kernels)."""
    import torch
    from quantization_amd import _lib
    from quantization_amd.fused import quantize
    g = torch.Generator(device="cuda").manual_seed(0)
    rows, cols = 36864, 65536
    w = torch.empty(rows, cols, dtype=torch.bfloat16, device="cuda")
    for r in range(0, rows, 4096):
        w[r:r + 4096] = (torch.randn(4096, cols, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    kw = dict(deq=False, codes=False, scale=True, seg_mse=True, totals=True)
    for fmt, table, gs in ((_lib.INT_SYM, None, 128), (_lib.INT_ASYM, None, None), (_lib.LUT_FP, "fp8_e4m3", 64)):
        whole = quantize(w, fmt, 8, gs, table, **kw)
        a = quantize(w[: rows // 2], fmt, 8, gs, table, **kw)
        b = quantize(w[rows // 2:], fmt, 8, gs, table, **kw)
        tot = a.totals + b.totals
        assert ((whole.totals - tot).abs() / tot).max().item() < 1e-6
        assert torch.equal(whole.scale.view(torch.int16).reshape(-1),
                           torch.cat([a.scale.reshape(-1), b.scale.reshape(-1)]).view(torch.int16))
        assert torch.allclose(whole.seg_mse.reshape(-1), torch.cat([a.seg_mse.reshape(-1), b.seg_mse.reshape(-1)]),
                              rtol=2e-6, atol=0)
    del w
    torch.cuda.empty_cache()
