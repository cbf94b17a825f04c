"""GPU: mq_error_stats_static -- the entry point that allocates nothing (its partial sums live in a device array
of the library) -- against mq_error_stats on the same inputs, through the C ABI: two calls in a row, the second
with a smaller input, every output compared bit for bit (same kernels, fixed summation order); then the size
limit."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _call(L, fn, a, b, rows, cols, seg_len, ws=None):
    dev = a.device
    out = dict(abs_err=torch.empty(rows * cols, dtype=torch.float32, device=dev),
               seg_sse=torch.empty(rows * cols // seg_len, dtype=torch.float64, device=dev),
               col_sse=torch.empty(cols, dtype=torch.float64, device=dev),
               totals=torch.empty(2, dtype=torch.float64, device=dev))
    for t in out.values():
        t.fill_(-1.0)
    st = torch.cuda.current_stream().cuda_stream
    args = [a.data_ptr(), b.data_ptr(), 0, rows, cols, seg_len, out["abs_err"].data_ptr(), 2,
            out["seg_sse"].data_ptr(), out["col_sse"].data_ptr(), out["totals"].data_ptr()]
    if ws is not None:
        args += [ws.data_ptr(), ws.numel()]
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]          # hipMemGetInfo: device memory nobody has allocated
    rc = fn(*args, st)
    torch.cuda.synchronize()
    out["_allocated"] = free0 - torch.cuda.mem_get_info()[0]
    return rc, out


def test_static_entry_point_equals_workspace_entry_point():
    from quantization_amd import _lib
    L = _lib.lib()
    vp, i64, i32, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_size_t
    L.mq_error_workspace_bytes.restype = sz
    L.mq_error_workspace_bytes.argtypes = [i64, i64, i32]
    L.mq_error_stats.restype = ctypes.c_int
    L.mq_error_stats.argtypes = [vp, vp, i32, i64, i64, i64, vp, i32, vp, vp, vp, vp, sz, vp]
    L.mq_error_stats_static.restype = ctypes.c_int
    L.mq_error_stats_static.argtypes = [vp, vp, i32, i64, i64, i64, vp, i32, vp, vp, vp, vp]
    torch.cuda.set_device(0)
    g = torch.Generator(device="cuda").manual_seed(3)
    # the second call is smaller: whatever the first left in the array must not leak into it
    for rows, cols, seg in [(1536, 4096, 128), (48, 1000, 125)]:
        a = (torch.randn(rows, cols, generator=g, device="cuda") * 0.02).half()
        b = (a.float() + torch.randn(rows, cols, generator=g, device="cuda") * 1e-3).half()
        ws = torch.empty(L.mq_error_workspace_bytes(rows, cols, 1), dtype=torch.uint8, device="cuda")
        rc0, want = _call(L, L.mq_error_stats, a, b, rows, cols, seg, ws)
        rc1, got = _call(L, L.mq_error_stats_static, a, b, rows, cols, seg)
        assert rc0 == 0 and rc1 == 0, (rc0, rc1)
        want.pop("_allocated")
        assert got.pop("_allocated") == 0, "the static entry point took device memory"
        for k in want:
            assert torch.equal(got[k].view(torch.int64 if got[k].dtype == torch.float64 else torch.int32),
                               want[k].view(torch.int64 if want[k].dtype == torch.float64 else torch.int32)), (rows, cols, k)
        # and against numpy, so that two wrong results cannot agree
        e = a.float().cpu().numpy() - b.float().cpu().numpy()
        sq = (e * e).astype(np.float64)
        np.testing.assert_allclose(got["totals"].cpu().numpy(), [sq.sum(), np.abs(e).astype(np.float64).sum()], rtol=1e-6)
        np.testing.assert_allclose(got["col_sse"].cpu().numpy(), sq.sum(axis=0), rtol=1e-6)
        np.testing.assert_allclose(got["seg_sse"].cpu().numpy(), sq.reshape(-1, seg).sum(axis=1), rtol=1e-6)
    # beyond the fixed capacity: refused before anything runs
    a = torch.zeros(2, 8200, dtype=torch.float16, device="cuda")
    col = torch.empty(8200, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rc = L.mq_error_stats_static(a.data_ptr(), a.data_ptr(), 0, 2, 8200, 0, None, 2, None, col.data_ptr(), None, st)
    assert rc == _lib.ERR_TOO_LARGE and b"capacity" in L.mq_strerror(rc)
    # without column sums the array holds only per-workgroup partials: any width is fine
    tot = torch.empty(2, dtype=torch.float64, device="cuda")
    rc = L.mq_error_stats_static(a.data_ptr(), a.data_ptr(), 0, 2, 8200, 0, None, 2, None, None, tot.data_ptr(), st)
    torch.cuda.synchronize()
    assert rc == 0 and tot.tolist() == [0.0, 0.0]
