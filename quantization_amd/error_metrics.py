"""Error analysis on the device: the arithmetic of the reference's MSE&P99/ and elwise_error/ scripts,
taking the two tensors those scripts read from the baseline and quantized ONNX files.

Citations are into /root/reference.  Every function takes ROCm tensors (host tensors are moved to the
current device) and runs the HIP kernels behind include/mq.h: mq_error_stats, mq_percentile,
mq_bucketize_f16, mq_compare_formats_f16.  Sums of squares are carried in float64 (the scripts use
numpy's float32 pairwise mean), so MSE values agree with the scripts to ~1e-7 relative, not bit for bit;
percentiles of a given array, element-wise errors, bucket indices and masks are bit-exact.
"""
import ctypes
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .fused import require_device, workspace

_DT = {torch.float16: _lib.F16, torch.float32: _lib.F32, torch.float64: _lib.F64}


def _sel_dtype(dtype, wide):
    """(dtype code, result dtype) of a selection.  wide: float32 values standing for the float64 array that
    holds them (a list of float32 means turned into np.array(...)): float32 keys, float64 ranks and results."""
    if wide:
        if dtype != torch.float32:
            raise ValueError("wide=True takes float32 tensors")
        return _lib.F32_AS_F64, torch.float64
    return _DT[dtype], dtype


def _bind():
    L = _lib.lib()
    if getattr(L, "_err_bound", False):
        return L
    vp, i64, i32, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_size_t
    L.mq_percentile_workspace_bytes.restype = sz
    L.mq_percentile.restype = ctypes.c_int
    L.mq_percentile.argtypes = [vp, i32, i64, ctypes.c_double, vp, vp, vp, sz, vp]
    L.mq_percentile_batch_workspace_bytes.restype = sz
    L.mq_percentile_batch_workspace_bytes.argtypes = [i32]
    L.mq_percentile_batch.restype = ctypes.c_int
    L.mq_percentile_batch.argtypes = [ctypes.POINTER(vp), i32, i32, i64, ctypes.c_double, vp, vp, vp, sz, vp]
    L.mq_percentile_multi_beside.restype = ctypes.c_int
    L.mq_percentile_multi_beside.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(i64), i32, i32, ctypes.c_double, vp, vp, vp,
                                             ctypes.c_size_t, vp]
    L.mq_percentile_multi.restype = ctypes.c_int
    L.mq_percentile_multi.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(i64), i32, i32, ctypes.c_double, vp, vp, vp,
                                      sz, vp]
    L.mq_error_workspace_bytes.restype = sz
    L.mq_error_workspace_bytes.argtypes = [i64, i64, i32]
    L.mq_error_stats.restype = ctypes.c_int
    L.mq_error_stats.argtypes = [vp, vp, i32, i64, i64, i64, vp, i32, vp, vp, vp, vp, sz, vp]
    L.mq_bucketize_f16.restype = ctypes.c_int
    L.mq_bucketize_f16.argtypes = [vp, i64, ctypes.POINTER(ctypes.c_double), i32, vp, vp]
    L.mq_compare_formats_f16.restype = ctypes.c_int
    L.mq_compare_formats_f16.argtypes = [vp, ctypes.POINTER(vp), i32, i64, ctypes.c_uint16, vp, vp, vp]
    L.mq_minmax_workspace_bytes.restype = sz
    L.mq_minmax.restype = ctypes.c_int
    L.mq_minmax.argtypes = [vp, i32, i64, i32, vp, vp, sz, vp]
    L.mq_histogram.restype = ctypes.c_int
    L.mq_histogram.argtypes = [vp, i32, i64, vp, i32, vp, vp]
    L._err_bound = True
    return L


def _ptr(t):
    return None if t is None else t.data_ptr()


@torch.no_grad()
def percentile(x: torch.Tensor, pct: float = 99.0, return_pair: bool = False, nan_check: bool = True,
               wide: bool = False):
    """np.percentile(x, pct) (method 'linear') of a non-negative float32 / float64 device tensor, exact:
    two-pass (float32) or four-pass (float64) radix select on the bit patterns.  Returns a 0-d device
    tensor of x's dtype (and the two interpolated order statistics when return_pair).  nan_check: one more pass
    over x that turns the result into NaN when x holds one, as numpy does (without it a NaN counts as the
    largest value).  wide (float32 x): np.percentile(x.astype(np.float64), pct) -- the same two passes over the
    float32 bits, ranks / interpolation / result in float64."""
    x = require_device(x)
    if x.dtype not in (torch.float32, torch.float64):
        raise ValueError("percentile() takes float32 or float64")
    code, rdt = _sel_dtype(x.dtype, wide)
    x = x.contiguous().reshape(-1)
    if x.numel() == 0:
        raise IndexError("percentile of an empty array")
    if x.data_ptr() % 16:
        x = x.clone()   # (a sliced view: the kernels read 16-byte vectors)
    L = _bind()
    out = torch.empty((), dtype=rdt, device=x.device)
    pair = torch.empty(2, dtype=rdt, device=x.device) if return_pair else None
    with torch.cuda.device(x.device):
        ws, stream = workspace(x.device, L.mq_percentile_workspace_bytes())
        st = L.mq_percentile(x.data_ptr(), code, x.numel(), float(pct), out.data_ptr(), _ptr(pair),
                             ws.data_ptr(), ws.numel(), stream.cuda_stream)
    _lib.check(st)
    if nan_check:
        # the select orders a NaN above every number; np.percentile returns NaN when the array holds one
        has_nan = torch.isnan(x).any()
        out = torch.where(has_nan, torch.full_like(out, float("nan")), out)
        if pair is not None:   # (the two order statistics were ranked with NaN as the largest value: not numpy's)
            pair = torch.where(has_nan, torch.full_like(pair, float("nan")), pair)
    return (out, pair) if return_pair else out


@torch.no_grad()
def percentile_batch(x: torch.Tensor, pct: float = 99.0, wide: bool = False) -> torch.Tensor:
    """np.percentile(x[i], pct) for every row of a [B, n] float32 / float64 device tensor (rows must be
    contiguous and 16-byte aligned: n * itemsize % 16 == 0, or B == 1); one set of launches for all rows.
    wide: as for percentile()."""
    x = require_device(x)
    if x.dtype not in (torch.float32, torch.float64) or x.dim() != 2:
        raise ValueError("percentile_batch() takes a 2-D float32 or float64 tensor")
    x = x.contiguous()
    B, n = x.shape
    if n == 0:
        raise IndexError("percentile of an empty array")
    if (n * x.element_size()) % 16 and B > 1:
        return torch.stack([percentile(x[i], pct, wide=wide) for i in range(B)])
    code, rdt = _sel_dtype(x.dtype, wide)
    L = _bind()
    if x.data_ptr() % 16:
        x = x.clone()
    out = torch.empty(B, dtype=rdt, device=x.device)
    ptrs = (ctypes.c_void_p * B)(*[x.data_ptr() + i * n * x.element_size() for i in range(B)])
    with torch.cuda.device(x.device):
        ws, stream = workspace(x.device, L.mq_percentile_batch_workspace_bytes(min(B, 64)))
        st = L.mq_percentile_batch(ptrs, B, code, n, float(pct), out.data_ptr(), None, ws.data_ptr(),
                                   ws.numel(), stream.cuda_stream)
    _lib.check(st)
    # np.percentile returns NaN for a row that holds one (the select would rank it as the largest value)
    return torch.where(torch.isnan(x).any(dim=1), torch.full_like(out, float("nan")), out)


@torch.no_grad()
def percentile_multi(xs: Sequence[torch.Tensor], pct: float = 99.0, nan_check: bool = True,
                     wide: bool = False, beside: bool = False) -> torch.Tensor:
    """np.percentile(x, pct) for each of a list of non-negative float32 / float64 device tensors of one dtype
    (any lengths; each flattened; unaligned views are copied): one set of launches per 64 arrays.  nan_check:
    as for percentile() (one small reduction per array; callers that already know which arrays hold a NaN --
    e.g. from their means -- pass False and patch the results themselves).  wide: as for percentile().
    beside: the call runs on a stream of its own next to a kernel that fills the chip -- the same results from
    256-thread workgroups (mq_percentile_multi_beside)."""
    if not xs:
        raise ValueError("percentile_multi() needs at least one array")
    dtype, dev = xs[0].dtype, xs[0].device
    if dtype not in (torch.float32, torch.float64):
        raise ValueError("percentile_multi() takes float32 or float64 tensors")
    code, rdt = _sel_dtype(dtype, wide)
    flat = []
    for x in xs:
        if x.dtype != dtype or x.device != dev or not x.is_cuda:
            raise ValueError("percentile_multi(): one dtype, one ROCm device")
        if x.numel() == 0:
            raise IndexError("percentile of an empty array")
        x = x.contiguous().reshape(-1)
        if x.data_ptr() % 16:
            x = x.clone()
        flat.append(x)
    B = len(flat)
    has_nan = torch.stack([torch.isnan(x).any() for x in flat]) if nan_check else None
    L = _bind()
    out = torch.empty(B, dtype=rdt, device=dev)
    ptrs = (ctypes.c_void_p * B)(*[x.data_ptr() for x in flat])
    ns = (ctypes.c_int64 * B)(*[x.numel() for x in flat])
    with torch.cuda.device(dev):
        ws, stream = workspace(dev, L.mq_percentile_batch_workspace_bytes(min(B, 64)))
        fn = L.mq_percentile_multi_beside if beside else L.mq_percentile_multi
        st = fn(ptrs, ns, B, code, float(pct), out.data_ptr(), None, ws.data_ptr(), ws.numel(), stream.cuda_stream)
    _lib.check(st)
    if has_nan is not None:
        out = torch.where(has_nan, torch.full_like(out, float("nan")), out)
    return out


@torch.no_grad()
def percentile_rows(mats: Sequence[torch.Tensor], pct: float = 99.0, wide: bool = False) -> torch.Tensor:
    """np.percentile(row, pct) for every row of every 2-D tensor of `mats` (one dtype; rows contiguous, 16-byte
    aligned, possibly padded: only stride(1) == 1 is required), results concatenated in order.  percentile_multi
    without a tensor object per row: a model sweep selects from hundreds of per-layer lists per pass.  No NaN
    check (as percentile_multi(nan_check=False)): a NaN is ranked as the largest value, where np.percentile
    returns NaN for the whole row -- callers patch rows they know to hold one (a NaN mean)."""
    if not mats:
        raise ValueError("percentile_rows() needs at least one tensor")
    dtype, dev = mats[0].dtype, mats[0].device
    code, rdt = _sel_dtype(dtype, wide)
    ptrs, ns = [], []
    for m in mats:
        if m.dim() != 2 or m.dtype != dtype or m.device != dev or not m.is_cuda or m.stride(1) != 1:
            raise ValueError("percentile_rows(): 2-D ROCm tensors of one dtype with contiguous rows")
        if m.shape[1] == 0:
            raise IndexError("percentile of an empty array")
        base, step = m.data_ptr(), m.stride(0) * m.element_size()
        if base % 16 or (step % 16 and m.shape[0] > 1):
            raise ValueError("percentile_rows(): rows must start on 16-byte boundaries")
        ptrs += [base + i * step for i in range(m.shape[0])]
        ns += [m.shape[1]] * m.shape[0]
    B = len(ptrs)
    L = _bind()
    out = torch.empty(B, dtype=rdt, device=dev)
    with torch.cuda.device(dev):
        ws, stream = workspace(dev, L.mq_percentile_batch_workspace_bytes(min(B, 64)))
        st = L.mq_percentile_multi((ctypes.c_void_p * B)(*ptrs), (ctypes.c_int64 * B)(*ns), B, code, float(pct),
                                   out.data_ptr(), None, ws.data_ptr(), ws.numel(), stream.cuda_stream)
    _lib.check(st)
    return out


def true_div(x: torch.Tensor, d) -> torch.Tensor:
    """x / d as an IEEE division per element.  (torch turns `tensor / python_scalar` on the device into a
    multiplication by the rounded reciprocal, which is off by one ulp for some elements.)"""
    return x / torch.full((), d, dtype=x.dtype, device=x.device)


@torch.no_grad()
def vector_mse_stats(seg_sse: torch.Tensor, seg_len: int, pct: float = 99.0):
    """(mean, percentile) of the per-vector MSE list seg_sse[i, :] / seg_len for every row i of a [B, nseg]
    float64 tensor of squared-error sums -- the `mse_rep` / `mse_p99` pair of
    quant_error_multi_weight.py:84-85 and quant_error_single_weight.py:78-79.

    For a power-of-two seg_len the division is an exponent shift, which commutes exactly with the order
    statistics and with numpy's interpolation, so the select runs on the sums themselves and only the B
    results are divided; any other length divides the whole list first, as the scripts do."""
    if seg_sse.dim() != 2 or seg_sse.dtype != torch.float64:
        raise ValueError("vector_mse_stats() takes a [B, nseg] float64 tensor")
    L = int(seg_len)
    B, nseg = seg_sse.shape

    def row_mean(t):  # mean of equal-sized chunk means: many workgroups per row instead of one
        if nseg % 256 == 0 and nseg >= 65536:
            return t.reshape(B * 256, nseg // 256).mean(dim=1).view(B, 256).mean(dim=1)
        return t.mean(dim=1)

    if L > 0 and (L & (L - 1)) == 0:
        return row_mean(seg_sse) / L, percentile_batch(seg_sse, pct) / L  # exponent shifts: exact
    per_vec = true_div(seg_sse, L)
    return row_mean(per_vec), percentile_batch(per_vec, pct)


@torch.no_grad()
def error_stats(a: torch.Tensor, b: torch.Tensor, seg_len: Optional[int] = None, abs_err=None,
                cols: bool = False, totals: bool = True):
    """One pass over (a, b): dict with 'totals' ([sum err^2, sum |err|], float64), 'seg_sse' (row-form sums
    over runs of seg_len), 'col_sse' (per-column sums), 'abs_err' (float32, or float16 arithmetic for
    float16 inputs)."""
    a, b = require_device(a), require_device(b).to(a.device)
    if a.shape != b.shape:
        raise ValueError("shape mismatch")
    if a.dtype != b.dtype or a.dtype not in (torch.float16, torch.float32):
        a, b = a.float(), b.float()  # the scripts widen everything to float32 (proto_to_array :35-38)
    a, b = a.contiguous(), b.contiguous()
    ncols = a.shape[-1] if a.dim() else 1
    nrows = a.numel() // ncols if ncols else 0
    dev = a.device
    res = {}
    if totals:
        res["totals"] = torch.zeros(2, dtype=torch.float64, device=dev)
    if seg_len:
        if a.numel() % seg_len:
            raise RuntimeError("group size mismatch")
        res["seg_sse"] = torch.empty(a.numel() // seg_len, dtype=torch.float64, device=dev)
    if cols:
        res["col_sse"] = torch.empty(ncols, dtype=torch.float64, device=dev)
    if abs_err is not None:
        res["abs_err"] = torch.empty(a.shape, dtype=abs_err, device=dev)
    if a.numel() == 0:
        return res
    L = _bind()
    with torch.cuda.device(dev):
        ws, stream = workspace(dev, L.mq_error_workspace_bytes(nrows, ncols, int(cols)))
        st = L.mq_error_stats(a.data_ptr(), b.data_ptr(), _DT[a.dtype], nrows, ncols, int(seg_len or 0),
                              _ptr(res.get("abs_err")), _lib.F16 if abs_err == torch.float16 else _lib.F32,
                              _ptr(res.get("seg_sse")), _ptr(res.get("col_sse")), _ptr(res.get("totals")),
                              ws.data_ptr(), ws.numel(), stream.cuda_stream)
    _lib.check(st)
    return res


def compute_mse(a, b) -> float:
    """float(np.mean((a32 - b32)**2)) -- quant_error_multi_weight.py:55-58, quant_error_single_weight.py:46-48."""
    st = error_stats(a, b)
    n = a.numel()
    return float(st["totals"][0].item() / n) if n else float("nan")


def slice_means_numpy(sq: torch.Tensor, offsets: Sequence[int], n: int, run: int, stride: int,
                      sums: bool = False) -> torch.Tensor:
    """np.mean of slices of the contiguous float32 tensor `sq`, bit for bit (mq_slice_mean_numpy_f32): slice i starts
    at element offsets[i] and holds n values in runs of `run` contiguous ones, `stride` elements apart -- a column of
    a [K, C] matrix: n = K, run = 1, stride = C; a block of gs columns: n = K * gs, run = gs; a row: run = n = C.
    sums: np.sum instead (mq_slice_sum_numpy_f32).  -> float32 [len(offsets)]."""
    from .fused import bind_batch
    L = bind_batch()
    if sq.dtype != torch.float32 or not sq.is_contiguous():
        raise ValueError("slice_means_numpy takes a contiguous float32 tensor")
    out = torch.empty(len(offsets), dtype=torch.float32, device=sq.device)
    base, ob = sq.data_ptr(), out.data_ptr()
    firsts = (ctypes.c_void_p * len(offsets))(*[base + 4 * int(o) for o in offsets])
    outs = (ctypes.c_void_p * len(offsets))(*[ob + 4 * i for i in range(len(offsets))])
    fn = L.mq_slice_sum_numpy_f32 if sums else L.mq_slice_mean_numpy_f32
    _lib.check(fn(firsts, len(offsets), n, run, stride, outs, torch.cuda.current_stream(sq.device).cuda_stream))
    return out


def _compute_errors_numpy(a, b, grouping, gs, form, keep_per_vec):
    """compute_errors in numpy's bits: the float32 squared errors as the scripts form them, every mean summed in
    numpy's order (slice_means_numpy; the list's float64 mean: report.numpy_list_means)."""
    from .report import numpy_list_means
    d = a.float() - b.float()            # proto_to_array: float32 (quant_error_multi_weight.py:35-38)
    sq = (d * d).contiguous()            # (a32 - b32) ** 2: float32 products
    n = sq.numel()
    result = {"mse_tensor": float(slice_means_numpy(sq.reshape(-1), [0], n, n, n)[0].item())}
    if a.dim() == 2 and grouping in ("per-channel", "per-group"):
        K, C = a.shape
        if grouping == "per-group":
            assert gs and C % gs == 0, "group size mismatch"
        if form == "multi":     # columns / column blocks, walked in numpy's C order
            if grouping == "per-channel":
                vec = slice_means_numpy(sq, range(C), K, 1, C)
            else:
                vec = slice_means_numpy(sq, range(0, C, gs), K * gs, gs, C)
        else:                   # rows / (row, group) runs: contiguous
            L = C if grouping == "per-channel" else gs
            vec = slice_means_numpy(sq, range(0, K * C, L), L, L, L)
        result["mse_rep"] = float(numpy_list_means([vec])[0].item())
        result["mse_p99"] = float(percentile(vec, 99, wide=True).item())
        if keep_per_vec or form == "single":
            result["per_vec"] = vec
    return result


def compute_errors(a, b, grouping: str, gs: Optional[int], form: str = "multi", keep_per_vec: bool = False,
                   numpy_order: bool = False):
    """mse_tensor, and for 2-D inputs the per-vector MSE list's mean (mse_rep) and P99 (mse_p99).

    form="multi":  vectors are COLUMNS (per-channel) or column blocks of gs columns (per-group)
                   -- quant_error_multi_weight.py:61-87
    form="single": vectors are ROWS (per-channel) or (row, group) runs in row-major order (per-group)
                   -- quant_error_single_weight.py:50-79 (without its plt.hist side effect)
    numpy_order:   every mean is summed in numpy's order (its 8192-element chunks and pairwise split, over the slice
                   in the order numpy walks it) and the three numbers have the scripts' BITS; a correctness mode that
                   materialises the float32 squared errors and spends one workgroup per vector.  Default: float64
                   sums in one pass over the two tensors, equal to the scripts' numbers to 1e-6.
    """
    a = require_device(a)
    b = require_device(b)
    if numpy_order:
        return _compute_errors_numpy(a, b, grouping, gs, form, keep_per_vec)
    K, C = (a.shape if a.dim() == 2 else (0, 0))
    per_vec = None
    if a.dim() == 2 and grouping in ("per-channel", "per-group"):
        if grouping == "per-group":
            assert gs and C % gs == 0, "group size mismatch"
        if form == "multi":
            st = error_stats(a, b, cols=True)
            col = st["col_sse"]
            per_vec = true_div(col, K) if grouping == "per-channel" else true_div(col.reshape(C // gs, gs).sum(dim=1), K * gs)
        else:
            L = C if grouping == "per-channel" else gs
            st = error_stats(a, b, seg_len=L)
            per_vec = true_div(st["seg_sse"], L)
    else:
        st = error_stats(a, b)
    result = {"mse_tensor": float(st["totals"][0].item() / a.numel())}
    if per_vec is not None:
        result["mse_rep"] = float(per_vec.mean().item())
        result["mse_p99"] = float(percentile(per_vec, 99).item())
        if keep_per_vec or form == "single":
            result["per_vec"] = per_vec
    return result


class QuantizationErrorAnalyzer:
    """The statistics half of elwise_error/onnx_error_cli.py:73-127 on two tensors (the ONNX loading and
    the matplotlib plots stay on the CPU side): float32 element-wise error, its absolute value, P99 and
    sum, and MSE / absolute-error sum of an index range."""

    def __init__(self, w_fp16: torch.Tensor, w_quant: torch.Tensor, analysis_range: Optional[tuple] = None,
                 numpy_order: bool = False):
        """numpy_order: the sums and the sliced mean in numpy's float32 summation order -- get_error_sum,
        get_sliced_mse and get_sliced_abs_error_sum then return the script's float32 values bit for bit (default:
        float64 sums, equal to 1e-6)."""
        self.numpy_order = bool(numpy_order)
        self.w_fp16 = require_device(w_fp16).float().reshape(-1)   # _load_weight: .astype(np.float32) (:110)
        self.w_quant = require_device(w_quant).float().reshape(-1)
        st = error_stats(self.w_fp16, self.w_quant, abs_err=torch.float32)
        self.abs_elementwise_error = st["abs_err"]
        self._totals = st["totals"]
        self.analysis_range = analysis_range
        self._sliced = None
        if analysis_range:
            start, end = analysis_range
            end = min(end, self.w_fp16.numel())
            self._sliced = error_stats(self.w_fp16[start:end], self.w_quant[start:end])
            self._sliced_n = end - start

    @property
    def elementwise_error(self):
        return self.w_fp16 - self.w_quant

    def get_p99_error(self):
        """np.percentile(abs_err, 99) (:113-114): float32, bit-exact."""
        return percentile(self.abs_elementwise_error, 99)

    def _range(self):
        start, end = self.analysis_range
        return start, min(end, self.w_fp16.numel())

    def get_error_sum(self) -> float:
        """np.sum(abs_err) (:116-117); accumulated in float64 here, or -- numpy_order -- in numpy's float32 order."""
        if self.numpy_order:
            x = self.abs_elementwise_error.contiguous()
            return float(slice_means_numpy(x, [0], x.numel(), x.numel(), x.numel(), sums=True)[0].item())
        return float(self._totals[1].item())

    def get_sliced_mse(self) -> float:
        """np.mean(sliced_error ** 2) (:119-123)."""
        if self.numpy_order and self._sliced:
            start, end = self._range()
            d = (self.w_fp16[start:end] - self.w_quant[start:end])
            sq = (d * d).contiguous()
            return float(slice_means_numpy(sq, [0], sq.numel(), sq.numel(), sq.numel())[0].item())
        return float(self._sliced["totals"][0].item() / self._sliced_n) if self._sliced else 0.0

    def get_sliced_abs_error_sum(self) -> float:
        """np.sum(np.abs(sliced_error)) (:125-127)."""
        if self.numpy_order and self._sliced:
            start, end = self._range()
            x = self.abs_elementwise_error[start:end].contiguous()
            return float(slice_means_numpy(x, [0], x.numel(), x.numel(), x.numel(), sums=True)[0].item())
        return float(self._sliced["totals"][1].item()) if self._sliced else 0.0

    def abs_error_histogram(self, bins: int = 200, range=None):
        """What _plot_histogram's plt.hist(abs_err, bins=200) draws (:208-215, :273-277): np.histogram counts and
        edges of the absolute error, counted on the device."""
        return histogram(self.abs_elementwise_error, bins=bins, range=range)


@torch.no_grad()
def bucketize(w_fp16: torch.Tensor, w_q: torch.Tensor, step: float = 0.0001, max_value: float = 0.0007):
    """single_weight_quant_error.py:39-52: float16 |w - w_q|, bins arange(step, max+step, step),
    idx = min(np.digitize(abs, bins, right=True), len(bins)-1), value = idx * step.
    Returns (abs_err float16, idx uint8, values float64 on the host for the bucket levels)."""
    a = require_device(w_fp16).half().contiguous().reshape(-1)
    b = require_device(w_q).half().contiguous().reshape(-1).to(a.device)
    abs_err = error_stats(a, b, abs_err=torch.float16, totals=False)["abs_err"]
    bins = np.arange(step, max_value + step, step)
    values = np.arange(len(bins)) * step
    if len(bins) > 64:
        raise ValueError("at most 64 buckets")
    idx = torch.empty(a.numel(), dtype=torch.uint8, device=a.device)
    L = _bind()
    with torch.cuda.device(a.device):
        st = L.mq_bucketize_f16(abs_err.data_ptr(), a.numel(), bins.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                len(bins), idx.data_ptr(), torch.cuda.current_stream(a.device).cuda_stream)
    _lib.check(st)
    return abs_err, idx, values


@torch.no_grad()
def compare_formats(w_fp16: torch.Tensor, w_qs: Sequence[torch.Tensor], threshold: float = 0.02,
                    out: Optional[torch.Tensor] = None):
    """compare.py:99-125: per-format float16 |w - w_q|, element-wise minimum, outlier mask (min > threshold)
    and per-format "is best" masks.  Returns dict(min_err float16, mask_out bool, best list of bool, flags).
    With `out` (uint8, one byte per element) only the packed flags are produced, into it: bit 7 = outlier,
    bit f = format f is best."""
    if not 1 <= len(w_qs) <= 7:
        raise ValueError("1..7 formats")
    w = require_device(w_fp16).half().contiguous().reshape(-1)
    qs = [require_device(q).half().contiguous().reshape(-1).to(w.device) for q in w_qs]
    n = w.numel()
    if out is not None:
        if out.dtype != torch.uint8 or out.numel() != n or not out.is_contiguous() or out.device != w.device:
            raise ValueError("out must be a contiguous uint8 device tensor with one byte per element")
        arr = (ctypes.c_void_p * len(qs))(*[q.data_ptr() for q in qs])
        with torch.cuda.device(w.device):
            st = _bind().mq_compare_formats_f16(w.data_ptr(), arr, len(qs), n, int(np.float16(threshold).view(np.uint16)),
                                                out.data_ptr(), None,
                                                torch.cuda.current_stream(w.device).cuda_stream)
        _lib.check(st)
        return {"flags": out}
    flags = torch.empty(n, dtype=torch.uint8, device=w.device)
    min_err = torch.empty(n, dtype=torch.float16, device=w.device)
    arr = (ctypes.c_void_p * len(qs))(*[q.data_ptr() for q in qs])
    thr_bits = int(np.float16(threshold).view(np.uint16))
    L = _bind()
    with torch.cuda.device(w.device):
        st = L.mq_compare_formats_f16(w.data_ptr(), arr, len(qs), n, thr_bits, flags.data_ptr(), min_err.data_ptr(),
                                      torch.cuda.current_stream(w.device).cuda_stream)
    _lib.check(st)
    return {"min_err": min_err, "mask_out": (flags & 0x80) != 0,
            "best": [(flags & (1 << i)) != 0 for i in range(len(qs))], "flags": flags}


@torch.no_grad()
def to_ranges(mask: torch.Tensor) -> List[tuple]:
    """compare.py:130-143 for a boolean mask: [(start, length), ...] of its runs of True, computed with
    device-side differences instead of a Python loop over indices."""
    m = mask.reshape(-1).to(torch.int8)
    if m.numel() == 0:
        return []
    d = torch.diff(m, prepend=m.new_zeros(1), append=m.new_zeros(1))
    starts = torch.nonzero(d == 1).reshape(-1)
    ends = torch.nonzero(d == -1).reshape(-1)
    return list(zip(starts.tolist(), (ends - starts).tolist()))


# --------------------------------------------------------------------------- histograms (plot feeders)

@torch.no_grad()
def minmax(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(min, max) of a float16 / float32 / float64 device tensor as a float64 device tensor [2] (NaN when x
    holds one, like np.min / np.max).  Passing the result of an earlier call as `out` folds it in: the running
    global range of plot_matmul_weight_rev0.py:53-73."""
    x = require_device(x)
    if x.dtype not in _DT:
        raise ValueError("minmax() takes float16, float32 or float64")
    x = x.contiguous().reshape(-1)
    if x.numel() == 0 and out is None:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    L = _bind()
    merge = out is not None
    if out is None:
        out = torch.empty(2, dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        ws, stream = workspace(x.device, L.mq_minmax_workspace_bytes())
        st = L.mq_minmax(x.data_ptr(), _DT[x.dtype], x.numel(), int(merge), out.data_ptr(), ws.data_ptr(), ws.numel(),
                         stream.cuda_stream)
    _lib.check(st)
    return out


def _np_dtype(t: torch.Tensor):
    # the reference widens float16 initializers to float32 before every histogram (proto_to_array)
    return np.float64 if t.dtype == torch.float64 else np.float32


@torch.no_grad()
def histogram(x: torch.Tensor, bins=10, range=None, counts: Optional[torch.Tensor] = None):
    """np.histogram(x, bins, range) for a device tensor: (counts int64 device tensor [nbins], edges ndarray).

    bins: an int (uniform bins over `range`, default the array's own min / max -- found by a first scan, as
    plt.hist(data, bins=200) does in onnx_error_cli.py:211 and quant_error_single_weight.py:81, :122) or an
    ascending edge array (plot_matmul_weight_rev0.py:98).  The edges are numpy's own (np.histogram_bin_edges
    on the two extreme values, in the array's dtype; float16 counts as its float32 widening); the device
    decides every count by np.histogram's comparison rule, so the counts equal numpy's.  `counts` (int64 [nbins])
    is accumulated into when given."""
    x = require_device(x)
    if x.dtype not in _DT:
        raise ValueError("histogram() takes float16, float32 or float64")
    x = x.contiguous().reshape(-1)
    if np.ndim(bins) == 0:
        if range is None and x.numel():
            lo, hi = minmax(x).tolist()
            if not (np.isfinite(lo) and np.isfinite(hi)):
                raise ValueError(f"autodetected range of [{lo}, {hi}] is not finite")  # numpy's own check
            ends = np.array([lo, hi], dtype=_np_dtype(x))
        else:
            ends = np.array([], dtype=_np_dtype(x))
        edges = np.histogram_bin_edges(ends, bins=bins, range=range)
    else:
        edges = np.asarray(bins)
        if edges.ndim != 1 or edges.size < 2 or np.any(edges[:-1] > edges[1:]):
            raise ValueError("`bins` must be 1d and increase monotonically")
    nbins = edges.size - 1
    if nbins > 4096:
        raise ValueError("at most 4096 bins")
    if counts is None:
        counts = torch.zeros(nbins, dtype=torch.int64, device=x.device)
    elif counts.shape != (nbins,) or counts.dtype != torch.int64 or not counts.is_contiguous():
        raise ValueError("counts must be a contiguous int64 tensor with one entry per bin")
    L = _bind()
    dev_edges = torch.from_numpy(edges.astype(np.float64)).to(x.device)
    with torch.cuda.device(x.device):
        st = L.mq_histogram(x.data_ptr(), _DT[x.dtype], x.numel(), dev_edges.data_ptr(), nbins, counts.data_ptr(),
                            torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(st)
    return counts, edges


@torch.no_grad()
def weight_histogram(tensors: Sequence[torch.Tensor], bins: int = 200):
    """The two-pass histogram over all weights of a model (plot_matmul_weight_rev0.py:51-101, :221-238): global
    min / max first, np.linspace(gmin, gmax, bins + 1) as edges, then the counts of every tensor added up.
    -> (counts int64 device tensor, edges ndarray)."""
    tensors = [require_device(t) for t in tensors]
    if not tensors:
        raise ValueError("weight_histogram() needs at least one tensor")
    rng = None
    for t in tensors:
        rng = minmax(t, rng)
    gmin, gmax = (np.float32(v) for v in rng.tolist())   # the script widens every initializer to float32
    edges = np.linspace(gmin, gmax, bins + 1)
    counts = None
    for t in tensors:
        counts, _ = histogram(t, edges, counts=counts)
    return counts, edges
