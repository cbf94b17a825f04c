"""Host side of the fused quantize -> dequantize -> error pass (one call = one mq_quantize launch).

`quantize()` is the superset entry point; quantization_amd.quant_weight wraps it in the reference's
function signatures.  Tensors stay on the ROCm device; nothing here synchronises with the host.
"""
import ctypes
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib

# name -> (exp_bits, man_bits, bias) of the value-table formats that have a closed form
# (reference: quant_weight.py).  Without subnormals: +-(1 + m/2^M) * 2^(e - bias) over ALL exponent codes, the
# (e=0, m=0) code being the single zero: FP8_E4M3 :114-138, FP8_E3M4 :91-113, FP8_E2M5 :65-90.
# With subnormals (LUT_SUBNORMAL): +-m * 2^(1-bias-M) for e = 0: FP4_E2M1 :22, FP5_E2M2 :38, FP5_E3M1 :40,
# FP6_E2M3 :50, FP6_E3M2 :56; exp_bits 0 = the uniform grid alone, i.e. the INTn lists used as tables
# (:10, :18, :34, :44, :64) when quant_datatype is called with an 'int*' name.
LUT_FORMATS = {"fp8_e4m3": (4, 3, 7), "fp8_e3m4": (3, 4, 3), "fp8_e2m5": (2, 5, 1),
               "fp4": (2, 1, 0), "fp5": (2, 2, -1), "fp5_e2m2": (2, 2, -1), "fp5_e3m1": (3, 1, 0),
               "fp6": (2, 3, -2), "fp6_e2m3": (2, 3, -2), "fp6_e3m2": (3, 2, -1),
               "int3": (0, 2, -1), "int4": (0, 3, -2), "int5": (0, 4, -3), "int6": (0, 5, -4), "int8": (0, 7, -6)}
LUT_SUBNORMAL = {"fp4", "fp5", "fp5_e2m2", "fp5_e3m1", "fp6", "fp6_e2m3", "fp6_e3m2",
                 "int3", "int4", "int5", "int6", "int8"}



def lut_values(datatype: str):
    """The value list of a closed-form table format (LUT_FORMATS), ascending -- what the reference spells out as a
    literal list (quant_weight.py:9-138); pinned against those lists by tests/golden/lut_tables.json."""
    E, M, bias = LUT_FORMATS[datatype]
    pos = []
    if datatype in LUT_SUBNORMAL:
        pos += [m * 2.0 ** (1 - bias - M) for m in range(1, 2 ** M)]
        pos += [(1 + m / 2 ** M) * 2.0 ** (e - bias) for e in range(1, 2 ** E) for m in range(2 ** M)]
        if E == 0:   # the uniform grid alone reaches 2^M steps (the INTn lists: 1 .. 2^(bits-1) - 1 after scaling)
            pos = [m * 2.0 ** (1 - bias - M) for m in range(1, 2 ** M)]
    else:
        pos = [(1 + m / 2 ** M) * 2.0 ** (e - bias) for e in range(2 ** E) for m in range(2 ** M) if e or m]
    return _mirrored(sorted(pos))


def _mirrored(positive):
    return [-v for v in reversed(positive)] + [0.0] + [float(v) for v in positive]


def _with(values, extra):
    return sorted(values + [float(extra)])


_FP3 = _mirrored([1, 2, 4])
_FP4 = _mirrored([1, 2, 3, 4, 6, 8, 12])
# Value lists without a closed form (reference: quant_weight.py FP3 :11, FP3_ER_* / FP3_EA_* :12-15 = FP3 plus
# one value, FLINT4 :20, FP4_ER_* / FP4_EA_* :24-31 = FP4_E2M1 plus one value, FLINT5 :36), quantized by
# mq_quantize_table.  tests/golden/lut_tables.json pins them to the reference's lists.
TABLE_LISTS = {
    "fp3": _FP3,
    "fp3_er_pos": _with(_FP3, 3), "fp3_er_neg": _with(_FP3, -3),
    "fp3_ea_pos": _with(_FP3, 6), "fp3_ea_neg": _with(_FP3, -6),
    "flint4": _mirrored([1, 2, 3, 4, 6, 8, 16]),
    "fp4_er_pos": _with(_FP4, 10), "fp4_er_neg": _with(_FP4, -10),
    "fp4_ea_pos": _with(_FP4, 16), "fp4_ea_neg": _with(_FP4, -16),
    "flint5": _mirrored([1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 24, 32, 64]),
    # FP8_E5M2 :139-169: +-(1 + m/4) * 2^(e-15) over all 32 exponent codes, (e=0, m=0) being the zero; its
    # path runs in float32 (:505-540)
    "fp8_e5m2": _mirrored([(1 + m / 4) * 2.0 ** (e - 15) for e in range(32) for m in range(4) if e or m]),
}
# quant_mx's tables (quant_weight.py:177-188)
MX_TABLES = {"mx_int3": _mirrored([1, 2, 3]), "mx_fp3": _FP3,
             "mx_int4": _mirrored([1, 2, 3, 4, 5, 6, 7]), "mx_fp4": _FP4}

_DTYPES = {torch.float16: _lib.F16, torch.bfloat16: _lib.BF16, torch.float32: _lib.F32}
_workspaces = {}


@dataclass
class QuantResult:
    rows: int
    cols: int
    nseg: int
    deq: Optional[torch.Tensor] = None      # float16 [rows, cols]
    codes: Optional[torch.Tensor] = None    # uint8 [rows, cols] (INT) or float16 [rows, cols] (LUT)
    scale: Optional[torch.Tensor] = None    # float16 [nseg]
    zero_point: Optional[torch.Tensor] = None
    abs_err: Optional[torch.Tensor] = None  # float32 or float16 [rows, cols]
    seg_sse: Optional[torch.Tensor] = None  # float64 [nseg]: sums of squared errors per segment
    seg_mse: Optional[torch.Tensor] = None  # float32 [nseg]: float32(sum / segment length), instead of seg_sse
    totals: Optional[torch.Tensor] = None   # float64 [2]: sum err^2, sum |err|
    abs_p99: Optional[torch.Tensor] = None  # float32 [B]: percentile of the float32 |error| (quantize_batch(p99=))
    list_p99: Optional[torch.Tensor] = None  # float64 [B]: percentile of the seg_mse list (quantize_batch_passes)


def require_device(t: torch.Tensor) -> torch.Tensor:
    """ROCm tensors pass through; a host tensor is copied to the current ROCm device (there is no CPU
    compute path)."""
    if t.is_cuda:
        return t
    if not torch.cuda.is_available():
        raise RuntimeError("quantization_amd needs a ROCm device (MI355X); no CPU fallback exists")
    return t.cuda()


def workspace(device, nbytes):
    stream = torch.cuda.current_stream(device)
    key = (device.index, stream.cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws, stream


def make_cfg(fmt, bits=8, group_size=None, datatype=None, f32_math=False):
    gs = 0 if group_size is None else int(group_size)
    if gs < -1:
        gs = 0
    cfg = _lib.QuantCfg(format=fmt, bits=int(bits), group_size=gs, f32_math=int(bool(f32_math)))
    if fmt == _lib.LUT_FP:
        cfg.exp_bits, cfg.man_bits, cfg.exp_bias = LUT_FORMATS[datatype]
        if datatype in LUT_SUBNORMAL:
            cfg.format = _lib.LUT_FP_SUB
    return cfg


def _ptr(t):
    return None if t is None else t.data_ptr()


def _seg_flag(seg_mse):
    """mq_quant_out.seg_mse_f32: 0 float64 sums, 1 float32 MSEs, 2 float32 MSEs with numpy's summation order."""
    if seg_mse == "numpy":
        return 2
    if seg_mse in (True, False, None, 0, 1):
        return int(bool(seg_mse))
    raise ValueError("seg_mse: False, True or 'numpy'")


@torch.no_grad()
def quantize(w: torch.Tensor, fmt: int, bits: int = 8, group_size=None, datatype: Optional[str] = None, *,
             f32_math=False, deq=True, codes=True, scale=True, abs_err=None, seg_sse=False, seg_mse=False,
             totals=False, out: Optional[QuantResult] = None) -> QuantResult:
    """w: ROCm tensor viewed as [rows, cols] = [prod(shape[:-1]), shape[-1]], float16 / bfloat16 / float32.

    abs_err: None, torch.float32 or torch.float16.  seg_sse: float64 sums of squared errors per segment; seg_mse:
    the float32 mean squared error per segment instead (the per-vector MSE list of
    quant_error_single_weight.py:56-71, whose entries are float32 means) -- one of the two.  seg_mse="numpy": the
    entries are summed in numpy's order and equal np.mean(diff * diff) bit for bit (statistics-only calls on float16 /
    bfloat16 weights, segments of a multiple of 8 and at most 32768 elements; ValueError otherwise).  `out` may carry
    preallocated tensors to reuse.
    """
    if seg_sse and seg_mse:
        raise ValueError("seg_sse and seg_mse share one output slot: ask for one of them")
    if w.dtype not in _DTYPES:
        raise ValueError(f"unsupported weight dtype {w.dtype}")
    if not w.is_cuda:
        raise RuntimeError("quantize() expects a ROCm device tensor")
    cols = w.shape[-1] if w.dim() else 1
    rows = w.numel() // cols if cols else 0
    w2 = w.contiguous()
    cfg = make_cfg(fmt, bits, group_size, datatype, f32_math)
    gs = cfg.group_size
    n = rows * cols
    L = n if gs == -1 else (cols if gs <= 0 else gs)
    if gs > 0 and cols % gs:
        raise RuntimeError(f"shape '[{rows}, {cols // gs}, {gs}]' is invalid for input of size {n}")
    nseg = n // L if L else 0
    dev = w.device
    res = out if out is not None else QuantResult(rows, cols, nseg)
    res.rows, res.cols, res.nseg = rows, cols, nseg

    def buf(cur, want, shape, dtype):
        if not want:
            return None
        if cur is not None and cur.shape == torch.Size(shape) and cur.dtype == dtype and cur.device == dev:
            return cur
        return torch.empty(shape, dtype=dtype, device=dev)

    res.deq = buf(res.deq, deq, (rows, cols), torch.float16)
    res.codes = buf(res.codes, codes, (rows, cols), torch.float16 if fmt == _lib.LUT_FP else torch.uint8)
    res.scale = buf(res.scale, scale, (nseg,), torch.float16)
    res.zero_point = buf(res.zero_point, scale and fmt == _lib.INT_ASYM, (nseg,), torch.float16)
    res.abs_err = buf(res.abs_err, abs_err is not None, (rows, cols), abs_err or torch.float32)
    res.seg_sse = buf(res.seg_sse, seg_sse, (nseg,), torch.float64)
    res.seg_mse = buf(res.seg_mse, seg_mse, (nseg,), torch.float32)
    res.totals = buf(res.totals, totals, (2,), torch.float64)
    if n == 0:
        if res.totals is not None:
            res.totals.zero_()
        return res

    L_ = _lib.lib()
    o = _lib.QuantOut(deq=_ptr(res.deq), codes=_ptr(res.codes), scale=_ptr(res.scale),
                      zero_point=_ptr(res.zero_point), abs_err=_ptr(res.abs_err),
                      abs_err_dtype=_lib.F16 if abs_err == torch.float16 else _lib.F32,
                      seg_mse_f32=_seg_flag(seg_mse), seg_sse=_ptr(res.seg_mse if seg_mse else res.seg_sse),
                      totals=_ptr(res.totals))
    with torch.cuda.device(dev):
        ws, stream = workspace(dev, L_.mq_quantize_workspace_bytes(rows, cols, ctypes.byref(cfg)))
        st = L_.mq_quantize(w2.data_ptr(), _DTYPES[w.dtype], rows, cols, ctypes.byref(cfg), ctypes.byref(o),
                            ws.data_ptr(), ws.numel(), stream.cuda_stream)
    _lib.check(st)
    return res


class _QuantItem(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("out", _lib.QuantOut)]


class _MultiItem(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("list", ctypes.c_void_p * 4), ("totals", ctypes.c_void_p * 4)]


def bind_batch():
    """libmq.so with the batch entry points' ctypes signatures set."""
    L_ = _lib.lib()
    if not getattr(L_, "_batch_bound", False):
        L_.mq_quantize_batch_workspace_bytes.restype = ctypes.c_size_t
        L_.mq_quantize_batch_workspace_bytes.argtypes = [ctypes.c_int32, ctypes.c_int64, ctypes.c_int64,
                                                         ctypes.POINTER(_lib.QuantCfg)]
        L_.mq_quantize_batch.restype = ctypes.c_int
        L_.mq_quantize_batch.argtypes = [ctypes.POINTER(_QuantItem), ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                         ctypes.c_int64, ctypes.POINTER(_lib.QuantCfg), ctypes.c_void_p,
                                         ctypes.c_size_t, ctypes.c_void_p]
        L_.mq_quantize_p99_batch_workspace_bytes.restype = ctypes.c_size_t
        L_.mq_quantize_p99_batch_workspace_bytes.argtypes = L_.mq_quantize_batch_workspace_bytes.argtypes
        L_.mq_quantize_p99_batch.restype = ctypes.c_int
        L_.mq_quantize_p99_batch.argtypes = [ctypes.POINTER(_QuantItem), ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                             ctypes.c_int64, ctypes.POINTER(_lib.QuantCfg), ctypes.c_double,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        i32, i64, vp, sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t
        L_.mq_quantize_p99_passes_workspace_bytes.restype = sz
        L_.mq_quantize_p99_passes_workspace_bytes.argtypes = [i32, i32, i64, i64]
        L_.mq_quantize_p99_passes.restype = ctypes.c_int
        L_.mq_quantize_p99_passes.argtypes = [vp, i32, i32, i32, i64, i64, ctypes.c_double, ctypes.c_double, vp, sz, vp]
        L_.mq_sweep_multi_table_bytes.restype = sz
        L_.mq_sweep_multi_table_bytes.argtypes = [i32]
        L_.mq_sweep_multi_workspace_bytes.restype = sz
        L_.mq_sweep_multi_workspace_bytes.argtypes = [i64]
        L_.mq_sweep_multi_prepare.restype = ctypes.c_int
        L_.mq_sweep_multi_prepare.argtypes = [ctypes.POINTER(_MultiItem), ctypes.POINTER(i64), i64, i32, i32,
                                              ctypes.POINTER(_lib.QuantCfg), ctypes.c_uint32, vp, sz, ctypes.POINTER(i64)]
        L_.mq_sweep_multi_launch.restype = ctypes.c_int
        L_.mq_sweep_multi_launch.argtypes = [vp, i32, i64, i64, i32, ctypes.POINTER(_lib.QuantCfg), ctypes.c_uint32, vp, sz, vp]
        L_.mq_sweep_multi_finalize.restype = ctypes.c_int
        L_.mq_sweep_multi_finalize.argtypes = [vp, i32, i64, vp, sz, vp]
        L_.mq_sweep_multi_run.restype = ctypes.c_int
        L_.mq_sweep_multi_run.argtypes = [vp, i32, i64, i64, i32, ctypes.POINTER(_lib.QuantCfg), ctypes.c_uint32, vp, sz, vp]
        L_.mq_sweep_table_bytes.restype = sz
        L_.mq_sweep_table_bytes.argtypes = [i32]
        L_.mq_sweep_workspace_bytes.restype = sz
        L_.mq_sweep_workspace_bytes.argtypes = [i64]
        L_.mq_sweep_prepare.restype = ctypes.c_int
        L_.mq_sweep_prepare.argtypes = [ctypes.POINTER(_QuantItem), ctypes.POINTER(i64), ctypes.POINTER(i64), i32, i32,
                                        ctypes.POINTER(_lib.QuantCfg), vp, sz, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        L_.mq_sweep_run.restype = ctypes.c_int
        L_.mq_sweep_run.argtypes = [vp, i32, i64, i64, i32, ctypes.POINTER(_lib.QuantCfg), i32, vp, sz, vp]
        L_.mq_tensor_mse_numpy_workspace_bytes.restype = sz
        L_.mq_tensor_mse_numpy_workspace_bytes.argtypes = [i64, i64]
        L_.mq_tensor_mse_numpy.restype = ctypes.c_int
        L_.mq_tensor_mse_numpy.argtypes = [vp, i32, i64, i64, ctypes.POINTER(_lib.QuantCfg), vp, vp, vp, vp, sz, vp]
        L_.mq_list_mean_numpy.restype = ctypes.c_int
        L_.mq_list_mean_numpy.argtypes = [ctypes.POINTER(vp), i32, i64, ctypes.POINTER(vp), vp]
        L_.mq_slice_mean_numpy_f32.restype = ctypes.c_int
        L_.mq_slice_mean_numpy_f32.argtypes = [ctypes.POINTER(vp), i32, i64, i64, i64, ctypes.POINTER(vp), vp]
        L_.mq_slice_sum_numpy_f32.restype = ctypes.c_int
        L_.mq_slice_sum_numpy_f32.argtypes = L_.mq_slice_mean_numpy_f32.argtypes
        pi64 = ctypes.POINTER(i64)
        L_.mq_sweep_numpy_table_bytes.restype = sz
        L_.mq_sweep_numpy_table_bytes.argtypes = [i32]
        L_.mq_sweep_numpy_prepare.restype = ctypes.c_int
        L_.mq_sweep_numpy_prepare.argtypes = [ctypes.POINTER(_QuantItem), pi64, pi64, i32, i32, ctypes.POINTER(_lib.QuantCfg),
                                              vp, vp, sz, pi64, pi64, pi64, pi64]
        L_.mq_sweep_numpy_run.restype = ctypes.c_int
        L_.mq_sweep_numpy_run.argtypes = [vp, i32, i64, i64, i32, ctypes.POINTER(_lib.QuantCfg), vp, sz, vp]
        L_.mq_sweep_numpy_finish_workspace_bytes.restype = sz
        L_.mq_sweep_numpy_finish_workspace_bytes.argtypes = [i64, i64]
        L_.mq_sweep_numpy_finish.restype = ctypes.c_int
        L_.mq_sweep_numpy_finish.argtypes = [vp, i32, i64, vp, i64, i64, vp, vp, vp, sz, vp]
        psz = ctypes.POINTER(sz)
        L_.mq_sweep_feed_sizes.restype = ctypes.c_int
        L_.mq_sweep_feed_sizes.argtypes = [pi64, pi64, i32, ctypes.POINTER(_lib.QuantCfg), ctypes.c_double,
                                           ctypes.POINTER(i32), psz, psz]
        L_.mq_sweep_feed_prepare.restype = ctypes.c_int
        L_.mq_sweep_feed_prepare.argtypes = [ctypes.POINTER(_QuantItem), pi64, pi64, i32, i32, ctypes.POINTER(_lib.QuantCfg),
                                             ctypes.c_double, vp, sz, vp, vp, sz, vp, sz, pi64, pi64, pi64, ctypes.POINTER(i32),
                                             ctypes.POINTER(i32)]
        L_.mq_sweep_feed_sample.restype = ctypes.c_int
        L_.mq_sweep_feed_sample.argtypes = [vp, i32, i64, i64, i32, ctypes.POINTER(_lib.QuantCfg), vp, vp]
        L_.mq_sweep_feed_run.restype = ctypes.c_int
        L_.mq_sweep_feed_run.argtypes = [vp, i32, i64, i64, i32, ctypes.POINTER(_lib.QuantCfg), vp, sz, vp]
        L_.mq_sweep_feed_finish.restype = ctypes.c_int
        L_.mq_sweep_feed_finish.argtypes = [vp, i32, i64, i64, i32, ctypes.POINTER(_lib.QuantCfg), vp, vp, vp]
        L_.mq_sweep_records.restype = ctypes.c_int
        L_.mq_sweep_records.argtypes = [vp, vp, vp, vp, i32, vp, vp]
        L_.mq_sweep_numpy_groups_workspace_bytes.restype = ctypes.c_size_t
        L_.mq_sweep_numpy_groups_workspace_bytes.argtypes = [i64]
        L_.mq_sweep_numpy_run_groups.restype = ctypes.c_int
        L_.mq_sweep_numpy_run_groups.argtypes = [vp, vp, vp, i32, i64, i32, ctypes.POINTER(_lib.QuantCfg), vp, sz, vp]
        L_.mq_sweep_table.restype = ctypes.c_int
        L_.mq_sweep_table.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp]
        L_._batch_bound = True
    return L_


@torch.no_grad()
def quantize_batch(w: torch.Tensor, fmt: int, bits: int = 8, group_size=None, datatype: Optional[str] = None, *,
                   f32_math=False, deq=True, codes=True, scale=True, abs_err=None, seg_sse=False, seg_mse=False,
                   totals=False, p99: Optional[float] = None, out: Optional[QuantResult] = None) -> QuantResult:
    """quantize() for a stack of same-shaped weights w[B, rows, cols] (e.g. one projection of every layer):
    one fused launch per 32 tensors.  Every field of the result has a leading batch dimension.

    p99: a percentile (e.g. 99) of each tensor's float32 |error| array, selected exactly while the errors
    are produced (result.abs_p99, float32 [B]); needs abs_err=torch.float32."""
    res, items, cfg, (B, rows, cols, w0) = _batch_setup(w, fmt, bits, group_size, datatype, f32_math, deq, codes, scale,
                                                         abs_err, seg_sse, seg_mse, totals, p99, out)
    if items is None:
        return res
    dev = w0.device
    L_ = bind_batch()
    with torch.cuda.device(dev):
        if p99 is not None:
            ws, stream = workspace(dev, L_.mq_quantize_p99_batch_workspace_bytes(B, rows, cols, ctypes.byref(cfg)))
            st = L_.mq_quantize_p99_batch(items, B, _DTYPES[w0.dtype], rows, cols, ctypes.byref(cfg), float(p99),
                                          res.abs_p99.data_ptr(), ws.data_ptr(), ws.numel(), stream.cuda_stream)
        else:
            ws, stream = workspace(dev, L_.mq_quantize_batch_workspace_bytes(B, rows, cols, ctypes.byref(cfg)))
            st = L_.mq_quantize_batch(items, B, _DTYPES[w0.dtype], rows, cols, ctypes.byref(cfg), ws.data_ptr(),
                                      ws.numel(), stream.cuda_stream)
    _lib.check(st)
    return res


class _P99Pass(ctypes.Structure):
    _fields_ = [("items", ctypes.POINTER(_QuantItem)), ("cfg", ctypes.POINTER(_lib.QuantCfg)), ("p99", ctypes.c_void_p),
                ("list_p99", ctypes.c_void_p)]


@torch.no_grad()
def quantize_batch_passes(w, passes, p99: float = 99.0, list_p99: Optional[float] = None,
                          out: Optional[list] = None) -> list:
    """Several quantize_batch(..., p99=p99) passes over ONE stack of same-shaped weights -- the configurations of a
    sweep (mq_quantize_p99_passes): the same results as the separate calls, with the short selection kernels of each
    pass running beside the neighbouring full passes.

    passes: a list of dicts of quantize_batch's arguments (fmt, bits, group_size, datatype, deq, codes, scale, seg_mse,
    totals ...; abs_err is float32 in every pass).  list_p99: also the percentile of each tensor's per-segment MSE list
    (float64, np.percentile of the list as float64; every pass must then ask for seg_mse=True) -> result.list_p99.
    out: the list a previous call returned, to reuse its buffers.  -> one QuantResult per pass."""
    if not passes:
        return []
    if out is not None and len(out) != len(passes):
        raise ValueError("quantize_batch_passes(): `out` must hold one result per pass")
    # the same call again on the results of the last one: the item tables built then are still right
    key = ((w.data_ptr(), tuple(w.shape), w.dtype) if isinstance(w, torch.Tensor) else
           tuple((t.data_ptr(), tuple(t.shape), t.dtype) for t in w),
           tuple(tuple(sorted((a, str(b)) for a, b in ps.items())) for ps in passes), p99, list_p99)
    cached = getattr(out[0], "_passes_call", None) if out else None
    if cached is not None and cached[0] == key:
        _, arr, npass, B, dt, rows, cols, dev, keep = cached
        L_ = bind_batch()
        with torch.cuda.device(dev):
            ws, stream = workspace(dev, L_.mq_quantize_p99_passes_workspace_bytes(npass, B, rows, cols))
            st = L_.mq_quantize_p99_passes(arr, npass, B, dt, rows, cols, float(p99),
                                           float(list_p99 if list_p99 is not None else 50.0), ws.data_ptr(), ws.numel(),
                                           stream.cuda_stream)
        _lib.check(st)
        return out
    setups = []
    for k, ps in enumerate(passes):
        kw = dict(bits=8, group_size=None, datatype=None, f32_math=False, deq=True, codes=True, scale=True,
                  seg_sse=False, seg_mse=False, totals=False)
        kw.update(ps)
        if kw.pop("abs_err", torch.float32) != torch.float32:
            raise ValueError("quantize_batch_passes(): the |error| percentile needs the float32 abs_err output")
        if list_p99 is not None and kw["seg_mse"] is not True:
            raise ValueError("quantize_batch_passes(): list_p99 needs seg_mse=True in every pass")
        setups.append(_batch_setup(w, kw["fmt"], kw["bits"], kw["group_size"], kw["datatype"], kw["f32_math"], kw["deq"],
                                   kw["codes"], kw["scale"], torch.float32, kw["seg_sse"], kw["seg_mse"], kw["totals"],
                                   p99, None if out is None else out[k]))
    results = [s[0] for s in setups]
    _, items0, _, (B, rows, cols, w0) = setups[0]
    if items0 is None:
        return results
    dev = w0.device
    arr = (_P99Pass * len(setups))()
    # the passes' percentile arrays follow each other in memory (the library then takes them in one set of launches)
    all_p99 = torch.empty(len(setups), B, dtype=torch.float32, device=dev)
    all_lp = torch.empty(len(setups), B, dtype=torch.float64, device=dev) if list_p99 is not None else None
    for k, (res, items, cfg, _) in enumerate(setups):
        bufs = res._bufs
        res.abs_p99 = bufs["abs_p99"] = all_p99[k]
        bufs.pop("list_p99", None)
        lp = None
        if all_lp is not None:
            lp = bufs["list_p99"] = all_lp[k]
        res.list_p99 = lp
        res._keep = (items, cfg)
        arr[k].items = items
        arr[k].cfg = ctypes.pointer(cfg)
        arr[k].p99 = res.abs_p99.data_ptr()
        arr[k].list_p99 = None if lp is None else lp.data_ptr()
    L_ = bind_batch()
    with torch.cuda.device(dev):
        ws, stream = workspace(dev, L_.mq_quantize_p99_passes_workspace_bytes(len(setups), B, rows, cols))
        st = L_.mq_quantize_p99_passes(arr, len(setups), B, _DTYPES[w0.dtype], rows, cols, float(p99),
                                       float(list_p99 if list_p99 is not None else 50.0), ws.data_ptr(), ws.numel(),
                                       stream.cuda_stream)
    _lib.check(st)
    # (kept with the results: the tables, and the weights whose addresses they hold)
    results[0]._passes_call = (key, arr, len(setups), B, _DTYPES[w0.dtype], rows, cols, dev, (w, [s[1:3] for s in setups]))
    return results


def passes_percentiles(results):
    """The percentile arrays of quantize_batch_passes' results as the two buffers they are slices of:
    (abs_p99 float32 [passes * B], list_p99 float64 [passes * B] or None), pass after pass."""
    a = results[0].abs_p99._base
    lp = results[0].list_p99
    return a.view(-1), (None if lp is None else lp._base.view(-1))


def _batch_setup(w, fmt, bits, group_size, datatype, f32_math, deq, codes, scale, abs_err, seg_sse, seg_mse, totals, p99,
                 out):
    """The argument checks, output buffers and item table of a quantize_batch call.
    -> (result, items or None when there is nothing to launch, cfg, (B, rows, cols, first tensor))"""
    if p99 is not None and abs_err != torch.float32:
        raise ValueError("p99 needs the float32 abs_err output")
    if seg_sse and seg_mse:
        raise ValueError("seg_sse and seg_mse share one output slot: ask for one of them")
    if isinstance(w, torch.Tensor):
        if w.dim() != 3:
            raise ValueError("quantize_batch() expects [B, rows, cols] or a list of [rows, cols] tensors")
        w = w.contiguous()
        tensors = [w[i] for i in range(w.shape[0])]
    else:
        # separate same-shaped tensors (the layers of a model): their pointers go to the launch as they are
        tensors = [t.contiguous() for t in w]
        if not tensors:
            raise ValueError("quantize_batch() needs at least one tensor")
        t0 = tensors[0]
        if t0.dim() != 2 or any(t.shape != t0.shape or t.dtype != t0.dtype or t.device != t0.device for t in tensors):
            raise ValueError("quantize_batch(): the tensors of a list must share shape [rows, cols], dtype and device")
    w0 = tensors[0] if tensors else w
    if w0.dtype not in _DTYPES:
        raise ValueError(f"unsupported weight dtype {w0.dtype}")
    if not w0.is_cuda:
        raise RuntimeError("quantize_batch() expects ROCm device tensors")
    B = len(tensors)
    rows, cols = (w0.shape[-2], w0.shape[-1])
    cfg = make_cfg(fmt, bits, group_size, datatype, f32_math)
    gs = cfg.group_size
    n = rows * cols
    L = n if gs == -1 else (cols if gs <= 0 else gs)
    if gs > 0 and cols % gs:
        raise RuntimeError(f"shape '[{rows}, {cols // gs}, {gs}]' is invalid for input of size {n}")
    nseg = n // L if L else 0
    dev = w0.device
    res = out if out is not None else QuantResult(rows, cols, nseg)
    res.rows, res.cols, res.nseg = rows, cols, nseg

    # Per-tensor slices of the small outputs are padded to a multiple of 8 entries so that every slice keeps
    # the 16-byte alignment the kernels want; the result exposes unpadded views.  The padded buffers are kept
    # on the result object so that passing it back as `out` reuses them.
    bufs = getattr(res, "_bufs", None) or {}

    def buf(name, want, shape, dtype):
        if not want:
            bufs.pop(name, None)
            return None
        cur = bufs.get(name)
        if cur is None or cur.shape != torch.Size(shape) or cur.dtype != dtype or cur.device != dev:
            cur = torch.empty(shape, dtype=dtype, device=dev)
            bufs[name] = cur
        return cur

    nseg_pad = (nseg + 7) // 8 * 8
    b_deq = buf("deq", deq, (B, rows, cols), torch.float16)
    b_codes = buf("codes", codes, (B, rows, cols), torch.float16 if fmt == _lib.LUT_FP else torch.uint8)
    b_scale = buf("scale", scale, (B, nseg_pad), torch.float16)
    b_zp = buf("zero_point", scale and fmt == _lib.INT_ASYM, (B, nseg_pad), torch.float16)
    b_abs = buf("abs_err", abs_err is not None, (B, rows, cols), abs_err or torch.float32)
    b_sse = buf("seg_sse", seg_sse, (B, nseg_pad), torch.float64)
    b_mse = buf("seg_mse", seg_mse, (B, nseg_pad), torch.float32)
    b_tot = buf("totals", totals, (B, 2), torch.float64)
    res.abs_p99 = buf("abs_p99", p99 is not None, (B,), torch.float32)
    res._bufs = bufs
    res.deq, res.codes, res.abs_err, res.totals = b_deq, b_codes, b_abs, b_tot
    res.scale = None if b_scale is None else b_scale[:, :nseg]
    res.zero_point = None if b_zp is None else b_zp[:, :nseg]
    res.seg_sse = None if b_sse is None else b_sse[:, :nseg]
    res.seg_mse = None if b_mse is None else b_mse[:, :nseg]
    if n == 0 or B == 0:
        if res.totals is not None:
            res.totals.zero_()
        return res, None, cfg, (B, rows, cols, w0)

    def span(t):  # (address of slice 0, bytes between slices)
        return (None, 0) if t is None else (t.data_ptr(), t.stride(0) * t.element_size())

    items = (_QuantItem * B)()
    aed = _lib.F16 if abs_err == torch.float16 else _lib.F32
    seg_flag = _seg_flag(seg_mse)
    (p_deq, s_deq), (p_codes, s_codes), (p_scale, s_scale), (p_zp, s_zp) = span(b_deq), span(b_codes), span(b_scale), span(b_zp)
    (p_abs, s_abs), (p_seg, s_seg), (p_tot, s_tot) = span(b_abs), span(b_mse if seg_mse else b_sse), span(b_tot)
    for i in range(B):
        it = items[i]
        it.w = tensors[i].data_ptr()
        o = it.out
        if p_deq is not None:
            o.deq = p_deq + i * s_deq
        if p_codes is not None:
            o.codes = p_codes + i * s_codes
        if p_scale is not None:
            o.scale = p_scale + i * s_scale
        if p_zp is not None:
            o.zero_point = p_zp + i * s_zp
        if p_abs is not None:
            o.abs_err = p_abs + i * s_abs
        o.abs_err_dtype = aed
        o.seg_mse_f32 = seg_flag
        if p_seg is not None:
            o.seg_sse = p_seg + i * s_seg
        if p_tot is not None:
            o.totals = p_tot + i * s_tot
    return res, items, cfg, (B, rows, cols, w0)


@torch.no_grad()
def quantize_table(w: torch.Tensor, values, math: int, group_size=None, *, deq_dtype=torch.float16,
                   neg_inf_zero=False, deq=True, codes=True, scale=True, abs_err=None, seg_sse=False,
                   seg_mse=False, totals=False) -> QuantResult:
    """quantize() for a value table given as an explicit ascending list (mq_quantize_table): math is
    _lib.TABLE_F16 (float16 op by op; codes / scale float16), _lib.TABLE_F32 (float32; codes / scale float32,
    deq float16 or float32) or _lib.TABLE_MX (shared power-of-two exponent per group; deq only)."""
    if seg_sse and seg_mse:
        raise ValueError("seg_sse and seg_mse share one output slot: ask for one of them")
    if w.dtype not in _DTYPES:
        raise ValueError(f"unsupported weight dtype {w.dtype}")
    if not w.is_cuda:
        raise RuntimeError("quantize_table() expects a ROCm device tensor")
    if not 2 <= len(values) <= _lib.TABLE_MAX_VALUES:
        raise ValueError("a value table holds 2 .. 256 entries")
    cols = w.shape[-1] if w.dim() else 1
    rows = w.numel() // cols if cols else 0
    w2 = w.contiguous()
    gs = 0 if group_size is None else int(group_size)
    if gs < -1:
        gs = 0
    n = rows * cols
    L = n if gs == -1 else (cols if gs <= 0 else gs)
    if gs > 0 and cols % gs:
        raise RuntimeError(f"shape '[{rows}, {cols // gs}, {gs}]' is invalid for input of size {n}")
    nseg = n // L if L else 0
    cfg = _lib.TableCfg(math=math, count=len(values), group_size=gs,
                        deq_dtype=_lib.F32 if deq_dtype == torch.float32 else _lib.F16,
                        flags=_lib.TABLE_NEG_INF_ZERO if neg_inf_zero else 0)
    for i, v in enumerate(values):
        cfg.values[i] = v
    dev = w.device
    wide = torch.float16 if math == _lib.TABLE_F16 else torch.float32
    if math == _lib.TABLE_MX:
        codes = scale = False
    res = QuantResult(rows, cols, nseg)

    def buf(want, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=dev) if want else None

    res.deq = buf(deq, (rows, cols), deq_dtype)
    res.codes = buf(codes, (rows, cols), wide)
    res.scale = buf(scale, (nseg,), wide)
    res.abs_err = buf(abs_err is not None, (rows, cols), abs_err or torch.float32)
    res.seg_sse = buf(seg_sse, (nseg,), torch.float64)
    res.seg_mse = buf(seg_mse, (nseg,), torch.float32)
    res.totals = buf(totals, (2,), torch.float64)
    if n == 0:
        if res.totals is not None:
            res.totals.zero_()
        return res
    L_ = _lib.lib()
    o = _lib.QuantOut(deq=_ptr(res.deq), codes=_ptr(res.codes), scale=_ptr(res.scale), zero_point=None,
                      abs_err=_ptr(res.abs_err), abs_err_dtype=_lib.F16 if abs_err == torch.float16 else _lib.F32,
                      seg_mse_f32=_seg_flag(seg_mse), seg_sse=_ptr(res.seg_mse if seg_mse else res.seg_sse),
                      totals=_ptr(res.totals))
    with torch.cuda.device(dev):
        ws, stream = workspace(dev, L_.mq_quantize_table_workspace_bytes(rows, cols, ctypes.byref(cfg)))
        st = L_.mq_quantize_table(w2.data_ptr(), _DTYPES[w.dtype], rows, cols, ctypes.byref(cfg), ctypes.byref(o),
                                  ws.data_ptr(), ws.numel(), stream.cuda_stream)
    _lib.check(st)
    return res


# search_datatype's candidate lists per (bit width, name) -- quant_weight.py:606-625
MIXED_LISTS = {
    (3, "mixed_bitmod"): ["fp3_er_pos", "fp3_er_neg", "fp3_ea_pos", "fp3_ea_neg"],
    (3, "mixed_er"): ["fp3_er_pos", "fp3_er_neg"],
    (3, "mixed_ea"): ["fp3_ea_pos", "fp3_ea_neg"],
    (3, "mixed_ant"): ["int3", "fp3"],
    (4, "mixed_bitmod"): ["fp4_er_pos", "fp4_er_neg", "fp4_ea_pos", "fp4_ea_neg"],
    (4, "mixed_er"): ["fp4_er_pos", "fp4_er_neg"],
    (4, "mixed_ea"): ["fp4_ea_pos", "fp4_ea_neg"],
    (4, "mixed_ant"): ["int4", "flint4"],
}
_SEARCH_VALUES = dict(TABLE_LISTS, int3=MX_TABLES["mx_int3"], int4=MX_TABLES["mx_int4"])


_TORCH_SCALAR_NAMES = {torch.bfloat16: "BFloat16", torch.float32: "Float", torch.float64: "Double"}


def search_dtype_check(dtype) -> None:
    """search_datatype accepts float16 weights only, and so does the reference in effect: its running-best error
    tensor and its result take the weight's dtype while the candidates are float16 and their errors float32, so the
    masked assignments of quant_weight.py:644-645 raise for every other dtype.  Same exception, same message
    (recorded from the reference in tests/golden/search_errors.json): the error assignment (:644) fails first when
    the weight is bfloat16 (float32 errors into a bfloat16 tensor); for float32 / float64 weights that one passes
    or promotes and the candidate assignment (:645) fails."""
    if dtype == torch.float16:
        return
    if dtype == torch.bfloat16:
        dst, src = "BFloat16", "Float"
    elif dtype in _TORCH_SCALAR_NAMES:
        dst, src = _TORCH_SCALAR_NAMES[dtype], "Half"
    else:
        raise RuntimeError(f"search_datatype: unsupported weight dtype {dtype}")
    raise RuntimeError(f"Index put requires the source and destination dtypes match, got {dst} for the destination "
                       f"and {src} for the source.")


@torch.no_grad()
def search_tables(w: torch.Tensor, tables, group_size=None, want_choice=False):
    """Per row (or group) of a float16 ROCm tensor [rows, cols], the first of `tables` (value lists) with the
    smallest float16 mean squared error (mq_search_tables).  -> deq float16 [rows, cols] (, choice uint8 [nseg]).
    Other dtypes raise as the reference does (search_dtype_check)."""
    search_dtype_check(w.dtype)
    if not w.is_cuda:
        raise RuntimeError("search_tables() expects a ROCm device tensor")
    rows, cols = w.shape
    gs = int(group_size) if group_size and group_size > 0 else 0
    if gs > 0 and cols % gs:
        raise RuntimeError(f"shape '[{rows}, {cols // gs}, {gs}]' is invalid for input of size {rows * cols}")
    cfgs = (_lib.TableCfg * len(tables))()
    for t, values in enumerate(tables):
        if not 2 <= len(values) <= 32:
            raise ValueError("the datatype search takes lists of 2 .. 32 values")
        cfgs[t].count = len(values)
        for i, v in enumerate(values):
            cfgs[t].values[i] = v
    w2 = w.contiguous()
    deq = torch.empty_like(w2)
    L = gs if gs > 0 else cols
    choice = torch.empty(rows * cols // L if L else 0, dtype=torch.uint8, device=w.device) if want_choice else None
    with torch.cuda.device(w.device):
        st = _lib.lib().mq_search_tables(w2.data_ptr(), rows, cols, gs, cfgs, len(tables), deq.data_ptr(),
                                         _ptr(choice), torch.cuda.current_stream(w.device).cuda_stream)
    _lib.check(st)
    return (deq, choice) if want_choice else deq
