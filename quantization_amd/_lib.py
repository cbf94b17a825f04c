"""ctypes binding of libmq.so (quantization_amd/csrc), the C ABI declared in include/mq.h.

There is no CPU fallback: importing this module without a built library, or calling into it without
a ROCm device, raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmq.so")

F16, BF16, F32, F64, F32_AS_F64 = 0, 1, 2, 3, 4
INT_SYM, INT_ASYM, LUT_FP, LUT_FP_SUB = 0, 1, 2, 3
TABLE_F16, TABLE_F32, TABLE_MX = 0, 1, 2
TABLE_MAX_VALUES = 256
TABLE_NEG_INF_ZERO = 1

OK, ERR_ARG, ERR_GROUP, ERR_UNSUPPORTED, ERR_WORKSPACE, ERR_HIP, ERR_ALIGN, ERR_TOO_LARGE = 0, -1, -2, -3, -4, -5, -6, -7


class QuantCfg(ctypes.Structure):
    _fields_ = [("format", ctypes.c_int32), ("bits", ctypes.c_int32), ("exp_bits", ctypes.c_int32),
                ("man_bits", ctypes.c_int32), ("exp_bias", ctypes.c_int32), ("f32_math", ctypes.c_int32),
                ("group_size", ctypes.c_int64)]


class QuantOut(ctypes.Structure):
    _fields_ = [("deq", ctypes.c_void_p), ("codes", ctypes.c_void_p), ("scale", ctypes.c_void_p),
                ("zero_point", ctypes.c_void_p), ("abs_err", ctypes.c_void_p), ("abs_err_dtype", ctypes.c_int32),
                ("seg_mse_f32", ctypes.c_int32), ("seg_sse", ctypes.c_void_p), ("totals", ctypes.c_void_p)]


class TableCfg(ctypes.Structure):
    _fields_ = [("math", ctypes.c_int32), ("count", ctypes.c_int32), ("deq_dtype", ctypes.c_int32),
                ("flags", ctypes.c_int32), ("group_size", ctypes.c_int64),
                ("values", ctypes.c_float * TABLE_MAX_VALUES)]


class MqError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded library; raises ImportError with build instructions when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `make -C {os.path.dirname(LIB_PATH)}` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        vp, i64, i32, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_size_t
        L.mq_version.restype = ctypes.c_int
        L.mq_strerror.restype = ctypes.c_char_p
        L.mq_strerror.argtypes = [ctypes.c_int]
        L.mq_last_hip_error.restype = ctypes.c_char_p
        L.mq_quantize_workspace_bytes.restype = sz
        L.mq_quantize_workspace_bytes.argtypes = [i64, i64, ctypes.POINTER(QuantCfg)]
        L.mq_store_host.restype = ctypes.c_int
        L.mq_store_host.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(sz), i32, vp]
        L.mq_profile_reserve.restype = ctypes.c_int
        L.mq_profile_reserve.argtypes = [i64]
        L.mq_profile_begin.restype = ctypes.c_int
        L.mq_profile_begin.argtypes = []
        L.mq_profile_end.restype = ctypes.c_int
        L.mq_profile_end.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]
        L.mq_quantize.restype = ctypes.c_int
        L.mq_quantize.argtypes = [vp, i32, i64, i64, ctypes.POINTER(QuantCfg), ctypes.POINTER(QuantOut), vp, sz, vp]
        L.mq_quantize_table_workspace_bytes.restype = sz
        L.mq_quantize_table_workspace_bytes.argtypes = [i64, i64, ctypes.POINTER(TableCfg)]
        L.mq_quantize_table.restype = ctypes.c_int
        L.mq_quantize_table.argtypes = [vp, i32, i64, i64, ctypes.POINTER(TableCfg), ctypes.POINTER(QuantOut), vp, sz, vp]
        L.mq_search_tables.restype = ctypes.c_int
        L.mq_search_tables.argtypes = [vp, i64, i64, i64, ctypes.POINTER(TableCfg), i32, vp, vp, vp]
        _lib = L
    return _lib


def check(status, shape_msg=None):
    """Map a negative mq_status onto the exception type the reference raises in the same situation."""
    if status == OK:
        return
    L = lib()
    msg = L.mq_strerror(status).decode()
    if status == ERR_GROUP:
        # torch's reshape failure in the reference (quant_weight.py:253-255) is a RuntimeError
        raise RuntimeError(shape_msg or msg)
    if status in (ERR_ARG, ERR_UNSUPPORTED, ERR_TOO_LARGE):
        raise ValueError(msg)
    if status == ERR_HIP:
        raise MqError(f"{msg}: {L.mq_last_hip_error().decode()}")
    raise MqError(msg)
