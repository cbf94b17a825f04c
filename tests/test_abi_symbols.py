"""CPU: libmq.so loads and exports every function include/*.h declares (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for header in ("mq.h", "mq_onnx.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(mq_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_header_symbols():
    from quantization_amd import _lib
    names = declared_functions()
    assert "mq_quantize" in names and "mq_version" in names and "mq_onnx_open" in names
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"libmq.so does not export {n}"
    assert _lib.lib().mq_version() >= 100
    assert _lib.lib().mq_strerror(-2).decode().startswith("cols")


def test_struct_layout_matches_header():
    from quantization_amd import _lib
    # mq_quant_cfg: 6 x int32 + int64; mq_quant_out: 5 pointers, 2 x int32, 2 pointers
    assert ctypes.sizeof(_lib.QuantCfg) == 32
    assert ctypes.sizeof(_lib.QuantOut) == 5 * 8 + 8 + 2 * 8
    assert ctypes.sizeof(_lib.TableCfg) == 4 * 4 + 8 + 4 * 256   # mq_table_cfg


def test_argument_validation_needs_no_gpu():
    """Status codes that are decided before any HIP call."""
    from quantization_amd import _lib
    L = _lib.lib()
    cfg = _lib.QuantCfg(format=_lib.INT_SYM, bits=8, group_size=64)
    out = _lib.QuantOut()
    assert L.mq_quantize(None, _lib.F16, 4, 100, ctypes.byref(cfg), ctypes.byref(out), None, 0, None) == _lib.ERR_GROUP
    cfg.group_size = 0
    cfg.bits = 9
    assert L.mq_quantize(None, _lib.F16, 4, 64, ctypes.byref(cfg), ctypes.byref(out), None, 0, None) == _lib.ERR_UNSUPPORTED
    cfg.bits = 8
    assert L.mq_quantize(None, 7, 4, 64, ctypes.byref(cfg), ctypes.byref(out), None, 0, None) == _lib.ERR_ARG
    assert L.mq_quantize(None, _lib.F16, 0, 64, ctypes.byref(cfg), ctypes.byref(out), None, 0, None) == _lib.OK
    assert L.mq_quantize(None, _lib.F16, 4, 64, ctypes.byref(cfg), ctypes.byref(out), None, 0, None) == _lib.ERR_ARG


def test_table_and_histogram_argument_validation_needs_no_gpu():
    from quantization_amd import _lib
    L = _lib.lib()
    out = _lib.QuantOut()
    cfg = _lib.TableCfg(math=_lib.TABLE_F16, count=3, deq_dtype=_lib.F16, group_size=64)
    for i, v in enumerate((-1.0, 0.0, 1.0)):
        cfg.values[i] = v
    call = lambda rows, cols: L.mq_quantize_table(None, _lib.F16, rows, cols, ctypes.byref(cfg), ctypes.byref(out),
                                                  None, 0, None)
    assert call(4, 100) == _lib.ERR_GROUP
    assert call(0, 64) == _lib.OK
    assert call(4, 64) == _lib.ERR_ARG                      # no weight pointer
    cfg.values[2] = -0.5
    assert call(4, 64) == _lib.ERR_ARG                      # not ascending
    cfg.values[2] = 1.0 + 2.0 ** -12
    assert call(4, 64) == _lib.ERR_UNSUPPORTED              # not a float16 value under float16 arithmetic
    cfg.values[2] = 1.0
    cfg.math, cfg.group_size = _lib.TABLE_MX, 0
    assert call(4, 64) == _lib.ERR_UNSUPPORTED              # MX needs a group size
    cfg.math, cfg.deq_dtype = _lib.TABLE_F16, _lib.F32
    assert call(4, 64) == _lib.ERR_UNSUPPORTED              # float32 deq only with float32 arithmetic
    cfg.deq_dtype, cfg.count = _lib.F16, 1
    assert call(4, 64) == _lib.ERR_ARG
    assert L.mq_search_tables(None, 4, 100, 64, ctypes.byref(cfg), 1, None, None, None) == _lib.ERR_GROUP
    assert L.mq_search_tables(None, 4, 64, 64, ctypes.byref(cfg), 9, None, None, None) == _lib.ERR_ARG
    L.mq_histogram.restype = ctypes.c_int
    L.mq_histogram.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32,
                               ctypes.c_void_p, ctypes.c_void_p]
    assert L.mq_histogram(None, _lib.F32, 10, None, 10, None, None) == _lib.ERR_ARG
    assert L.mq_histogram(None, 9, 0, None, 10, None, None) == _lib.ERR_ARG
