"""Weight initializers of ONNX files, read without the `onnx` package: the input side of the reference's
analysis scripts (MSE&P99/quant_error_multi_weight.py:30-38, :90-116 under /root/reference).

The C++ reader behind include/mq_onnx.h memory-maps model.onnx and its external-data file; tensors come out
as numpy views of those mappings (no copy) and go to the ROCm device with one host-to-device copy each."""
import ctypes
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib

# TensorProto.DataType -> numpy dtype (the scripts' DTYPE_MAP, quant_error_multi_weight.py:16-27; bfloat16 is
# carried as its 16-bit pattern)
DTYPE_MAP = {1: np.float32, 2: np.uint8, 3: np.int8, 6: np.int32, 7: np.int64, 10: np.float16, 11: np.float64,
             16: np.uint16}
BFLOAT16 = 16


class _Tensor(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data_type", ctypes.c_int32), ("ndims", ctypes.c_int32),
                ("dims", ctypes.c_int64 * 8), ("data", ctypes.c_void_p), ("nbytes", ctypes.c_uint64),
                ("external", ctypes.c_int32), ("compute_weight", ctypes.c_int32)]


def _bind():
    L = _lib.lib()
    if getattr(L, "_onnx_bound", False):
        return L
    L.mq_onnx_open.restype = ctypes.c_int
    L.mq_onnx_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
    L.mq_onnx_close.restype = None
    L.mq_onnx_close.argtypes = [ctypes.c_void_p]
    L.mq_onnx_num_initializers.restype = ctypes.c_int64
    L.mq_onnx_num_initializers.argtypes = [ctypes.c_void_p]
    L.mq_onnx_initializer.restype = ctypes.c_int
    L.mq_onnx_initializer.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(_Tensor)]
    L.mq_onnx_find.restype = ctypes.c_int64
    L.mq_onnx_find.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    L.mq_onnx_error.restype = ctypes.c_char_p
    L._onnx_bound = True
    return L


class OnnxWeights:
    """with OnnxWeights(path) as m: m.names(), m.compute_weights(), m.array(name), m.tensor(name, device)"""

    def __init__(self, model_path: str):
        self._L = _bind()
        h = ctypes.c_void_p()
        if self._L.mq_onnx_open(model_path.encode(), ctypes.byref(h)) != 0:
            raise OSError(f"{model_path}: {self._L.mq_onnx_error().decode()}")
        self._h = h
        self._info: Dict[str, _Tensor] = {}
        self._order: List[str] = []
        for i in range(self._L.mq_onnx_num_initializers(h)):
            t = _Tensor()
            if self._L.mq_onnx_initializer(h, i, ctypes.byref(t)) != 0:
                raise ValueError(self._L.mq_onnx_error().decode())
            name = t.name.decode()
            self._info[name] = t
            self._order.append(name)

    def close(self):
        if self._h:
            self._L.mq_onnx_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def names(self) -> List[str]:
        """Initializer names in file order (the iteration order of the scripts' dict)."""
        return list(self._order)

    def compute_weights(self) -> List[str]:
        """find_compute_weights (quant_error_multi_weight.py:90-116), in file order."""
        return [n for n in self._order if self._info[n].compute_weight]

    def shape(self, name: str):
        t = self._info[name]
        return tuple(t.dims[i] for i in range(t.ndims))

    def data_type(self, name: str) -> int:
        return self._info[name].data_type

    def array(self, name: str) -> np.ndarray:
        """np.frombuffer(init.raw_data, dtype).reshape(init.dims) -- a read-only view of the mapped file, valid
        until close()."""
        t = self._info[name]
        if t.data_type not in DTYPE_MAP:
            raise KeyError(t.data_type)  # the scripts' DTYPE_MAP lookup fails the same way
        if not t.data and t.nbytes == 0 and int(np.prod(self.shape(name))) != 0:
            raise ValueError(f"{name}: no raw_data")
        dt = np.dtype(DTYPE_MAP[t.data_type])
        buf = (ctypes.c_uint8 * t.nbytes).from_address(t.data) if t.nbytes else b""
        return np.frombuffer(buf, dtype=dt).reshape(self.shape(name))

    def tensor(self, name: str, device=None) -> torch.Tensor:
        """The initializer on the ROCm device in its stored dtype (bfloat16 as torch.bfloat16)."""
        a = self.array(name)
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        t = torch.from_numpy(np.array(a, copy=True)) if a.size else torch.empty(a.shape, dtype=torch.float32)
        if self._info[name].data_type == BFLOAT16:
            t = t.view(torch.int16).view(torch.bfloat16)
        return t.to(dev)
