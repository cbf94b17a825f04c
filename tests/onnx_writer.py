"""A few lines of protobuf encoding, enough to write the ONNX files the tests read back (the `onnx` package
is not installed here).  Field numbers are onnx.proto's; tensors of `external_threshold` bytes and more go to
one external-data file next to the model, as export_and_quant.py:118-126 of the reference saves them."""
import os

import numpy as np

_DT = {np.dtype(np.float32): 1, np.dtype(np.uint8): 2, np.dtype(np.int8): 3, np.dtype(np.int32): 6,
       np.dtype(np.int64): 7, np.dtype(np.float16): 10, np.dtype(np.float64): 11}


def _varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _vi(field, v):
    return _varint(field << 3) + _varint(v)


def tensor_proto(name, arr, external=None, data_type=None, packed_dims=True):
    """external: (location, offset, length) or None."""
    dims = b"".join(_varint(d) for d in arr.shape)
    body = _ld(1, dims) if packed_dims else b"".join(_vi(1, d) for d in arr.shape)
    body += _vi(2, data_type if data_type is not None else _DT[arr.dtype])
    body += _ld(8, name.encode())
    if external is None:
        body += _ld(9, arr.tobytes())
    else:
        loc, off, length = external
        for k, v in (("location", loc), ("offset", str(off)), ("length", str(length))):
            body += _ld(13, _ld(1, k.encode()) + _ld(2, v.encode()))
        body += _vi(14, 1)
    return body


def node_proto(op, inputs, outputs, name=""):
    body = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    return body + _ld(3, name.encode()) + _ld(4, op.encode())


def write_model(path, initializers, nodes, external_threshold=1024, data_types=None):
    """initializers: dict name -> numpy array; nodes: list of (op, inputs, outputs)."""
    data_types = data_types or {}
    os.makedirs(os.path.dirname(path), exist_ok=True)
    ext_name = os.path.basename(path) + ".data"
    ext = bytearray()
    graph = b""
    for op, ins, outs in nodes:
        graph += _ld(1, node_proto(op, ins, outs))
    graph += _ld(2, b"g")
    for k, (name, arr) in enumerate(initializers.items()):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes >= external_threshold:
            while len(ext) % 64:
                ext.append(0)
            tp = tensor_proto(name, arr, external=(ext_name, len(ext), arr.nbytes), data_type=data_types.get(name))
            ext += arr.tobytes()
        else:
            tp = tensor_proto(name, arr, data_type=data_types.get(name), packed_dims=(k % 2 == 0))
        graph += _ld(5, tp)
    model = _vi(1, 8) + _ld(2, b"onnx_writer") + _ld(7, graph) + _ld(8, _ld(1, b"") + _vi(2, 17))
    with open(path, "wb") as f:
        f.write(model)
    if ext:
        with open(os.path.join(os.path.dirname(path), ext_name), "wb") as f:
            f.write(bytes(ext))
