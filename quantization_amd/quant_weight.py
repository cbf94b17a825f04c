"""Drop-in for the reference's quant_utils/quant_weight.py, running on MI355X.

Same function names, argument meaning, return shapes/dtypes and exception types as the reference
(citations are into /root/reference/quant_utils/quant_weight.py); the arithmetic runs in the fused HIP
kernel behind include/mq.h.  A host tensor is moved to the current ROCm device and the result moved
back, so the reference's CPU-tensor call sites keep working; there is no CPU compute path.

Supported on the device: symmetric / asymmetric INT with 2..8 bits and every value-table format of the
reference, each per-tensor (group_size == -1), per-channel (None / 0 / other negatives) or per-group (> 0):
  * regular minifloat grids through the fused kernels -- fp8_e4m3, fp8_e3m4, fp8_e2m5 (no subnormals: packed
    fast path), fp4, fp5 / fp5_e2m2, fp5_e3m1, fp6 / fp6_e2m3, fp6_e3m2 and the INTn lists when passed to
    quant_datatype (tables with subnormals: packed rounding with a per-element shift);
  * the irregular lists (fp3, flint4, flint5, the *_er_* / *_ea_* variants), fp8_e5m2's float32 path and MX
    (quant_mx) through the explicit-table kernel (mq_quantize_table).
search_datatype (the mixed-datatype search) runs on float16 weights (mq_search_tables).
"""
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .fused import (LUT_FORMATS, MIXED_LISTS, MX_TABLES, TABLE_LISTS, _SEARCH_VALUES, lut_values, quantize, quantize_batch,
                    quantize_table, require_device, search_dtype_check, search_tables)

# names the reference accepts per bit width (quant_weight.py:172-203); only used to reproduce its
# ValueError / AssertionError behaviour for unknown names
_KNOWN_TABLES = {
    3: {"int3", "fp3", "fp3_er_pos", "fp3_er_neg", "fp3_ea_pos", "fp3_ea_neg"},
    4: {"int4", "fp4", "flint4", "fp4_er_pos", "fp4_er_neg", "fp4_ea_pos", "fp4_ea_neg"},
    5: {"int5", "fp5", "flint5", "fp5_e2m2", "fp5_e3m1"},
    6: {"int6", "fp6", "fp6_e2m3", "fp6_e3m2"},
    8: {"int8", "fp8_e2m5", "fp8_e3m4", "fp8_e4m3", "fp8_e5m2"},
}


def _is_per_tensor(group_size):
    return group_size == -1


def _is_per_group(group_size):
    return group_size is not None and group_size > 0


def _run(w, fmt, bits, group_size, datatype=None, **kw):
    """-> (QuantResult, original device).  Per-group needs a 2-D weight, as `K, C = w.size()` does
    in the reference (quant_weight.py:252)."""
    if _is_per_group(group_size):
        K, C = w.size()  # raises ValueError for non 2-D input like the reference's unpacking
    dev = w.device
    res = quantize(require_device(w), fmt, bits, group_size, datatype, **kw)
    return res, dev


def _check_table(wq_bits, datatype):
    if wq_bits not in _KNOWN_TABLES:
        # quant_weight.py:552-554 / :790-791
        raise ValueError(f"Currently only support 3-, 4-, 5-,6- and 8-bit quantization, not {wq_bits}-bit")
    assert datatype in _KNOWN_TABLES[wq_bits], f"unexpected data type {datatype}."  # :556 / :793


def _table_run(w, datatype, group_size, math, **kw):
    """The explicit-list formats (TABLE_LISTS); same conventions as _run."""
    if _is_per_group(group_size):
        K, C = w.size()
    dev = w.device
    res = quantize_table(require_device(w), TABLE_LISTS[datatype], math, group_size, **kw)
    return res, dev


def _e5m2(w, group_size, meta):
    """FP8_E5M2: float32 arithmetic and float32 results (quant_weight.py:505-540, :796-826)."""
    res, dev = _table_run(w, "fp8_e5m2", group_size, _lib.TABLE_F32, deq_dtype=torch.float32, neg_inf_zero=True,
                          codes=meta, scale=meta)
    return res, dev


@torch.no_grad()
def quant_int(w_fp16, wq_bits: int = 4, group_size: Optional[int] = None):
    """Symmetric INT fake-quantization; returns the dequantized float16 weight (quant_weight.py:229-266)."""
    res, dev = _run(w_fp16, _lib.INT_SYM, wq_bits, group_size, codes=False, scale=False)
    return res.deq.reshape(w_fp16.shape).to(dev)


@torch.no_grad()
def quant_int_asym(w_fp16, wq_bits: int = 4, group_size: Optional[int] = None):
    """Asymmetric INT fake-quantization; returns the dequantized float16 weight (quant_weight.py:298-335)."""
    res, dev = _run(w_fp16, _lib.INT_ASYM, wq_bits, group_size, codes=False, scale=False)
    return res.deq.reshape(w_fp16.shape).to(dev)


@torch.no_grad()
def quant_datatype(w_fp16, wq_bits: int = 4, datatype: str = "", group_size: Optional[int] = None):
    """Nearest-value quantization to a format's value table (quant_weight.py:503-601).  The per-tensor
    branch computes in float32 as the reference does (:562-575)."""
    if datatype == "fp8_e5m2":  # tested before the bit width, as in the reference (:505)
        res, dev = _e5m2(w_fp16, group_size, meta=False)
        return res.deq.reshape(w_fp16.shape).to(dev)
    _check_table(wq_bits, datatype)
    if datatype in TABLE_LISTS:
        math = _lib.TABLE_F32 if _is_per_tensor(group_size) else _lib.TABLE_F16
        res, dev = _table_run(w_fp16, datatype, group_size, math, codes=False, scale=False)
    elif _is_per_tensor(group_size) and (w_fp16.numel() % 8 or w_fp16.data_ptr() % 16 or not w_fp16.is_contiguous()):
        # the fused kernels' float32 per-tensor branch reads whole, aligned 8-element vectors: any other tensor goes
        # through the explicit-list kernel with the format's value list (the same float32 arithmetic, :562-575)
        dev = w_fp16.device
        res = quantize_table(require_device(w_fp16), lut_values(datatype), _lib.TABLE_F32, group_size, codes=False,
                             scale=False)
    else:
        res, dev = _run(w_fp16, _lib.LUT_FP, wq_bits, group_size, datatype, f32_math=_is_per_tensor(group_size),
                        codes=False, scale=False)
    return res.deq.reshape(w_fp16.shape).to(dev)


@torch.no_grad()
def quant_mx(w_fp16, wq_bits: int = 4, datatype: str = "", group_size: int = 32):
    """MX quantization: a shared power-of-two exponent per group of group_size input channels
    (quant_weight.py:337-380).  Returns the dequantized float16 weight."""
    if wq_bits not in (3, 4):
        raise ValueError(f"Currently only support 3-bit, 4-bit quantization, not {wq_bits}-bit")  # :348
    names = {3: ("mx_int3", "mx_fp3"), 4: ("mx_int4", "mx_fp4")}[wq_bits]
    assert datatype in names, f"unexpected data type {datatype}."  # :351
    K, C = w_fp16.size()
    dev = w_fp16.device
    res = quantize_table(require_device(w_fp16), MX_TABLES[datatype], _lib.TABLE_MX, group_size)
    return res.deq.reshape(K, C).to(dev)


@torch.no_grad()
def search_datatype(w_fp16, wq_bits: int = 4, datatype: str = "mixed_bitmod", group_size: Optional[int] = None):
    """Per row / group, the first of the candidate value lists with the smallest float16 mean squared error
    (quant_weight.py:604-649).  Returns the dequantized weight in the input's dtype (float16)."""
    if wq_bits not in (3, 4):
        raise ValueError(f"Currently only support 3-bit and 4-bit mixed quantization, not {wq_bits}-bit")  # :627
    if (wq_bits, datatype) not in MIXED_LISTS:
        # (the reference falls through to an unbound datatype_list here)
        raise ValueError(f"unknown mixed datatype {datatype!r}")
    K, C = w_fp16.size()
    dev = w_fp16.device
    search_dtype_check(w_fp16.dtype)   # (float16 only, as the reference: its masked assignments raise otherwise)
    tables = [_SEARCH_VALUES[name] for name in MIXED_LISTS[(wq_bits, datatype)]]
    return search_tables(require_device(w_fp16), tables, group_size).reshape(K, C).to(dev)


@torch.no_grad()
def _int_quant_meta(w_fp16: torch.Tensor, wq_bits: int = 4, asym: bool = False, group_size: Optional[int] = None):
    """(w_deq float16, q int8/uint8, scale float16, zp float16 | None) -- quant_weight.py:717-775.

    Code dtype follows the reference: uint8 for asymmetric per-channel/per-group (:766) and for symmetric
    per-tensor (:745), int8 for symmetric per-channel/per-group (:775) and asymmetric per-tensor (:738);
    the byte is the low byte of the integer code in every case."""
    res, dev = _run(w_fp16, _lib.INT_ASYM if asym else _lib.INT_SYM, wq_bits, group_size)
    w_deq = res.deq.reshape(w_fp16.shape)
    q, scale, zp = _shape_int_meta(w_fp16.shape, asym, group_size, res.codes, res.scale, res.zero_point)
    return w_deq.to(dev), q.to(dev), scale.to(dev), (zp.to(dev) if zp is not None else None)


def _shape_int_meta(shape, asym, group_size, codes, scale, zero_point):
    """Views / dtypes of the integer codes and parameters as _int_quant_meta returns them (:738-775)."""
    if _is_per_tensor(group_size):
        q = codes.reshape(shape)
        q = q.view(torch.int8) if asym else q
        return q, scale.reshape(()), (zero_point.reshape(()) if asym else None)
    if _is_per_group(group_size):
        K, C = shape
        q = codes.reshape(K, C // group_size, group_size)
        sshape = (K, C // group_size)
    else:
        q = codes.reshape(shape)
        sshape = tuple(shape[:-1])
    q = q if asym else q.view(torch.int8)
    return q, scale.reshape(sshape), (zero_point.reshape(sshape) if asym else None)


@torch.no_grad()
def _fp_quant_meta(w_fp16, wq_bits: int = 4, datatype: str = "", group_size: Optional[int] = None):
    """(w_deq float16, q float16 table values, scale float16) -- quant_weight.py:778-872; all three are
    float32 for fp8_e5m2 (:796-826), whose q keeps the weight's shape also per group (:824)."""
    _check_table(wq_bits, datatype)
    if datatype == "fp8_e5m2":
        res, dev = _e5m2(w_fp16, group_size, meta=True)
        _, scale = _shape_fp_meta(w_fp16.shape, group_size, res.codes, res.scale)
        return (res.deq.reshape(w_fp16.shape).to(dev), res.codes.reshape(w_fp16.shape).to(dev), scale.to(dev))
    if datatype in TABLE_LISTS:
        res, dev = _table_run(w_fp16, datatype, group_size, _lib.TABLE_F16)
    else:
        res, dev = _run(w_fp16, _lib.LUT_FP, wq_bits, group_size, datatype)
    w_deq = res.deq.reshape(w_fp16.shape)
    q, scale = _shape_fp_meta(w_fp16.shape, group_size, res.codes, res.scale)
    return w_deq.to(dev), q.to(dev), scale.to(dev)


def _shape_fp_meta(shape, group_size, codes, scale):
    """Views of the table values and scales as _fp_quant_meta returns them (:850-872)."""
    if _is_per_tensor(group_size):
        return codes.reshape(shape), scale.reshape(())
    if _is_per_group(group_size):
        K, C = shape
        return codes.reshape(K, C // group_size, group_size), scale.reshape(K, C // group_size)
    return codes.reshape(shape), scale.reshape(tuple(shape[:-1]))


class _LinearDump(nn.Linear):
    """nn.Linear carrying weight_q / weight_scale / weight_zp buffers (quant_weight.py:876-914).
    As in the reference the buffers come from quantizing the weight it is given -- which quant_model has
    already replaced by its dequantized value."""

    def __init__(self, src: nn.Linear, wq_bits: int, mode: str, group_size: Optional[int], meta=None):
        """meta: (q, scale, zp | None) already computed for src.weight (quant_model's batched pass)."""
        super().__init__(src.in_features, src.out_features, bias=src.bias is not None,
                         device=src.weight.device, dtype=src.weight.dtype)
        self.weight = nn.Parameter(src.weight.detach())
        if src.bias is not None:
            self.bias = nn.Parameter(src.bias.detach())
        if meta is not None:
            q, scale, zp = meta
        elif mode.startswith("int"):
            _, q, scale, zp = _int_quant_meta(self.weight.data, wq_bits=wq_bits, asym="asym" in mode,
                                              group_size=group_size)
        else:
            _, q, scale = _fp_quant_meta(self.weight.data, wq_bits=wq_bits, datatype=mode, group_size=group_size)
            zp = None
        if mode == "fp8_e5m2":  # float32 buffers (:898-900)
            self.register_buffer("weight_q", q)
            self.register_buffer("weight_scale", scale)
        else:
            self.register_buffer("weight_q", q.to(torch.float16))
            self.register_buffer("weight_scale", scale.to(torch.float16))
        if zp is not None:
            self.register_buffer("weight_zp", zp)

    def forward(self, x):
        out = F.linear(x, self.weight, self.bias)
        keep = self.weight_q.float().sum() + self.weight_scale.float().sum()
        if hasattr(self, "weight_zp"):
            keep = keep + self.weight_zp.float().sum()
        return out + keep * 0.0


def _linears(model):
    return [(n, m) for n, m in model.named_modules() if isinstance(m, nn.Linear)]


def quant_model(model, wq_bits: Optional[int] = None, wq_datatype: Optional[str] = None,
                wq_groupsize: Optional[int] = None, _dump_int: bool = True):
    """Fake-quantize every nn.Linear weight in place, then (unless _dump_int is False) swap each Linear
    for a _LinearDump holding codes / scales / zero-points (quant_weight.py:652-706, :921-936).
    Dispatch on the datatype string follows the reference's order.

    The reference loops over the layers; here the Linear weights are grouped by (shape, dtype, device) and
    every group goes through one fused launch per 32 layers, once for the dequantized weights and once more
    -- on the already dequantized weights, as the reference's _LinearDump does (:882-895) -- for the codes."""
    if wq_datatype is None or wq_datatype in ("fp16", "fp32"):
        print("Not applying quantization")
        return
    if wq_datatype.startswith("int"):
        asym = "asym" in wq_datatype
        fmt, table = (_lib.INT_ASYM if asym else _lib.INT_SYM), None
        kind = "asymmetric" if asym else "symmetric"
        print(f"Applying {kind} INT quantization with bits: {wq_bits}, group size: {wq_groupsize}")
    elif "mx" in wq_datatype:
        # hard-coded group size 32, and no _LinearDump for these names (:674-686, :929-930)
        print(f"Applying MX quantization with bits: {wq_bits}, datatype: {wq_datatype}, group size: 32")
        for _, m in _linears(model):
            m.weight.data = quant_mx(m.weight.data, wq_bits=wq_bits, datatype=wq_datatype, group_size=32)
        return
    elif "mixed" in wq_datatype:
        print(f"Applying mixed datatype quantization with bits: {wq_bits}, datatype: {wq_datatype}, "
              f"group size: {wq_groupsize}")
        for _, m in _linears(model):
            m.weight.data = search_datatype(m.weight.data, wq_bits=wq_bits, datatype=wq_datatype,
                                            group_size=wq_groupsize)
        return
    elif "fp" in wq_datatype:
        print(f"Applying floating-point datatype quantization with bits: {wq_bits}, group size: {wq_groupsize}")
        if wq_datatype in TABLE_LISTS:
            # explicit-list formats: layer by layer through the table kernel, then the reference's _LinearDump
            for _, m in _linears(model):
                m.weight.data = quant_datatype(m.weight.data, wq_bits=wq_bits, datatype=wq_datatype,
                                               group_size=wq_groupsize)
            if _dump_int:
                for name, mod in _linears(model):
                    parent, attr = _find_parent(model, name)
                    setattr(parent, attr, _LinearDump(mod, wq_bits, wq_datatype, wq_groupsize))
            return
        _check_table(wq_bits, wq_datatype)
        asym, fmt, table = False, _lib.LUT_FP, wq_datatype
    else:
        raise ValueError(f"Unsupported datatype {wq_datatype}")

    linears = _linears(model)
    groups = {}
    for name, m in linears:
        w = m.weight.data
        if _is_per_group(wq_groupsize):
            K, C = w.size()
        groups.setdefault((tuple(w.shape), w.dtype, w.device), []).append((name, m))
    meta = {}
    for (shape, _, dev), members in groups.items():
        rows = 1
        for d in shape[:-1]:
            rows *= d
        stack = [require_device(m.weight.data).reshape(rows, shape[-1]) for _, m in members]
        # 1) dequantized weights (quant_datatype's per-tensor branch runs in float32, :562-575)
        res = quantize_batch(stack, fmt, wq_bits, wq_groupsize, table, codes=False, scale=False,
                             f32_math=(fmt == _lib.LUT_FP and _is_per_tensor(wq_groupsize)))
        for i, (_, m) in enumerate(members):
            m.weight.data = res.deq[i].reshape(shape).to(dev)
        if not _dump_int:
            continue
        # 2) codes / scales / zero-points of the DEQUANTIZED weights
        stack = [require_device(m.weight.data).reshape(rows, shape[-1]) for _, m in members]
        res = quantize_batch(stack, fmt, wq_bits, wq_groupsize, table, deq=False)
        for i, (name, _) in enumerate(members):
            if fmt == _lib.LUT_FP:
                q, scale = _shape_fp_meta(shape, wq_groupsize, res.codes[i], res.scale[i])
                zp = None
            else:
                q, scale, zp = _shape_int_meta(shape, asym, wq_groupsize, res.codes[i], res.scale[i],
                                               res.zero_point[i] if asym else None)
            meta[name] = (q.to(dev), scale.to(dev), zp.to(dev) if zp is not None else None)
    if not _dump_int:
        return
    for name, mod in linears:
        parent, attr = _find_parent(model, name)
        setattr(parent, attr, _LinearDump(mod, wq_bits, wq_datatype, wq_groupsize, meta=meta[name]))


def _find_parent(root: nn.Module, target_name: str):
    parts = target_name.split(".")
    for p in parts[:-1]:
        root = getattr(root, p)
    return root, parts[-1]
