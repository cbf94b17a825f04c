#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by IMPORTING the reference
(/root/reference/quant_utils/quant_weight.py) in the build container.

Run once, by hand, where /root/reference exists:  python tests/golden/gen_golden.py
The GPU box has no /root/reference; tests only read the .npz / .json files written here.

Files written:
  quant_small.npz       inputs (several distributions x input dtypes) and, for every
                        format x group size, the reference's deq / codes / scale / zero-point bits
  quant_exhaustive.npz  all 65 536 float16 bit patterns as one 256x256 matrix (non-finite ones
                        replaced by 0) and the reference outputs for per-channel / g=128
  lut_domain.npz        x-domain sweep of the LUT rounding: rows whose scale is exactly 1
  lut_tables.json       the reference's value lists of every table format the kernels support
  quant_more_formats.npz  the formats beyond the 8-bit headline set: value tables fp4, fp5_e2m2, fp5_e3m1,
                        fp6_e2m3, fp6_e3m2, fp8_e2m5 and the INTn lists used as tables (int3, int4, int6),
                        4-bit INT sym / asym; small inputs, all float16 patterns, the rounding-domain sweep
  quant_table_formats.npz  the explicit-list formats: fp3, flint4, flint5 and the *_er_* / *_ea_* lists
                        (_fp_quant_meta + quant_datatype), fp8_e5m2 (float32 results) and quant_mx; small inputs,
                        non-finite inputs, and all float16 patterns for a subset
  quant_search.npz      search_datatype (mixed_bitmod / mixed_er / mixed_ea / mixed_ant, 3 and 4 bits) on float16 inputs
  quant_specials.npz    the four headline formats on rows holding +-inf / NaN, rows on one side of zero, all-zero
                        and constant rows, values at the float16 limits (per-channel, g=64, per-tensor)
  quant_model.npz       quant_model (the _LinearDump wrapper included, quant_weight.py:876-936) on a seeded
                        nn.Sequential of four float16 nn.Linear: weights before / after, every registered buffer
                        (bytes, dtype, shape) and the module class names, for int8 / int8_asym / int4_asym /
                        fp8_e4m3 / fp4 x {per-channel, g=64, per-tensor}
  quant_zero_sign.npz   int8_asym / int4_asym on rows whose minimum is a zero, with zeros of both signs in every segment
                        (g = 8 / 16 / 32 / 64 / 128, per-channel, per-tensor): which zero torch.amin returns decides the
                        sign of a zero zero-point
  onnx/                 model files written by PyTorch's ONNX serializer (plain and quantized by the reference's
                        quant_model), every parameter's bits, the MatMul / Gemm weight names
  search_errors.json    what search_datatype raises for weights that are not float16
  error_stats.json      outputs of the reference's quantizers + the analysis scripts' numpy
                        arithmetic on seeded 512x1024 inputs (scalars and digests only)
"""
import contextlib
import hashlib
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from quant_utils import quant_weight as qw  # noqa: E402  (the reference)

from oracle import oracle as orc  # noqa: E402  (numpy restatement of the analysis scripts)

FORMATS = ["int8", "int8_asym", "fp8_e4m3", "fp8_e3m4"]
GROUPS = [None, 32, 64, 128, -1]


def gs_key(gs):
    return "none" if gs is None else str(gs)


def bits16(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def run_reference(w, fmt, gs):
    """-> dict of numpy bit arrays exactly as the reference returns them."""
    out = {}
    if fmt.startswith("int"):
        asym = "asym" in fmt
        d, q, s, zp = qw._int_quant_meta(w, wq_bits=8, asym=asym, group_size=gs)
        d2 = (qw.quant_int_asym if asym else qw.quant_int)(w, wq_bits=8, group_size=gs)
        assert torch.equal(bits_t(d), bits_t(d2.reshape_as(d)))
        out["deq"] = bits16(d)
        out["q"] = q.contiguous().view(torch.uint8).numpy().copy()
        out["q_is_int8"] = np.array(q.dtype == torch.int8)
        out["scale"] = bits16(s.reshape(-1))
        if zp is not None:
            out["zp"] = bits16(zp.reshape(-1))
    else:
        d, q, s = qw._fp_quant_meta(w, wq_bits=8, datatype=fmt, group_size=gs)
        out["deq"] = bits16(d)
        out["q"] = bits16(q)
        out["scale"] = bits16(s.reshape(-1))
        # quant_datatype: same values, except the per-tensor branch which runs in float32
        d2 = qw.quant_datatype(w, wq_bits=8, datatype=fmt, group_size=gs)
        out["deq_quant_datatype"] = bits16(d2.reshape_as(d))
    return out


def bits_t(t):
    return t.contiguous().view(torch.int16)


def distributions(K, C, gen):
    g = torch.randn(K, C, generator=gen)
    yield "normal", g * 0.02
    yield "heavy", g * 0.6
    yield "tiny", g * 1e-6
    yield "allpos", 1 + torch.rand(K, C, generator=gen) * 1e-3
    o = g * 0.02
    o[:, 3] = 60000.0
    yield "outlier", o
    z = g * 0.02
    z[1] = 0
    z[2, : C // 2] = 0
    z[3, 5] = -0.0
    yield "zerorows", z
    yield "relu", torch.relu(g) * 0.05
    yield "big", g * 3000
    t = torch.distributions.StudentT(3.0).sample((K, C)) * 0.02
    yield "student_t", t


def gen_quant_small():
    gen = torch.Generator().manual_seed(0)
    torch.manual_seed(0)
    K, C = 8, 512
    arrays = {}
    for name, w in distributions(K, C, gen):
        variants = [("f16", w.half())]
        if name in ("normal", "heavy", "big"):
            variants += [("bf16", w.bfloat16()), ("f32", w.float())]
        for dt, wt in variants:
            key = f"{name}.{dt}"
            arrays[f"in/{key}"] = wt.numpy().copy() if dt == "f32" else bits16(wt)
            for fmt in FORMATS:
                for gs in GROUPS:
                    for k, v in run_reference(wt, fmt, gs).items():
                        arrays[f"out/{key}/{fmt}/{gs_key(gs)}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "quant_small.npz"), **arrays)
    print("quant_small.npz", len(arrays), "arrays")


def gen_exhaustive():
    bits = np.arange(65536, dtype=np.uint16).reshape(256, 256)
    w = torch.from_numpy(bits.view(np.int16).copy()).view(torch.float16)
    w = torch.where(torch.isfinite(w), w, torch.zeros_like(w))
    arrays = {"in": bits16(w)}
    for fmt in FORMATS:
        for gs in [None, 128]:
            for k, v in run_reference(w, fmt, gs).items():
                arrays[f"out/{fmt}/{gs_key(gs)}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "quant_exhaustive.npz"), **arrays)
    print("quant_exhaustive.npz", len(arrays), "arrays")


def gen_lut_domain():
    """Rows padded with +-LUTmax so that scale == 1 and x == the element itself: every float16
    pattern with |x| <= LUTmax goes through the reference's LUT rounding."""
    arrays = {}
    for fmt, lutmax in (("fp8_e4m3", 480.0), ("fp8_e3m4", 31.0)):
        allv = torch.from_numpy(np.arange(65536, dtype=np.uint16).view(np.int16).copy()).view(torch.float16)
        keep = allv[torch.isfinite(allv) & (allv.abs() <= lutmax)]
        C = 1024
        rows = (keep.numel() + C - 2) // (C - 1)
        w = torch.zeros(rows, C, dtype=torch.float16)
        w[:, 0] = lutmax
        flat = torch.zeros(rows * (C - 1), dtype=torch.float16)
        flat[: keep.numel()] = keep
        w[:, 1:] = flat.reshape(rows, C - 1)
        d, q, s = qw._fp_quant_meta(w, wq_bits=8, datatype=fmt, group_size=None)
        assert torch.all(s == 1.0)
        arrays[f"{fmt}/in"] = bits16(w)
        arrays[f"{fmt}/q"] = bits16(q)
        arrays[f"{fmt}/deq"] = bits16(d)
    np.savez_compressed(os.path.join(HERE, "lut_domain.npz"), **arrays)
    print("lut_domain.npz")


MORE_TABLES = [(4, "fp4"), (5, "fp5_e2m2"), (5, "fp5_e3m1"), (6, "fp6_e2m3"), (6, "fp6_e3m2"), (8, "fp8_e2m5"),
               (3, "int3"), (4, "int4"), (6, "int6")]
ALL_TABLES = [(8, "fp8_e4m3"), (8, "fp8_e3m4"), (5, "fp5"), (6, "fp6"), (5, "int5"), (8, "int8")] + MORE_TABLES
_MAPPING = {3: qw.DATATYPE_MAPPING_3_BIT, 4: qw.DATATYPE_MAPPING_4_BIT, 5: qw.DATATYPE_MAPPING_5_BIT,
            6: qw.DATATYPE_MAPPING_6_BIT, 8: qw.DATATYPE_MAPPING_8_BIT}


LIST_TABLES = [(3, "fp3"), (3, "fp3_er_pos"), (3, "fp3_er_neg"), (3, "fp3_ea_pos"), (3, "fp3_ea_neg"),
               (4, "flint4"), (4, "fp4_er_pos"), (4, "fp4_er_neg"), (4, "fp4_ea_pos"), (4, "fp4_ea_neg"),
               (5, "flint5")]
MX_TABLES = [(3, "mx_int3"), (3, "mx_fp3"), (4, "mx_int4"), (4, "mx_fp4")]
_MX_MAPPING = {3: qw.DATATYPE_MAPPING_3_BIT_MX, 4: qw.DATATYPE_MAPPING_4_BIT_MX}


def gen_lut_tables():
    tables = {name: list(map(float, _MAPPING[bits][name])) for bits, name in ALL_TABLES + LIST_TABLES}
    tables["fp8_e5m2"] = list(map(float, qw.DATATYPE_MAPPING_8_BIT["fp8_e5m2"]))
    tables.update({name: list(map(float, _MX_MAPPING[bits][name])) for bits, name in MX_TABLES})
    with open(os.path.join(HERE, "lut_tables.json"), "w") as f:
        json.dump(tables, f)
    print("lut_tables.json")


def gen_table_formats():
    gen = torch.Generator().manual_seed(11)
    torch.manual_seed(11)
    K, C = 8, 512
    arrays = {}
    inputs = {}
    for name, w in distributions(K, C, gen):
        if name in ("normal", "heavy", "zerorows", "student_t", "tiny"):
            inputs[f"{name}.f16"] = w.half()
        if name == "normal":
            inputs["normal.bf16"], inputs["normal.f32"] = w.bfloat16(), w.float()
    sp = torch.randn(4, 512, generator=gen) * 0.02
    sp[0, 3], sp[0, 7] = float("inf"), float("-inf")
    sp[1, 5] = float("-inf")
    sp[2, 9] = float("nan")
    inputs["specials.f16"] = sp.half()
    bits = np.arange(65536, dtype=np.uint16).reshape(256, 256)
    allf16 = torch.from_numpy(bits.view(np.int16).copy()).view(torch.float16)
    inputs["exhaustive.f16"] = torch.where(torch.isfinite(allf16), allf16, torch.zeros_like(allf16))
    # float32 weights whose group maxima sit 1 .. 3 float32 steps under a power of two: there float32 log2 rounds
    # up to the integer and quant_mx's floor(log2(max)) is one more than the exponent field of max (MX only)
    edge = (torch.randn(8, 512, generator=gen) * 0.3).float().clamp(-0.9, 0.9)
    for r in range(8):
        for gidx in range(16):
            k = (r * 16 + gidx) % 25 - 14
            a = np.float32(2.0 ** k)
            for _ in range(1 + (r + gidx) % 3):
                a = np.nextafter(a, np.float32(0))
            edge[r, gidx * 32: (gidx + 1) * 32] *= float(2.0 ** k)
            edge[r, gidx * 32 + (5 * gidx + r) % 32] = float(a) * (-1.0 if (r + gidx) % 2 else 1.0)
    inputs["pow2edge.f32"] = edge
    for name, w in inputs.items():
        arrays[f"in/{name}"] = w.numpy().copy() if name.endswith(".f32") else bits16(w)

    def groups(name, fmt):
        if name.startswith("pow2edge"):
            return []
        if name.startswith("exhaustive"):
            return [None] if fmt in ("flint4", "fp3_ea_neg", "flint5", "fp8_e5m2") else []
        return [None, 64, -1]

    for wq_bits, fmt in LIST_TABLES:
        for name, w in inputs.items():
            for gs in groups(name, fmt):
                d, q, s = qw._fp_quant_meta(w, wq_bits=wq_bits, datatype=fmt, group_size=gs)
                d2 = qw.quant_datatype(w, wq_bits=wq_bits, datatype=fmt, group_size=gs)
                pre = f"out/{name}/{fmt}/{gs_key(gs)}/"
                arrays[pre + "deq"], arrays[pre + "q"] = bits16(d), bits16(q)
                arrays[pre + "scale"] = bits16(s.reshape(-1))
                if gs == -1:  # the only scope where quant_datatype differs (float32 branch)
                    arrays[pre + "deq_quant_datatype"] = bits16(d2.reshape_as(d))
                else:
                    assert torch.equal(bits_t(d2.reshape_as(d)), bits_t(d))
    for name, w in inputs.items():
        for gs in groups(name, "fp8_e5m2"):
            d, q, s = qw._fp_quant_meta(w, wq_bits=8, datatype="fp8_e5m2", group_size=gs)
            d2 = qw.quant_datatype(w, wq_bits=8, datatype="fp8_e5m2", group_size=gs)
            assert d.dtype == q.dtype == s.dtype == torch.float32
            assert np.array_equal(d2.reshape_as(d).numpy().view(np.uint32), d.numpy().view(np.uint32))
            pre = f"out/{name}/fp8_e5m2/{gs_key(gs)}/"
            arrays[pre + "deq"], arrays[pre + "q"] = d.numpy().copy(), q.numpy().copy()
            arrays[pre + "scale"] = s.reshape(-1).numpy().copy()
    for wq_bits, fmt in MX_TABLES:
        for name, w in inputs.items():
            if name.startswith("exhaustive") and fmt not in ("mx_fp4", "mx_int3"):
                continue
            for gs in ([32] if name.startswith("exhaustive") else [32, 128]):
                d = qw.quant_mx(w, wq_bits=wq_bits, datatype=fmt, group_size=gs)
                assert d.dtype == torch.float16
                arrays[f"out/{name}/{fmt}/{gs}/deq"] = bits16(d)
    np.savez_compressed(os.path.join(HERE, "quant_table_formats.npz"), **arrays)
    print("quant_table_formats.npz", len(arrays), "arrays")


def gen_more_formats():
    gen = torch.Generator().manual_seed(7)
    torch.manual_seed(7)
    K, C = 8, 512
    arrays = {}
    inputs = {name: w.half() for name, w in distributions(K, C, gen)
              if name in ("normal", "heavy", "zerorows", "student_t")}
    bits = np.arange(65536, dtype=np.uint16).reshape(256, 256)
    allf16 = torch.from_numpy(bits.view(np.int16).copy()).view(torch.float16)
    inputs["exhaustive"] = torch.where(torch.isfinite(allf16), allf16, torch.zeros_like(allf16))
    for name, w in inputs.items():
        arrays[f"in/{name}"] = bits16(w)
    for wq_bits, fmt in MORE_TABLES:
        for name, w in inputs.items():
            for gs in ([None, 128] if name == "exhaustive" else [None, 64, -1]):
                d, q, s = qw._fp_quant_meta(w, wq_bits=wq_bits, datatype=fmt, group_size=gs)
                d2 = qw.quant_datatype(w, wq_bits=wq_bits, datatype=fmt, group_size=gs)
                pre = f"out/{name}/{fmt}/{gs_key(gs)}/"
                arrays[pre + "deq"], arrays[pre + "q"] = bits16(d), bits16(q)
                arrays[pre + "scale"] = bits16(s.reshape(-1))
                arrays[pre + "deq_quant_datatype"] = bits16(d2.reshape_as(d))
        # rounding-domain sweep: rows padded with the table maximum, so scale == 1 and x is the element itself
        lutmax = float(max(abs(v) for v in _MAPPING[wq_bits][fmt]))
        keep = allf16.reshape(-1)
        keep = keep[torch.isfinite(keep) & (keep.abs() <= lutmax)]
        Cd = 1024
        rows = (keep.numel() + Cd - 2) // (Cd - 1)
        w = torch.zeros(rows, Cd, dtype=torch.float16)
        w[:, 0] = lutmax
        flat = torch.zeros(rows * (Cd - 1), dtype=torch.float16)
        flat[: keep.numel()] = keep
        w[:, 1:] = flat.reshape(rows, Cd - 1)
        d, q, s = qw._fp_quant_meta(w, wq_bits=wq_bits, datatype=fmt, group_size=None)
        assert torch.all(s == 1.0)
        arrays[f"domain/{fmt}/in"], arrays[f"domain/{fmt}/q"] = bits16(w), bits16(q)
    for asym in (False, True):
        fmt = "int4_asym" if asym else "int4"
        for name, w in inputs.items():
            for gs in ([None, 128] if name == "exhaustive" else [None, 64, -1]):
                d, q, s, zp = qw._int_quant_meta(w, wq_bits=4, asym=asym, group_size=gs)
                pre = f"out/{name}/{fmt}.int/{gs_key(gs)}/"
                arrays[pre + "deq"] = bits16(d)
                arrays[pre + "q"] = q.contiguous().view(torch.uint8).numpy().copy()
                arrays[pre + "scale"] = bits16(s.reshape(-1))
                if zp is not None:
                    arrays[pre + "zp"] = bits16(zp.reshape(-1))
    np.savez_compressed(os.path.join(HERE, "quant_more_formats.npz"), **arrays)
    print("quant_more_formats.npz", len(arrays), "arrays")


def gen_specials():
    """Non-finite and one-sided rows through the reference's int8 / int8_asym / fp8 paths (the other fixture files
    hold them only for the explicit-list formats)."""
    gen = torch.Generator().manual_seed(21)
    w = torch.randn(16, 512, generator=gen) * 0.02
    w[0, 3], w[0, 7] = float("inf"), float("-inf")
    w[1, 5] = float("-inf")
    w[2, 9] = float("nan")
    w[3, 100] = float("inf")
    w[4] = w[4].abs() + 0.01          # all positive
    w[5] = -w[5].abs() - 0.01         # all negative
    w[6] = 0                          # all zero
    w[7] = 0.0123                     # constant
    w[8, :64] = w[8, :64].abs()       # one group on one side of zero
    w[9, 64:128] = -w[9, 64:128].abs() - 1.0
    w[10, 17] = 65504.0               # float16 maximum
    w[11, 17] = -65504.0
    w[12] = w[12] * 1e-4              # subnormal-sized values
    w[13, 40] = float("nan")
    w[13, 300] = float("inf")
    arrays = {"in": bits16(w.half())}
    for fmt in FORMATS:
        for gs in [None, 64, -1]:
            for k, v in run_reference(w.half(), fmt, gs).items():
                arrays[f"out/{fmt}/{gs_key(gs)}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "quant_specials.npz"), **arrays)
    print("quant_specials.npz", len(arrays), "arrays")


def gen_zero_sign():
    """Segments whose minimum is a zero and that hold +0 and -0: _int_quant_meta takes rmin = w.amin(-1)
    (quant_weight.py:179-183) and zero-point = clamp(round(-rmin / scale)), so the zero amin returns gives the sign of
    the zero-point.  torch's reduction meets up to 32 elements in memory order (the first zero wins: 200 of 200 trials
    per length, checked when this was written); longer segments go through vector lanes and the winner depends on the
    positions."""
    gen = torch.Generator().manual_seed(77)
    rows, cols = 64, 256
    w = (torch.randn(rows, cols, generator=gen).abs() * 0.02 + 0.001).half()
    sign = torch.randint(0, 2, (rows, cols), generator=gen).bool()
    place = torch.rand(rows, cols, generator=gen) < 0.3
    w[place & sign] = 0.0
    w[place & ~sign] = -0.0
    for r in range(rows):                       # every segment of 8 holds both zeros, in either order
        for c in range(0, cols, 8):
            i, j = torch.randperm(8, generator=gen)[:2].tolist()
            w[r, c + i], w[r, c + j] = 0.0, -0.0
    w[48:56] = torch.where(w[48:56] == 0, w[48:56], -w[48:56])      # all-negative rows: the maximum is the zero
    w[56:, ::5] = -w[56:, ::5].abs() - 0.004                           # rows where the zeros are not the extreme
    arrays = {"in": bits16(w)}
    for bits in (8, 4):
        for gs in [8, 16, 32, 64, 128, None, -1]:
            d, q, s, zp = qw._int_quant_meta(w, wq_bits=bits, asym=True, group_size=gs)
            d2, q2, s2, zp2 = qw._int_quant_meta(w.clone(), wq_bits=bits, asym=True, group_size=gs)
            pre = f"out/int{bits}_asym/{gs_key(gs)}/"
            arrays[pre + "deq"] = bits16(d)
            arrays[pre + "q"] = q.contiguous().view(torch.uint8).numpy().copy()
            arrays[pre + "scale"] = bits16(s.reshape(-1))
            arrays[pre + "zp"] = bits16(zp.reshape(-1))
            arrays[pre + "zp_again"] = bits16(zp2.reshape(-1))     # a second call on a copy at another address
    np.savez_compressed(os.path.join(HERE, "quant_zero_sign.npz"), **arrays)
    print("quant_zero_sign.npz", len(arrays), "arrays")


MIXED = ["mixed_bitmod", "mixed_er", "mixed_ea", "mixed_ant"]


def gen_search():
    gen = torch.Generator().manual_seed(13)
    torch.manual_seed(13)
    arrays = {}
    inputs = {name: w.half() for name, w in distributions(8, 512, gen)
              if name in ("normal", "heavy", "zerorows", "student_t", "tiny", "big", "outlier")}
    inputs["normal_wide"] = (torch.randn(12, 4096, generator=gen) * 0.02).half()
    for name, w in inputs.items():
        arrays[f"in/{name}"] = bits16(w)
        for wq_bits in (3, 4):
            for kind in MIXED:
                for gs in (None, 64):
                    d = qw.search_datatype(w, wq_bits=wq_bits, datatype=kind, group_size=gs)
                    assert d.dtype == torch.float16 and d.shape == w.shape
                    arrays[f"out/{name}/{wq_bits}/{kind}/{gs_key(gs)}"] = bits16(d)
    np.savez_compressed(os.path.join(HERE, "quant_search.npz"), **arrays)
    print("quant_search.npz", len(arrays), "arrays")
    # what the reference does with weights that are not float16: the exception type and message per dtype
    errors = {}
    for dt in (torch.bfloat16, torch.float32, torch.float64):
        w = (torch.randn(8, 256, generator=gen) * 0.02).to(dt)
        for gs in (None, 64):
            try:
                qw.search_datatype(w, wq_bits=4, datatype="mixed_bitmod", group_size=gs)
                errors[f"{str(dt).split('.')[-1]}/{gs_key(gs)}"] = None
            except Exception as e:   # noqa: BLE001 -- the point is to record whatever it raises
                errors[f"{str(dt).split('.')[-1]}/{gs_key(gs)}"] = [type(e).__name__, str(e)]
    with open(os.path.join(HERE, "search_errors.json"), "w") as f:
        json.dump(errors, f, indent=1, sort_keys=True)
    print("search_errors.json", errors)


def gen_onnx():
    """ONNX files written by PyTorch's own exporter (the TorchScript exporter's C++ serializer, the writer behind the
    reference's export_and_quant.py; its internals are called directly because torch.onnx.export() insists on the
    `onnx` package, which is not installed): fixtures for the initializer reader that do not come from this
    repository's test writer.  Stored: the model files and, per file, every parameter's bits by name."""
    import warnings
    import torch.nn as nn
    from torch.onnx._internal.torchscript_exporter import _globals as G, utils as U
    warnings.filterwarnings("ignore")
    out_dir = os.path.join(HERE, "onnx")
    os.makedirs(out_dir, exist_ok=True)

    class Block(nn.Module):   # Linear layers with and without bias, an embedding (Gather), a weight used twice
        def __init__(self):
            super().__init__()
            self.embed = nn.Embedding(50, 32)
            self.q_proj = nn.Linear(32, 32, bias=False)
            self.up = nn.Linear(32, 96)
            self.down = nn.Linear(96, 32, bias=False)
            self.norm = nn.LayerNorm(32)

        def forward(self, ids):
            h = self.embed(ids)
            h = h + self.q_proj(self.norm(h))
            return self.down(torch.relu(self.up(h)))

    arrays, meta = {}, {}
    # (file, dtype, constant folding, the reference's quant_model arguments or None): the last two are the
    # reference's own quantized exports of the first model -- _LinearDump modules with weight_q / weight_scale /
    # weight_zp buffers -- in the directory layout quant_error_multi_weight.py parses (:41-52)
    cases = (("block_f16", torch.float16, False, None), ("block_f32_folded", torch.float32, True, None),
             ("tree/per-group/int8_asym/w_8_gs_32/model", torch.float16, False, (8, "int8_asym", 32)),
             ("tree/per-channel/fp8_e4m3/w_8_gs_none/model", torch.float16, False, (8, "fp8_e4m3", None)))
    for name, dtype, fold, qcfg in cases:
        torch.manual_seed(23)
        m = Block().to(dtype).eval()
        if qcfg is not None:
            qw.time.sleep = lambda *_: None      # (it only paces the prints)
            with contextlib.redirect_stdout(io.StringIO()):
                qw.quant_model(m, *qcfg)
        os.makedirs(os.path.dirname(os.path.join(out_dir, name)), exist_ok=True)
        ids = torch.randint(0, 50, (2, 7))
        G.GLOBALS.export_onnx_opset_version = 17
        G.GLOBALS.operator_export_type = torch.onnx.OperatorExportTypes.ONNX
        with torch.no_grad():
            graph, params, _ = U._model_to_graph(m, (ids,), False, ["ids"], ["y"], torch.onnx.OperatorExportTypes.ONNX,
                                                 do_constant_folding=fold, training=torch.onnx.TrainingMode.EVAL,
                                                 dynamic_axes={})
        proto = graph._export_onnx(params, 17, {}, False, torch.onnx.OperatorExportTypes.ONNX, True, False, {}, True,
                                   "", {})[0]
        with open(os.path.join(out_dir, name + ".onnx"), "wb") as f:
            f.write(proto)
        for k, v in params.items():
            a = v.detach().cpu()
            arrays[f"{name}/{k}"] = a.view(torch.int16).numpy() if a.dtype == torch.float16 else a.numpy()
        # MatMul / Gemm weight operands as the graph names them (second input, possibly through one Transpose)
        weights = []
        producers = {o.debugName(): n for n in graph.nodes() for o in n.outputs()}
        for n in graph.nodes():
            if n.kind() in ("onnx::MatMul", "onnx::Gemm"):
                w = list(n.inputs())[1]
                src = producers.get(w.debugName())
                if src is not None and src.kind() == "onnx::Transpose":
                    w = list(src.inputs())[0]
                if w.debugName() in params:
                    weights.append(w.debugName())
        meta[name] = {"bytes": len(proto), "sha256": hashlib.sha256(proto).hexdigest(), "params": sorted(params),
                      "matmul_gemm_weights": sorted(set(weights)), "torch": torch.__version__}
    np.savez_compressed(os.path.join(out_dir, "params.npz"), **arrays)
    with open(os.path.join(out_dir, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("onnx/", meta)


API_ERROR_CASES = [
    # (function, positional / keyword arguments after the weight, weight shape)
    ("quant_int", dict(wq_bits=8, group_size=48), (4, 128)),
    ("quant_int", dict(wq_bits=8, group_size=32), (128,)),
    ("quant_int", dict(wq_bits=8, group_size=None), (128,)),
    ("quant_int", dict(wq_bits=8, group_size=-1), (128,)),
    ("quant_int", dict(wq_bits=8, group_size=32), (2, 4, 64)),
    ("quant_int_asym", dict(wq_bits=4, group_size=100), (4, 128)),
    ("_int_quant_meta", dict(wq_bits=8, asym=True, group_size=48), (4, 128)),
    ("_int_quant_meta", dict(wq_bits=8, asym=False, group_size=64), (128,)),
    ("quant_datatype", dict(wq_bits=7, datatype="fp8_e4m3", group_size=None), (4, 128)),
    ("quant_datatype", dict(wq_bits=8, datatype="nonsense", group_size=None), (4, 128)),
    ("quant_datatype", dict(wq_bits=4, datatype="fp8_e4m3", group_size=None), (4, 128)),
    ("quant_datatype", dict(wq_bits=8, datatype="fp8_e4m3", group_size=48), (4, 128)),
    ("quant_datatype", dict(wq_bits=2, datatype="fp8_e5m2", group_size=None), (4, 128)),
    ("_fp_quant_meta", dict(wq_bits=7, datatype="fp8_e4m3", group_size=None), (4, 128)),
    ("_fp_quant_meta", dict(wq_bits=8, datatype="int8_asym", group_size=None), (4, 128)),
    ("_fp_quant_meta", dict(wq_bits=8, datatype="fp8_e3m4", group_size=48), (4, 128)),
    ("quant_mx", dict(wq_bits=5, datatype="mx_fp4", group_size=32), (4, 128)),
    ("quant_mx", dict(wq_bits=4, datatype="mx_fp3", group_size=32), (4, 128)),
    ("quant_mx", dict(wq_bits=4, datatype="mx_fp4", group_size=48), (4, 128)),
    ("search_datatype", dict(wq_bits=5, datatype="mixed_er", group_size=None), (4, 128)),
    ("search_datatype", dict(wq_bits=4, datatype="mixed_er", group_size=48), (4, 128)),
]


def gen_api_errors():
    """What the reference raises (or returns) for arguments outside its domain: exception type and message per case,
    or the result's dtype / shape when it goes through."""
    out = []
    for fn, kw, shape in API_ERROR_CASES:
        torch.manual_seed(1)
        w = (torch.randn(*shape) * 0.02).half()
        entry = {"fn": fn, "kwargs": kw, "shape": list(shape)}
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                r = getattr(qw, fn)(w, **kw)
            r0 = r[0] if isinstance(r, tuple) else r
            entry["ok"] = {"dtype": str(r0.dtype), "shape": list(r0.shape)}
        except Exception as e:   # noqa: BLE001 -- the point is to record whatever it raises
            entry["raises"] = [type(e).__name__, str(e)]
        out.append(entry)
        print(fn, kw, shape, "->", entry.get("raises") or entry.get("ok"))
    with open(os.path.join(HERE, "api_errors.json"), "w") as f:
        json.dump(out, f, indent=1)


QUANT_MODEL_CASES = [("int8", 8), ("int8_asym", 8), ("int4_asym", 4), ("fp8_e4m3", 8), ("fp4", 4)]
QUANT_MODEL_GROUPS = [None, 64, -1]


def gen_quant_model():
    """The reference's quant_model on a small model (time.sleep patched out: it only paces the prints)."""
    import contextlib
    import io
    import torch.nn as nn
    qw.time.sleep = lambda *_: None
    out = {}
    torch.manual_seed(1234)
    base = nn.Sequential(nn.Linear(256, 128), nn.Linear(128, 256, bias=False), nn.Linear(256, 128), nn.Linear(128, 64)).half()
    base[2].weight.data[3, 5] = 0.75          # an outlier
    base[3].weight.data[1] = 0                # a zero row
    for i, m in enumerate(base):
        out[f"in/{i}/weight"] = bits16(m.weight.data)
        out[f"in/{i}/has_bias"] = np.array([m.bias is not None])
    for dt, bits in QUANT_MODEL_CASES:
        for gs in QUANT_MODEL_GROUPS:
            import copy
            model = copy.deepcopy(base)
            with contextlib.redirect_stdout(io.StringIO()):
                qw.quant_model(model, bits, dt, gs)
            key = f"{dt}/{gs_key(gs)}"
            out[f"{key}/classes"] = np.array([type(m).__name__ for m in model])
            for i, m in enumerate(model):
                out[f"{key}/{i}/weight"] = bits16(m.weight.data)
                names = sorted(n for n, _ in m.named_buffers())
                out[f"{key}/{i}/buffers"] = np.array(names)
                for n in names:
                    b = getattr(m, n)
                    out[f"{key}/{i}/{n}/bytes"] = b.contiguous().reshape(-1).view(torch.uint8).numpy().copy()
                    out[f"{key}/{i}/{n}/dtype"] = np.array([str(b.dtype)])
                    out[f"{key}/{i}/{n}/shape"] = np.array(list(b.shape), dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "quant_model.npz"), **out)
    print("quant_model.npz", len(out), "arrays")


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gen_error_stats():
    """Known answers for the error half.  Inputs come from numpy's PCG64 stream so the tests can
    regenerate them; the quantized tensors come from the imported reference; the statistics are the
    analysis scripts' numpy expressions (restated in oracle/oracle.py -- those scripts need the
    `onnx` package to import, which this image does not have)."""
    K, C = 512, 1024
    rng = np.random.default_rng(1)
    cases = {"normal": (rng.standard_normal((K, C)) * 0.02).astype(np.float16),
             "student_t": (rng.standard_t(3, (K, C)) * 0.02).astype(np.float16)}
    res = {"K": K, "C": C, "seed": 1, "cases": {}}
    for cname, w16 in cases.items():
        w = torch.from_numpy(w16)
        res["cases"][cname] = {"input_sha256": digest(w16), "formats": {}}
        for fmt in FORMATS:
            for gs in [None, 32, 64, 128]:
                ref = run_reference(w, fmt, gs)
                deq = ref["deq"].view(np.float16).reshape(K, C)
                a32, b32 = orc.proto_to_array(w16), orc.proto_to_array(deq)
                grouping = "per-channel" if gs is None else "per-group"
                multi = orc.compute_errors_multi(a32, b32, grouping, gs)
                single = orc.compute_errors_single(a32, b32, grouping, gs)
                el = orc.elementwise_f32(w16, deq)
                sl = orc.sliced_stats(el["err"], 1000, 200000)
                bk = orc.bucketize(w16, deq)
                entry = {
                    "deq_sha256": digest(ref["deq"]), "q_sha256": digest(ref["q"]),
                    "scale_sha256": digest(ref["scale"]),
                    "mse_tensor": multi["mse_tensor"],
                    "multi": {"mse_rep": multi["mse_rep"], "mse_p99": multi["mse_p99"],
                              "per_vec_sha256": digest(multi["per_vec"])},
                    "single": {"mse_rep": single["mse_rep"], "mse_p99": single["mse_p99"],
                               "per_vec_sha256": digest(single["per_vec"])},
                    "abs_p99_f32_bits": int(np.float32(el["p99"]).view(np.uint32)),
                    "abs_sum_f32_bits": int(np.float32(el["sum"]).view(np.uint32)),
                    "abs_sha256": digest(el["abs"]),
                    "sliced_mse_f32_bits": int(np.float32(sl["mse"]).view(np.uint32)),
                    "sliced_abs_sum_f32_bits": int(np.float32(sl["abs_sum"]).view(np.uint32)),
                    "bucket_idx_sha256": digest(bk["idx"].astype(np.int64)),
                    "abs_f16_sha256": digest(bk["abs_err"]),
                }
                if "zp" in ref:
                    entry["zp_sha256"] = digest(ref["zp"])
                res["cases"][cname]["formats"][f"{fmt}/{gs_key(gs)}"] = entry
        # compare.py on the three 8-bit formats, per-channel
        qs = [run_reference(w, f, None)["deq"].view(np.float16).reshape(K, C) for f in ("int8", "fp8_e4m3", "fp8_e3m4")]
        cmp_ = orc.compare_formats(w16, qs, threshold=0.02)
        cmp_lo = orc.compare_formats(w16, qs, threshold=1e-4)
        res["cases"][cname]["compare"] = {
            "formats": ["int8", "fp8_e4m3", "fp8_e3m4"],
            "min_err_sha256": digest(cmp_["min_err"]),
            "thr_0.02": {"n_out": int(cmp_["mask_out"].sum()), "n_best": [int(b.sum()) for b in cmp_["best"]],
                         "best_sha256": [digest(np.packbits(b)) for b in cmp_["best"]]},
            "thr_1e-4": {"n_out": int(cmp_lo["mask_out"].sum()), "n_best": [int(b.sum()) for b in cmp_lo["best"]],
                         "out_sha256": digest(np.packbits(cmp_lo["mask_out"])),
                         "n_ranges_out": len(orc.to_ranges(np.where(cmp_lo["mask_out"])[0]))},
        }
    with open(os.path.join(HERE, "error_stats.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("error_stats.json")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "more":   # only the files added for the extra formats
        gen_lut_tables()
        gen_more_formats()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "model":
        gen_quant_model()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "specials":
        gen_specials()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "zero_sign":
        gen_zero_sign()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "api":
        gen_api_errors()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "onnx":
        gen_onnx()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "search":
        gen_search()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tables":  # only the explicit-list formats
        gen_lut_tables()
        gen_table_formats()
        sys.exit(0)
    gen_lut_tables()
    gen_more_formats()
    gen_table_formats()
    gen_search()
    gen_specials()
    gen_zero_sign()
    gen_quant_model()
    gen_quant_small()
    gen_exhaustive()
    gen_lut_domain()
    gen_error_stats()
