"""Loaders for the committed golden vectors (tests/golden), shared by the CPU and GPU tests."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FORMATS = ["int8", "int8_asym", "fp8_e4m3", "fp8_e3m4"]
GROUPS = [None, 32, 64, 128, -1]
_CACHE = {}


def gs_key(gs):
    return "none" if gs is None else str(gs)


def npz(name):
    if name not in _CACHE:
        _CACHE[name] = dict(np.load(os.path.join(GOLDEN, name)))
    return _CACHE[name]


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def small_inputs():
    """-> list of (key, dtype_name, array) for every input of quant_small.npz"""
    z = npz("quant_small.npz")
    return [(k[3:], k.rsplit(".", 1)[1], z[k]) for k in sorted(z) if k.startswith("in/")]


def small_expected(key, fmt, gs):
    z = npz("quant_small.npz")
    pre = f"out/{key}/{fmt}/{gs_key(gs)}/"
    return {k[len(pre):]: z[k] for k in z if k.startswith(pre)}


def exhaustive_expected(fmt, gs):
    z = npz("quant_exhaustive.npz")
    pre = f"out/{fmt}/{gs_key(gs)}/"
    return z["in"], {k[len(pre):]: z[k] for k in z if k.startswith(pre)}


def specials_expected(fmt, gs):
    """quant_specials.npz: rows with +-inf / NaN, one-sided rows, constant rows, float16 limits."""
    z = npz("quant_specials.npz")
    pre = f"out/{fmt}/{gs_key(gs)}/"
    return z["in"], {k[len(pre):]: z[k] for k in z if k.startswith(pre)}


def same_f16_bits(got, exp):
    """Bit equality of two float16 bit arrays where every NaN equals every NaN (payloads and signs of NaNs are
    whatever the producing instruction leaves)."""
    got, exp = np.asarray(got).reshape(-1), np.asarray(exp).reshape(-1)
    gn, en = (got & 0x7fff) > 0x7c00, (exp & 0x7fff) > 0x7c00
    return bool(np.array_equal(gn, en) and np.array_equal(got[~gn], exp[~en]))


# formats beyond the 8-bit headline set (quant_more_formats.npz): (wq_bits, table name)
MORE_TABLES = [(4, "fp4"), (5, "fp5_e2m2"), (5, "fp5_e3m1"), (6, "fp6_e2m3"), (6, "fp6_e3m2"), (8, "fp8_e2m5"),
               (3, "int3"), (4, "int4"), (6, "int6")]


def more_cases(fmt_key):
    """-> list of (input name, array, group size, expected dict) stored for one format key
    ('fp4', ..., or 'int4.int' / 'int4_asym.int' for the 4-bit INT paths)."""
    z = npz("quant_more_formats.npz")
    out = []
    for k in sorted(z):
        if not k.startswith("out/") or not k.endswith("/deq"):
            continue
        _, name, fmt, gk, _ = k.split("/")
        if fmt != fmt_key:
            continue
        pre = f"out/{name}/{fmt}/{gk}/"
        exp = {kk[len(pre):]: z[kk] for kk in z if kk.startswith(pre)}
        out.append((name, z[f"in/{name}"], None if gk == "none" else int(gk), exp))
    return out


# explicit-list formats (quant_table_formats.npz): (wq_bits, table name)
LIST_TABLES = [(3, "fp3"), (3, "fp3_er_pos"), (3, "fp3_er_neg"), (3, "fp3_ea_pos"), (3, "fp3_ea_neg"),
               (4, "flint4"), (4, "fp4_er_pos"), (4, "fp4_er_neg"), (4, "fp4_ea_pos"), (4, "fp4_ea_neg"),
               (5, "flint5")]
MX_TABLES = [(3, "mx_int3"), (3, "mx_fp3"), (4, "mx_int4"), (4, "mx_fp4")]


def table_cases(fmt):
    """-> list of (input name 'dist.dtype', array, group size, expected dict) of quant_table_formats.npz for one
    format name (a LIST_TABLES / MX_TABLES name or 'fp8_e5m2')."""
    z = npz("quant_table_formats.npz")
    out = []
    for k in sorted(z):
        if not k.startswith("out/") or not k.endswith("/deq"):
            continue
        _, name, f, gk, _ = k.split("/")
        if f != fmt:
            continue
        pre = f"out/{name}/{f}/{gk}/"
        exp = {kk[len(pre):]: z[kk] for kk in z if kk.startswith(pre)}
        out.append((name, z[f"in/{name}"], None if gk == "none" else int(gk), exp))
    return out


def same_values(got, want):
    """Bitwise equality of two float16-bit (uint16) or float32 arrays, any NaN matching any NaN (the payload
    and sign of a NaN produced by 0 * inf or inf / inf differ between CPUs and the GPU)."""
    got, want = np.asarray(got).reshape(-1), np.asarray(want).reshape(-1)
    if got.dtype == np.uint16:
        gn, wn = (got & 0x7FFF) > 0x7C00, (want & 0x7FFF) > 0x7C00
        return bool(np.array_equal(gn, wn) and np.array_equal(got[~gn], want[~wn]))
    gn, wn = np.isnan(got), np.isnan(want)
    return bool(np.array_equal(gn, wn) and np.array_equal(got[~gn].view(np.uint32), want[~wn].view(np.uint32)))


def search_cases():
    """-> list of (input name, float16 bits, wq_bits, mixed name, group size, expected deq bits) of
    quant_search.npz (the reference's search_datatype)."""
    z = npz("quant_search.npz")
    out = []
    for k in sorted(z):
        if k.startswith("out/"):
            _, name, bits, kind, gk = k.split("/")
            out.append((name, z[f"in/{name}"], int(bits), kind, None if gk == "none" else int(gk), z[k]))
    return out
