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
