"""CPU: the host half of the sweep launches -- mq_sweep_prepare validates and lays out the launch table without
touching a device, so its rules are checked here with made-up (aligned) addresses: block ranges that tile the grid,
per-tensor segment counts, and the status code for everything the sweep form does not take."""
import ctypes
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _lib  # noqa: E402


class _Item(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("out", _lib.QuantOut)]


ENTRY = np.dtype([("w", "<u8"), ("scale", "<u8"), ("zp", "<u8"), ("seg_sse", "<u8"), ("totals", "<u8"),
                  ("nseg", "<i8"), ("blk0", "<u4"), ("nblk", "<u4"), ("pad", "<i8")])


@pytest.fixture(scope="module")
def L():
    lib = _lib.lib()
    i32, i64, vp, sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t
    lib.mq_sweep_table_bytes.restype = sz
    lib.mq_sweep_table_bytes.argtypes = [i32]
    lib.mq_sweep_workspace_bytes.restype = sz
    lib.mq_sweep_workspace_bytes.argtypes = [i64]
    lib.mq_sweep_prepare.restype = ctypes.c_int
    lib.mq_sweep_prepare.argtypes = [ctypes.POINTER(_Item), ctypes.POINTER(i64), ctypes.POINTER(i64), i32, i32,
                                     ctypes.POINTER(_lib.QuantCfg), vp, sz, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    return lib


def _prepare(L, shapes, group_size, dtype=_lib.BF16, fmt=_lib.INT_SYM, bits=8, w_base=0x10000000, seg_flag=1,
             table_bytes=None, **out):
    n = len(shapes)
    items = (_Item * n)()
    for j in range(n):
        items[j].w = w_base + j * 0x4000000
        items[j].out.seg_sse = 0x7000000000 + j * 0x1000000
        items[j].out.totals = 0x7800000000 + j * 16
        items[j].out.seg_mse_f32 = seg_flag
        for k, v in out.items():
            setattr(items[j].out, k, v)
    rows = (ctypes.c_int64 * n)(*[s[0] for s in shapes])
    cols = (ctypes.c_int64 * n)(*[s[1] for s in shapes])
    cfg = _lib.QuantCfg(format=fmt, bits=bits, group_size=0 if group_size is None else group_size,
                        exp_bits=4, man_bits=3, exp_bias=7)
    tab = np.zeros(n, dtype=ENTRY)
    assert L.mq_sweep_table_bytes(n) == tab.nbytes == 64 * n
    seg_len, blocks = ctypes.c_int64(-1), ctypes.c_int64(-1)
    rc = L.mq_sweep_prepare(items, rows, cols, n, dtype, ctypes.byref(cfg), tab.ctypes.data,
                            tab.nbytes if table_bytes is None else table_bytes, ctypes.byref(seg_len), ctypes.byref(blocks))
    return rc, tab, seg_len.value, blocks.value, items


def test_block_ranges_tile_the_grid_and_follow_the_sizes(L):
    shapes = [(4096, 4096), (1024, 4096), (14336, 4096), (4096, 14336), (8, 4096), (128256, 4096)]
    for gs in (32, 64, 128, 256, 512):
        rc, tab, seg_len, blocks, items = _prepare(L, shapes, gs)
        assert rc == _lib.OK and seg_len == gs
        assert tab["blk0"][0] == 0 and np.array_equal(tab["blk0"][1:], np.cumsum(tab["nblk"])[:-1])
        assert blocks == int(tab["nblk"].sum()) and (tab["nblk"] >= 1).all()
        assert np.array_equal(tab["nseg"], [r * c // gs for r, c in shapes])
        assert [int(x) for x in tab["w"]] == [items[j].w for j in range(len(shapes))]
        assert [int(x) for x in tab["totals"]] == [items[j].out.totals for j in range(len(shapes))]
        # one workgroup per 128 KiB of (2-byte) weights, rounded up per tensor
        want = [-(-(r * c * 2) // (128 << 10)) for r, c in shapes]
        assert np.array_equal(tab["nblk"], want), gs
        assert L.mq_sweep_workspace_bytes(blocks) == blocks * 16
    # per-channel: one column count per launch, any row counts; rows of 8 KiB -> 16 rows per workgroup
    rc, tab, seg_len, blocks, _ = _prepare(L, [(4096, 4096), (1000, 4096), (3, 4096)], None)
    assert rc == _lib.OK and seg_len == 4096 and list(tab["nseg"]) == [4096, 1000, 3]
    assert list(tab["nblk"]) == [256, 63, 1] and blocks == 320
    # rows of 28 KiB (one workgroup per row and step): 4 rows per workgroup
    rc, tab, seg_len, blocks, _ = _prepare(L, [(4096, 14336), (5, 14336)], None)
    assert rc == _lib.OK and seg_len == 14336 and list(tab["nblk"]) == [1024, 2]


def test_what_the_sweep_form_refuses(L):
    ok = [(256, 4096), (512, 4096)]
    U, A, G = _lib.ERR_UNSUPPORTED, _lib.ERR_ARG, _lib.ERR_GROUP
    assert _prepare(L, ok, 128)[0] == _lib.OK
    assert _prepare(L, ok, 128, dtype=_lib.F16)[0] == _lib.OK
    assert _prepare(L, ok, 128, dtype=_lib.F32)[0] == U                   # float32 weights
    assert _prepare(L, ok, -1)[0] == U                                    # per-tensor
    for gs in (16, 48, 96, 384):                                          # short groups that are not 32 * 2^k
        assert _prepare(L, [(256, 3072)], gs)[0] == U, gs
    assert _prepare(L, [(256, 3072)], 1024)[0] == _lib.OK                 # long groups go through the row kernels
    assert _prepare(L, ok, 128, w_base=0x10000008)[0] == U                # weights off the 16-byte grid
    assert _prepare(L, ok, 128, deq=0x9000000000)[0] == U                 # element-sized outputs: not statistics only
    assert _prepare(L, ok, 128, abs_err=0x9000000000)[0] == U
    assert _prepare(L, [(256, 4096), (256, 4160)], 128)[0] == G           # a column count the group does not divide
    assert _prepare(L, [(256, 4096), (256, 8192)], None)[0] == A          # per-channel: one column count per launch
    assert _prepare(L, [(256, 200)], None)[0] == U                        # short rows that are not 32 * 2^k elements
    assert _prepare(L, [(4, 65536)], None)[0] == U                        # ... and longer
    assert _prepare(L, [(60000, 40960)], 128)[0] == U                     # past the 32-bit offsets of the buffer loads
    assert _prepare(L, ok, 128, seg_flag=2)[0] == _lib.OK                 # numpy-order entries: groups of 32 / 64 / 128
    assert _prepare(L, ok, 256, seg_flag=2)[0] == U
    assert _prepare(L, ok, None, seg_flag=2)[0] == U
    assert _prepare(L, ok, 128, seg_flag=3)[0] == A
    assert _prepare(L, ok, 128, fmt=_lib.INT_SYM, bits=9)[0] == U
    assert _prepare(L, ok, 128, fmt=_lib.LUT_FP)[0] == _lib.OK
    assert _prepare(L, ok, 128, table_bytes=64)[0] == _lib.ERR_WORKSPACE
    assert _prepare(L, ok, 128, seg_sse=0x7000000004 + 0, seg_flag=0)[0] == _lib.ERR_ALIGN   # float64 sums, 4-byte aligned
    assert _prepare(L, [(0, 4096)], 128)[0] == A
