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


# (after the block range: the fields of the numpy-order and list-feeding launches, empty in a table of mq_sweep_prepare)
ENTRY = np.dtype([("w", "<u8"), ("scale", "<u8"), ("zp", "<u8"), ("seg_sse", "<u8"), ("totals", "<u8"),
                  ("nseg", "<i8"), ("blk0", "<u4"), ("nblk", "<u4"), ("leaf", "<u8"), ("lchunk0", "<u4"), ("pad", "<u4"),
                  ("feed", "<u8"), ("blo", "<u4"), ("bhi", "<u4"), ("wg_cnt", "<u8"), ("slots", "<u8"), ("sample", "<u8"),
                  ("sunit", "<u4"), ("scount", "<u4")])


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
    assert L.mq_sweep_table_bytes(n) == tab.nbytes == 120 * n
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


def test_numpy_order_entry_points_refuse_bad_arguments_before_any_launch(L):
    """The argument checks of mq_list_mean_numpy, mq_slice_mean_numpy_f32 / mq_slice_sum_numpy_f32 and
    mq_tensor_mse_numpy come before the first HIP call: their status codes can be checked without a device."""
    i32, i64, vp, sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t
    L.mq_list_mean_numpy.restype = ctypes.c_int
    L.mq_list_mean_numpy.argtypes = [ctypes.POINTER(vp), i32, i64, ctypes.POINTER(vp), vp]
    for fn in (L.mq_slice_mean_numpy_f32, L.mq_slice_sum_numpy_f32):
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.POINTER(vp), i32, i64, i64, i64, ctypes.POINTER(vp), vp]
    L.mq_tensor_mse_numpy_workspace_bytes.restype = sz
    L.mq_tensor_mse_numpy_workspace_bytes.argtypes = [i64, i64]
    L.mq_tensor_mse_numpy.restype = ctypes.c_int
    L.mq_tensor_mse_numpy.argtypes = [vp, i32, i64, i64, ctypes.POINTER(_lib.QuantCfg), vp, vp, vp, vp, sz, vp]
    A, U, G = _lib.ERR_ARG, _lib.ERR_UNSUPPORTED, _lib.ERR_GROUP
    xs = (vp * 2)(0x10000000, 0x10001000)
    out = (vp * 2)(0x20000000, 0x20000008)
    assert L.mq_list_mean_numpy(xs, 2, 0, out, None) == A                    # empty lists have no mean
    assert L.mq_list_mean_numpy(None, 2, 100, out, None) == A
    assert L.mq_list_mean_numpy(xs, 0, 100, out, None) == _lib.OK            # nothing to do
    bad = (vp * 2)(0x10000002, 0x10001000)
    assert L.mq_list_mean_numpy(bad, 2, 100, out, None) == _lib.ERR_ALIGN
    assert L.mq_slice_mean_numpy_f32(xs, 2, 100, 0, 8, out, None) == A       # runs of no values
    assert L.mq_slice_mean_numpy_f32(xs, 2, 100, 30, 64, out, None) == A     # the runs do not tile the slice
    assert L.mq_slice_mean_numpy_f32(xs, 2, 100, 20, 8, out, None) == A      # runs that overlap
    assert L.mq_slice_sum_numpy_f32(xs, 2, -5, 1, 8, out, None) == A
    cfg = _lib.QuantCfg(format=_lib.INT_SYM, bits=8, group_size=128)
    w, sc, mo, ws = 0x30000000, 0x40000000, 0x50000000, 0x60000000
    nb = L.mq_tensor_mse_numpy_workspace_bytes(64, 1024)
    assert nb == 256 + 8 * 4                                                  # 65536 elements: eight chunks of 8192
    call = lambda **k: L.mq_tensor_mse_numpy(k.get("w", w), k.get("dt", _lib.BF16), k.get("rows", 64), k.get("cols", 1024),
                                             ctypes.byref(k.get("cfg", cfg)), k.get("sc", sc), k.get("zp", None), mo,
                                             k.get("ws", ws), k.get("nb", nb), None)
    assert call(dt=_lib.F32) == U and call(dt=7) == A
    assert call(sc=None) == A and call(rows=0) == A
    assert call(cfg=_lib.QuantCfg(format=_lib.INT_ASYM, bits=8, group_size=128)) == A     # asymmetric without zero-points
    assert call(cfg=_lib.QuantCfg(format=_lib.INT_SYM, bits=8, group_size=100)) == G      # 1024 % 100
    assert call(cfg=_lib.QuantCfg(format=_lib.INT_SYM, bits=8, group_size=4)) == U        # a vector would straddle groups
    assert call(w=w + 2) == U                                                              # weights off the 16-byte grid
    assert call(nb=nb - 1) == _lib.ERR_WORKSPACE and call(ws=None) == _lib.ERR_WORKSPACE
    assert call(cfg=_lib.QuantCfg(format=_lib.INT_SYM, bits=9, group_size=128)) == U


# ---- the several-groupings launch (mq_sweep_multi_prepare): host rules, no device ---------------------------------
class _MultiItem(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("list", ctypes.c_void_p * 4), ("totals", ctypes.c_void_p * 4)]


MULTI_ENTRY = np.dtype([("w", "<u8"), ("list", "<u8", (4,)), ("totals", "<u8", (4,)), ("nrow", "<i8"), ("blk0", "<u4"),
                        ("nblk", "<u4")])


def _multi_prepare(L, rows, cols, mask, dtype=_lib.BF16, fmt=_lib.INT_SYM, bits=8, w_base=0x10000000, table_bytes=None,
                   list_base=0x7000000000):
    i32, i64, vp, sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t
    L.mq_sweep_multi_table_bytes.restype = sz
    L.mq_sweep_multi_table_bytes.argtypes = [i32]
    L.mq_sweep_multi_workspace_bytes.restype = sz
    L.mq_sweep_multi_workspace_bytes.argtypes = [i64]
    L.mq_sweep_multi_prepare.restype = ctypes.c_int
    L.mq_sweep_multi_prepare.argtypes = [ctypes.POINTER(_MultiItem), ctypes.POINTER(i64), i64, i32, i32,
                                         ctypes.POINTER(_lib.QuantCfg), ctypes.c_uint32, vp, sz, ctypes.POINTER(i64)]
    n = len(rows)
    items = (_MultiItem * n)()
    for j in range(n):
        items[j].w = w_base + j * 0x8000000
        for c in range(4):
            items[j].list[c] = list_base + (4 * j + c) * 0x1000000
            items[j].totals[c] = 0x7800000000 + (4 * j + c) * 16
    cfg = _lib.QuantCfg(format=fmt, bits=bits, group_size=0, exp_bits=4, man_bits=3, exp_bias=7)
    tab = np.zeros(n, dtype=MULTI_ENTRY)
    assert L.mq_sweep_multi_table_bytes(n) == tab.nbytes == 88 * n
    blocks = ctypes.c_int64(-1)
    rc = L.mq_sweep_multi_prepare(items, (ctypes.c_int64 * n)(*rows), cols, n, dtype, ctypes.byref(cfg), mask,
                                  tab.ctypes.data, tab.nbytes if table_bytes is None else table_bytes, ctypes.byref(blocks))
    return rc, tab, blocks.value, items


def test_multi_table_block_ranges_and_slots(L):
    rows = [4096, 1, 300, 60000]
    for cols in (128, 2048, 4096, 5632, 8192, 14336, 28672, 32768):
        for mask in (15, 3, 12, 9):
            rc, tab, blocks, items = _multi_prepare(L, rows, cols, mask)
            assert rc == _lib.OK, (cols, mask)
            assert tab["blk0"][0] == 0 and blocks == int(tab["nblk"].sum())
            assert np.array_equal(tab["blk0"][1:], np.cumsum(tab["nblk"])[:-1])
            assert np.array_equal(tab["nrow"], rows) and (tab["nblk"] >= 1).all()
            # a workgroup takes about 128 KiB of weights, in whole rows (four at a time where a wave holds a row)
            per = max(1, (128 * 1024) // (cols * 2) // (1 if cols > 8192 else 4)) * (1 if cols > 8192 else 4)
            assert np.array_equal(tab["nblk"], [(r + per - 1) // per for r in rows]), cols
            for c in range(4):   # only the slots of the mask carry pointers
                want = [(items[j].list[c] or 0) if (mask >> c) & 1 else 0 for j in range(len(rows))]
                assert np.array_equal(tab["list"][:, c], want) and ((tab["totals"][:, c] != 0) == bool((mask >> c) & 1)).all()
            assert L.mq_sweep_multi_workspace_bytes(blocks) == blocks * 64


def test_multi_table_refusals(L):
    ok = dict(rows=[64, 64], cols=4096, mask=15)
    assert _multi_prepare(L, **ok)[0] == _lib.OK
    assert _multi_prepare(L, [64], 4000, 15)[0] == _lib.ERR_UNSUPPORTED            # not a multiple of 128 columns
    assert _multi_prepare(L, [64], 64, 15)[0] == _lib.ERR_UNSUPPORTED              # shorter than a group of 128
    assert _multi_prepare(L, [64], 32768 + 128, 15)[0] == _lib.ERR_UNSUPPORTED     # longer than a workgroup holds
    assert _multi_prepare(L, [64], 4096, 0)[0] == _lib.ERR_ARG and _multi_prepare(L, [64], 4096, 16)[0] == _lib.ERR_ARG
    assert _multi_prepare(L, [0], 4096, 15)[0] == _lib.ERR_ARG
    assert _multi_prepare(L, [64], 4096, 15, dtype=_lib.F32)[0] == _lib.ERR_UNSUPPORTED
    assert _multi_prepare(L, [64], 4096, 15, w_base=0x10000008)[0] == _lib.ERR_UNSUPPORTED   # weights off the 16-byte grid
    assert _multi_prepare(L, [64], 4096, 15, list_base=0x7000000002)[0] == _lib.ERR_ALIGN
    assert _multi_prepare(L, [64, 64], 4096, 15, table_bytes=88)[0] == _lib.ERR_WORKSPACE
    assert _multi_prepare(L, [300000], 8192, 15)[0] == _lib.ERR_UNSUPPORTED         # past the 32-bit offsets of the loads
